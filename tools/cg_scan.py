"""MAP@200 / precision@10 / RMSE of CG-trained factors against the exact f64 oracle on C1, per cg_steps"""
import sys, numpy as np, scipy.sparse as sps
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import recommender_amd as ra
from recommender_amd import metrics
from oracle import als_ref, c_oracle
from test_gpu_recommender import _c1_split, _precision_at_10
f, lamb, alpha, iters = 20, 0.01, 40.0, 15
N, M, tr, te = _c1_split()
X0, Y0 = als_ref.init_factors(N, M, f, "ref")
trt = sps.csr_matrix(tr.T); trt.sort_indices()
Xo, Yo = X0, Y0
for _ in range(iters):
    Xo = c_oracle.half_sweep(tr.indptr, tr.indices, tr.data, Yo, Xo, lamb, alpha, nthreads=8)
    Yo = c_oracle.half_sweep(trt.indptr, trt.indices, trt.data, Xo, Yo, lamb, alpha, nthreads=8)
users = metrics.users_to_test(tr)
oidx, _ = als_ref.recommend_full(Xo, Yo, users)
oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
po = _precision_at_10(oidx, users, tr, te, True)
tu, ti = te.nonzero()
ro = als_ref.rmse(np.einsum("ij,ij->i", Xo[tu], Yo[ti]), np.ones(len(tu)))
print("exact: MAP %.5f p@10 %.5f rmse %.5f" % (oscore, po, ro))
for steps in [int(a) for a in sys.argv[1:]]:
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M, als_kwargs={"f": f, "lamb": lamb, "alpha": alpha, "epochs": iters, "solver": "cg", "cg_steps": steps})
    rec.input_data((tr, te)); rec.train(evaluate=False)
    score = rec.evaluate(users)
    gidx, _ = rec.recommend(users, k=10, filter_rated=True)
    pg = _precision_at_10(gidx, users, tr, te, False)
    rg = metrics.rmse(rec.predict(tu, ti), np.ones(len(tu)))
    X = rec.X.cpu().numpy().astype(np.float64)
    print("cg_steps=%2d: MAP %.5f (%+.5f) p@10 %.5f (%+.5f) rmse %.5f (%+.5f) factor err %.2e" % (steps, score, score - oscore, pg, pg - po, rg, rg - ro, np.abs(X - Xo).max() / np.abs(Xo).max()))
