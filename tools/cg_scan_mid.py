"""MAP@200 / precision@10 / RMSE of CG-trained factors against the exact float64 oracle on a mid-size synthetic
matrix, per cg_steps, at the factor counts the CG solver ships for (f = 128 on the formed matrix, f = 256 in lock
step).  usage: python tools/cg_scan_mid.py <f> <N> <M> <nnz> steps..."""
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import recommender_amd as ra
from recommender_amd import data, metrics
from oracle import als_ref, c_oracle
from test_gpu_recommender import _precision_at_10


def split(N, M, nnz, seed=1234):
    keys, r = data.synth_pairs(N, M, nnz, seed=seed, device="cpu")
    keys, r = keys.numpy(), r.numpy().astype(np.float64)
    test = np.random.default_rng(0).random(len(keys)) < 0.2

    def mat(mask):
        m = sps.csr_matrix((r[mask], (keys[mask] // M, keys[mask] % M)), shape=(N, M)); m.sort_indices(); return m
    return mat(~test), mat(test)


def main():
    f, N, M, nnz = (int(a) for a in sys.argv[1:5])
    lamb, alpha, iters = 0.01, 40.0, 15
    tr, te = split(N, M, nnz)
    X0, Y0 = als_ref.init_factors(N, M, f, "ref")
    trt = sps.csr_matrix(tr.T); trt.sort_indices()
    Xo, Yo = X0, Y0
    t0 = time.time()
    nt = c_oracle.max_threads()
    for _ in range(iters):
        Xo = c_oracle.half_sweep(tr.indptr, tr.indices, tr.data, Yo, Xo, lamb, alpha, nthreads=nt)
        Yo = c_oracle.half_sweep(trt.indptr, trt.indices, trt.data, Xo, Yo, lamb, alpha, nthreads=nt)
    print("oracle: %.1f s on %d threads" % (time.time() - t0, nt))
    users = metrics.users_to_test(tr)
    oidx, _ = als_ref.recommend_full(Xo, Yo, users)
    oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
    po = _precision_at_10(oidx, users, tr, te, True)
    tu, ti = te.nonzero()
    ro = als_ref.rmse(np.einsum("ij,ij->i", Xo[tu], Yo[ti]), np.ones(len(tu)))
    print("exact: %d users tested, MAP %.5f p@10 %.5f rmse %.5f" % (len(users), oscore, po, ro))
    for steps in [int(a) for a in sys.argv[5:]]:
        kw = {"f": f, "lamb": lamb, "alpha": alpha, "epochs": iters}
        if steps > 0:
            kw.update(solver="cg", cg_steps=steps)
        elif steps < 0:
            kw.update(short_rows=True)
        rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M, als_kwargs=kw)
        rec.input_data((tr, te)); rec.train(evaluate=False)
        score = rec.evaluate(users)
        gidx, _ = rec.recommend(users, k=10, filter_rated=True)
        pg = _precision_at_10(gidx, users, tr, te, False)
        rg = metrics.rmse(rec.predict(tu, ti), np.ones(len(tu)))
        X = rec.X.cpu().numpy().astype(np.float64)
        print("%s: MAP %.5f (%+.5f) p@10 %.5f (%+.5f) rmse %.5f (%+.5f) factor err %.2e"
              % ("cg_steps=%2d" % steps if steps > 0 else "exact" + (" short-rows" if steps < 0 else ""), score, score - oscore,
                 pg, pg - po, rg, rg - ro, np.abs(X - Xo).max() / np.abs(Xo).max()), flush=True)


if __name__ == "__main__":
    main()
