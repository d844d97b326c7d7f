"""GPU tests of the RecommenderALS facade and the scoring kernels against the oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sps

torch = pytest.importorskip("torch")

from oracle import als_ref, c_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd
    return recommender_amd


def _g2(golden):
    g = golden("g2_powerlaw_train.npz")
    shape = tuple(int(x) for x in g["U_shape"])
    U = sps.csr_matrix((g["U_data"], g["U_indices"], g["U_indptr"]), shape=shape)
    return g, U


@pytest.mark.parametrize("K,conv", [(20, "ref"), (20, "tf"), (64, "ref"), (8, "tf")])
def test_train_three_epochs_matches_alsref(ra, golden, K, conv):
    """RecommenderALS.train() vs ALSRef.train() on identical CSR input and seed-42 init:
    factors within 1e-4 relative (max-abs error / max-abs value) after every one of 3 epochs,
    trace within 1e-4 relative.  Where the float32 build of the CPU oracle itself cannot hold
    1e-4 against float64 (K=64 with only 200 users: the user Gram is close to singular), the
    bound is 3x that float32 deviation instead."""
    g, U = _g2(golden)
    tag = "K%d_%s" % (K, conv)
    Ut = sps.csr_matrix(U.T); Ut.sort_indices()
    Xf, Yf = g[tag + "_X0"].astype(np.float32), g[tag + "_Y0"].astype(np.float32)
    rec = ra.RecommenderALS(mode="train", n_users=U.shape[0], n_items=U.shape[1],
                            als_kwargs={"K": K, "lamb": 0.01, "alpha": 40.0, "steps": 3, "init": conv})
    rec.input_data(U)
    for ep in range(1, 4):
        rec.train(epochs=ep)
        X = rec.X.cpu().numpy().astype(np.float64); Y = rec.Y.cpu().numpy().astype(np.float64)
        Xg, Yg = g["%s_X%d" % (tag, ep)], g["%s_Y%d" % (tag, ep)]
        Xf = c_oracle.half_sweep(U.indptr, U.indices, U.data, Yf, Xf, 0.01, 40.0, dtype=np.float32)
        Yf = c_oracle.half_sweep(Ut.indptr, Ut.indices, Ut.data, Xf, Yf, 0.01, 40.0, dtype=np.float32)
        ex = np.abs(X - Xg).max() / np.abs(Xg).max(); ey = np.abs(Y - Yg).max() / np.abs(Yg).max()
        fx = np.abs(Xf - Xg).max() / np.abs(Xg).max(); fy = np.abs(Yf - Yg).max() / np.abs(Yg).max()
        assert ex < max(1e-4, 3 * fx) and ey < max(1e-4, 3 * fy), \
            "epoch %d: relative error X %.3e Y %.3e (float32 build of the oracle: %.3e / %.3e; bar 1e-4 or 3x that)" % (ep, ex, ey, fx, fy)
    np.testing.assert_allclose(rec.trace, g[tag + "_trace"], rtol=2e-4)


def test_save_load_predict_roundtrip(ra, golden, tmp_path):
    g, U = _g2(golden)
    path = str(tmp_path / "model")
    rec = ra.RecommenderALS(mode="train", n_users=U.shape[0], n_items=U.shape[1],
                            als_kwargs={"f": 20, "lamb": 0.01, "alpha": 40.0, "epochs": 2}, model_path=path)
    rec.input_data(U); rec.train(); rec.save(path)
    X = np.load(os.path.join(path, "X.npy")); Y = np.load(os.path.join(path, "Y.npy"))
    assert X.dtype == np.float64 and X.shape == (U.shape[0], 20) and Y.shape == (U.shape[1], 20)
    cfg = __import__("json").load(open(os.path.join(path, "algorithm_config.json")))
    assert cfg == {"epochs": 2, "start_epoch": 2, "alpha": 40.0, "lamb": 0.01}
    again = ra.RecommenderALS(mode="predict", model_path=path)
    u = np.array([0, 5, 199, 7]); i = np.array([3, 149, 0, 7])
    p = again.predict(u, i)
    np.testing.assert_allclose(p, np.einsum("ij,ij->i", X[u], Y[i]), rtol=2e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        again.train()
    # resuming: a model restored in train mode continues from start_epoch; the saved config
    # (epochs included) replaces the constructor's, as ALSTrainer.__env_restore does (ALS.py:297-298)
    cont = ra.RecommenderALS(mode="train", als_kwargs={"f": 20, "lamb": 0.5, "alpha": 1.0, "epochs": 9})
    cont.load(path); cont.input_data(U)
    assert cont.config["epochs"] == 2 and cont.config["lamb"] == 0.01
    assert cont.train() == [float(t) for t in rec.trace] and cont.start_epoch == 2       # nothing left to run
    assert cont.trace == [float(t) for t in rec.trace]             # the saved epochs' trace is read back ...
    cont.train(epochs=3)
    assert cont.start_epoch == 3 and len(cont.trace) == 3         # ... and the resumed epoch appended to it
    ref = np.abs(cont.X.cpu().numpy() - g["K20_ref_X3"]).max() / np.abs(g["K20_ref_X3"]).max()
    assert ref < 1e-4


def test_resumed_run_appends_to_eval_csv(ra, golden, tmp_path):
    """The reference's callback keeps every epoch's score and writes them all (EvalProto.py:74-94): a run that is
    saved, loaded and continued must end with one eval.csv line per epoch, not only the resumed ones."""
    g, U = _g2(golden)
    rng = np.random.default_rng(0)
    coo = U.tocoo()
    test = rng.random(coo.nnz) < 0.2
    tr = sps.csr_matrix((coo.data[~test], (coo.row[~test], coo.col[~test])), shape=U.shape)
    te = sps.csr_matrix((coo.data[test], (coo.row[test], coo.col[test])), shape=U.shape)
    path = str(tmp_path / "model")
    kw = {"f": 20, "lamb": 0.01, "alpha": 40.0, "epochs": 2}
    rec = ra.RecommenderALS(mode="train", n_users=U.shape[0], n_items=U.shape[1], als_kwargs=kw, model_path=path)
    rec.input_data((tr, te)); rec.train()
    first = open(os.path.join(path, "eval.csv")).read().strip().split("\n")
    assert first[0] == "epoch,score" and len(first) == 3
    cont = ra.RecommenderALS(mode="train", als_kwargs=kw, model_path=path)
    cont.load(path); cont.input_data((tr, te))
    assert cont.eval_epochs == [0, 1] and cont.eval_scores == [float(x) for x in rec.eval_scores]
    cont.train(epochs=4)
    lines = open(os.path.join(path, "eval.csv")).read().strip().split("\n")
    assert lines[:3] == first and [float(l.split(",")[0]) for l in lines[1:]] == [0.0, 1.0, 2.0, 3.0]
    assert len(open(os.path.join(path, "trace.csv")).read().strip().split("\n")) == 5
    # one uninterrupted run of 4 epochs ends with the same scores
    one = ra.RecommenderALS(mode="train", n_users=U.shape[0], n_items=U.shape[1], als_kwargs=dict(kw, epochs=4))
    one.input_data((tr, te)); one.train()
    np.testing.assert_allclose(cont.eval_scores, one.eval_scores, rtol=0, atol=1e-6)


def test_continues_a_run_folder_written_by_the_reference(ra, golden, tmp_path):
    """tests/golden/ref_deploy_als holds algorithm_config.json and eval.csv of a finished 15-epoch ALS run exactly as
    the reference's repository keeps them (deploy/ALS_2020-12-16.15-48-46-serve; written by ALS.py:283 and
    EvalProto.py:74-94).  The reference stores the INDEX of the last finished epoch as start_epoch (ALS.py:286,353), so its
    own resume repeats that epoch (ALS.py:343); loading such a folder here does the same - one more epoch, its score
    replacing the last line - and what is written back has the reference's keys, header and number format."""
    import json
    import shutil
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_deploy_als")
    path = str(tmp_path / "run")
    shutil.copytree(src, path)
    ref_cfg = json.load(open(os.path.join(src, "algorithm_config.json")))
    ref_eval = open(os.path.join(src, "eval.csv")).read().strip().split("\n")
    assert ref_cfg == {"epochs": 15, "start_epoch": 14, "alpha": 40.0, "lamb": 0.01} and len(ref_eval) == 16
    g, U = _g2(golden)
    rng = np.random.default_rng(3)
    coo = U.tocoo()
    test = rng.random(coo.nnz) < 0.2
    tr = sps.csr_matrix((coo.data[~test], (coo.row[~test], coo.col[~test])), shape=U.shape)
    te = sps.csr_matrix((coo.data[test], (coo.row[test], coo.col[test])), shape=U.shape)
    np.save(os.path.join(path, "X.npy"), rng.normal(0, 0.2, size=(U.shape[0], 20)))
    np.save(os.path.join(path, "Y.npy"), rng.normal(0, 0.2, size=(U.shape[1], 20)))
    rec = ra.RecommenderALS(mode="train", als_kwargs={"f": 8, "lamb": 0.5, "alpha": 1.0, "epochs": 3}, model_path=path)
    rec.load(path); rec.input_data((tr, te))
    assert (rec.config["epochs"], rec.config["alpha"], rec.config["lamb"], rec.start_epoch) == (15, 40.0, 0.01, 14)
    assert rec.eval_epochs == list(range(14))
    assert rec.eval_scores == [float(l.split(",")[1]) for l in ref_eval[1:15]]
    assert len(rec.train()) == 1 and rec.start_epoch == 15              # the one epoch the reference would run again
    cfg = json.load(open(os.path.join(path, "algorithm_config.json")))
    assert set(cfg) == set(ref_cfg) and cfg == {"epochs": 15, "start_epoch": 15, "alpha": 40.0, "lamb": 0.01}
    lines = open(os.path.join(path, "eval.csv")).read().strip().split("\n")
    assert lines[:15] == ref_eval[:15]                                   # header and 14 kept lines, character for character
    assert len(lines) == 16 and lines[15].split(",")[0] == "14.0" and 0.0 <= float(lines[15].split(",")[1]) <= 1.0


def test_recommend_and_similar_match_reference_ranking(ra):
    rng = np.random.default_rng(4)
    N, M, f = 130, 777, 20
    X = rng.normal(size=(N, f)); Y = rng.normal(size=(M, f)); Y[5] = 0.0      # a zero-norm item row
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M, als_kwargs={"f": f})
    rec.X = torch.from_numpy(X.astype(np.float32)).cuda(); rec.Y = torch.from_numpy(Y.astype(np.float32)).cuda()
    users = np.array([3, 0, 129, 64, 3])
    idx, sc = rec.recommend(users, k=200)
    ridx, rsc = als_ref.recommend_full(X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64), users)
    np.testing.assert_allclose(sc, rsc[:, :200], rtol=1e-5, atol=1e-5)
    assert (idx == ridx[:, :200]).mean() > 0.999               # identical ranking up to float32 near-ties
    assert np.all(np.diff(sc, axis=1) <= 0)
    items = np.array([0, 5, 776])
    sidx, ssc = rec.similar_to(items, k=50)
    oidx, osc = als_ref.cosine_similar(Y.astype(np.float32).astype(np.float64), items)
    np.testing.assert_allclose(ssc[[0, 2]], osc[[0, 2], :50], rtol=1e-5, atol=1e-5)
    assert sidx[0, 0] == 0 and sidx[2, 0] == 776 and np.all(ssc[1] == 0)       # zero row: all similarities 0
    assert (sidx[[0, 2]] == oidx[[0, 2], :50]).mean() > 0.99


def test_topk_ties_exclusions_and_large_rows(ra):
    rng = np.random.default_rng(8)
    n, M = 7, 100003
    S = rng.integers(0, 50, size=(n, M)).astype(np.float32)                    # massive ties
    S[2, :] = -np.inf; S[2, 17] = 1.0; S[3, 5] = np.nan
    ex_rows = [np.sort(rng.choice(M, size=c, replace=False)).astype(np.int32) for c in (0, 10, 3, 500, 0, 1, 99)]
    ptr = np.zeros(n + 1, dtype=np.int64); np.cumsum([len(e) for e in ex_rows], out=ptr[1:])
    for k in (1, 100, 1000):
        Sd = torch.from_numpy(S.copy()).cuda()
        idx, val = ra.topk_rows(Sd, k, (torch.from_numpy(ptr).cuda(), torch.from_numpy(np.concatenate(ex_rows)).cuda()))
        idx = idx.cpu().numpy(); val = val.cpu().numpy()
        for r in range(n):
            row = S[r].copy(); row[ex_rows[r]] = -np.inf
            row = np.where(np.isnan(row), -np.inf, row)                        # NaN ranks below everything
            order = np.lexsort((np.arange(M), -row))[:k]                      # score desc, id asc
            assert np.array_equal(idx[r], order), (k, r)
            exp = row[order]
            assert np.array_equal(val[r][np.isfinite(exp)], exp[np.isfinite(exp)])


def test_scores_edge_shapes(ra):
    """Both score kernels (f % 4 == 0 and f <= 128: three-way bf16 split on the matrix cores, every
    k-step count and ragged user / item / factor tails; otherwise float32 products).  The split is
    exact and the dropped piece products are below 2^-23 of a product, so the error bound is that of
    a float32 dot product: a few 2^-24 of sum |x_k y_k|."""
    rng = np.random.default_rng(2)
    for (nu, M, f) in [(1, 1, 5), (65, 257, 20), (3, 1000, 64), (70, 300, 50), (129, 33, 36), (200, 4000, 100),
                       (33, 70, 128), (300, 20000, 96), (5, 16385, 4), (17, 99, 132), (64, 513, 256)]:
        X = rng.normal(size=(nu, f)).astype(np.float32); Y = rng.normal(size=(M, f)).astype(np.float32)
        S = ra.scores(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()).cpu().numpy()
        exact = X.astype(np.float64) @ Y.astype(np.float64).T
        bound = 8 * 2.0 ** -24 * (np.abs(X).astype(np.float64) @ np.abs(Y).astype(np.float64).T) + 1e-30
        assert np.all(np.abs(S - exact) <= bound), (nu, M, f, float(np.max(np.abs(S - exact) / bound)))


def test_scores_user_subset_and_strided_factors(ra):
    """user_ids picks rows of X; factor matrices may be views with a leading dimension > f; a score
    does not depend on which other rows or items share the launch."""
    rng = np.random.default_rng(6)
    f, ld = 64, 80
    Xs = torch.from_numpy(rng.normal(size=(500, ld)).astype(np.float32)).cuda()
    Ys = torch.from_numpy(rng.normal(size=(3000, ld)).astype(np.float32)).cuda()
    X, Y = Xs[:, :f], Ys[:, :f]
    users = torch.from_numpy(rng.permutation(500)[:201]).cuda()
    S = ra.scores(X, Y, users)
    full = ra.scores(X.contiguous(), Y.contiguous())
    assert torch.equal(S, full[users])
    part = ra.scores(X, Y, users[:7], M=1000)
    assert torch.equal(part, full[users[:7], :1000])


def test_update_user_foldin(ra, golden):
    """test_ALS.py:88-101: a new user given user 0's ratings gets user 0's factors."""
    g, U = _g2(golden)
    N, M = U.shape
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M,
                            als_kwargs={"f": 20, "lamb": 0.01, "alpha": 40.0, "epochs": 2, "phantom_rows": 1})
    rec.input_data(U); rec.train()
    r, c, v = sps.find(U[0])
    rows = np.concatenate([r, np.full_like(r, N)])          # the phantom row N plays the brand-new user
    before = rec.X.clone()
    rec.update_user(rows, np.concatenate([c, c]), np.concatenate([v, v]))
    X = rec.X.cpu().numpy()
    assert np.array_equal(X[0], X[N])
    assert torch.equal(rec.X[1:N], before[1:N])
    # and it equals the oracle's fold-in of that row against the current item factors
    Yc = rec.Y.cpu().numpy().astype(np.float64)
    one = sps.csr_matrix(U[0])
    ref = als_ref.half_sweep_exact(one.indptr, one.indices, one.data, Yc, np.zeros((1, 20)), 0.01, 40.0)
    assert np.abs(X[0] - ref[0]).max() / np.abs(ref[0]).max() < 1e-4


def test_phantom_rows_enter_the_gram_like_the_tf_path(ra):
    """The TensorFlow path sizes its embedding layers N + 1 / M + 1 (ALS.py:456): the extra row is never updated
    but its random initial value is part of Y^T Y (ALS.py:243).  phantom_rows=1 reproduces that layout.  One
    iteration with a NON-ZERO phantom row against the float32 restatement of the padded-batch TF path
    (als_ref.half_sweep_batched_f32, ALS.py:160-233) and against the exact float64 solve with the phantom row in
    the Gram; the phantom rows stay untouched, and leaving them out of the Gram changes the factors."""
    rng = np.random.default_rng(31)
    N, M, f, lamb, alpha = 300, 200, 20, 0.01, 40.0
    U = sps.random(N, M, density=0.08, random_state=7, data_rvs=lambda n: rng.integers(1, 6, size=n).astype(np.float64)).tocsr()
    U.sort_indices()
    Ut = sps.csr_matrix(U.T); Ut.sort_indices()
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M,
                            als_kwargs={"f": f, "lamb": lamb, "alpha": alpha, "epochs": 1, "phantom_rows": 1, "init": "tf"})
    rec.input_data(U)
    rec._ensure_factors()
    X0 = rec.X.cpu().numpy().copy(); Y0 = rec.Y.cpu().numpy().copy()
    assert X0.shape == (N + 1, f) and Y0.shape == (M + 1, f)
    assert np.abs(Y0[M]).max() > 0 and np.abs(X0[N]).max() > 0          # the phantom rows are not zero
    rec.train(evaluate=False)
    X, Y = rec.X.cpu().numpy(), rec.Y.cpu().numpy()
    assert np.array_equal(X[N], X0[N]) and np.array_equal(Y[M], Y0[M])  # never updated
    # float32 restatement of the TF path: users against the item layer, then items against the new user layer
    Xb, _ = als_ref.half_sweep_batched_f32(U.indptr, U.indices, U.data, M, Y0, X0, lamb, alpha)
    Yb, _ = als_ref.half_sweep_batched_f32(Ut.indptr, Ut.indices, Ut.data, N, Xb, Y0, lamb, alpha)
    # exact float64 solve; its Gram runs over every row of the layer handed in, phantom row included
    Xe = als_ref.half_sweep_exact(U.indptr, U.indices, U.data, Y0.astype(np.float64), X0[:N].astype(np.float64), lamb, alpha)
    Xl = np.vstack([Xe, X0[N:].astype(np.float64)])
    Ye = als_ref.half_sweep_exact(Ut.indptr, Ut.indices, Ut.data, Xl, Y0[:M].astype(np.float64), lamb, alpha)

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    assert rel(X[:N], Xe) < 1e-4 and rel(Y[:M], Ye) < 1e-4
    # the float32 path inverts in float32 (np.linalg.inv, ALS.py:172): its own distance to the exact solve bounds ours to it
    assert rel(X[:N], Xb[:N]) < max(2e-3, 2 * rel(Xb[:N], Xe)) and rel(Y[:M], Yb[:M]) < max(2e-3, 2 * rel(Yb[:M], Ye))
    # without the phantom row in the Gram the result is another one
    Xn = als_ref.half_sweep_exact(U.indptr, U.indices, U.data, Y0[:M].astype(np.float64), X0[:N].astype(np.float64), lamb, alpha)
    assert rel(Xn, Xe) > 1e-3


def test_facade_names_and_refresh_gram(ra, golden):
    """The names of the reference's Environment facade (Environment.py:14-45): fit = train, similar(items) = cosine
    neighbours over the item factors (RecAlgos.py:61-73), make_update_data packs (rows, cols, vals) for update_user
    (ALS.py:356-373); update_user(refresh_gram=False) reuses the Gram of the last fold-in as ALS.py:375-385 reuses
    the one of the last epoch, refresh_gram=True recomputes it from the current item factors."""
    g, U = _g2(golden)
    N, M = U.shape
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M, als_kwargs={"f": 20, "lamb": 0.01, "alpha": 40.0, "epochs": 2})
    rec.input_data(U)
    rec.fit()
    assert rec.X is not None and len(rec.trace) == 2
    items = np.array([0, 5, 17])
    idx, sims = rec.similar(items)
    oidx, osims = als_ref.cosine_similar(rec.Y.cpu().numpy().astype(np.float64), items, M)
    assert idx.shape[0] == len(items) and np.allclose(np.asarray(sims)[:, 0], 1.0, atol=1e-5)        # an item is its own nearest
    k = min(10, np.asarray(idx).shape[1])
    assert np.array_equal(np.asarray(idx)[:, :k], oidx[:, :k])
    assert np.allclose(np.asarray(sims)[:, :k], osims[:, :k], atol=1e-5)
    r, c, v = sps.find(U[3])
    upd = rec.make_update_data((np.full_like(r, 3), c, v))
    assert rec.make_update_data((np.zeros(0), np.zeros(0), np.zeros(0))) == []           # ALS.py:390
    rec.update_user(upd, refresh_gram=True)
    fresh = rec.X[3].clone()
    one = sps.csr_matrix(U[3])
    Yc = rec.Y.cpu().numpy().astype(np.float64)
    ref = als_ref.half_sweep_exact(one.indptr, one.indices, one.data, Yc, np.zeros((1, 20)), 0.01, 40.0)
    assert np.abs(fresh.cpu().numpy() - ref[0]).max() / np.abs(ref[0]).max() < 1e-4
    # refresh_gram=False: the Gram the last epoch left behind, taken BEFORE its item sweep (ALS.py:345-349)
    rec.update_user(upd, refresh_gram=False)
    stale = rec.X[3].clone()
    assert not torch.equal(stale, fresh)
    Xt = rec.X.clone()
    ra.half_sweep(upd, rec.Y, rec.engine.cache_Y, Xt, 40.0)
    assert torch.equal(Xt[3], stale)
    rec.update_user(np.full_like(r, 3), c, v)                                            # the triplet form, refreshed
    assert torch.equal(rec.X[3], fresh)


def test_evaluate_matches_oracle_map(ra):
    """MAP@200 on a block-structured matrix (ExplicitDataDummy-like): same value as the oracle's
    evaluation of the oracle's factors to 3 decimals, and > 0.9 as tests/core/test_ALS.py:85-86."""
    rng = np.random.default_rng(12)
    m, n, nb = 10, 20, 10
    rows, cols, data = [], [], []
    for b in range(nb):
        rr, cc = np.meshgrid(np.arange(n * b, n * (b + 1)), np.arange(m * b, m * (b + 1)), indexing="ij")
        rows.append(rr.ravel()); cols.append(cc.ravel()); data.append(np.full(rr.size, b % 5 + 1.0))
    rows, cols, data = map(np.concatenate, (rows, cols, data))
    perm = rng.permutation(len(data)); cut = int(0.8 * len(data))
    N, M = n * nb, m * nb
    tr = sps.csr_matrix((data[perm[:cut]], (rows[perm[:cut]], cols[perm[:cut]])), shape=(N, M))
    te = sps.csr_matrix((data[perm[cut:]], (rows[perm[cut:]], cols[perm[cut:]])), shape=(N, M))
    tr.sort_indices(); te.sort_indices()
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M, als_kwargs={"f": 20, "lamb": 1e-2, "alpha": 40.0, "epochs": 3})
    rec.input_data((tr, te)); rec.train()
    users = np.arange(N)
    score = rec.evaluate(users)
    X0, Y0 = als_ref.init_factors(N, M, 20, "ref")
    Xo, Yo, _, _ = als_ref.train_exact(tr.indptr, tr.indices, tr.data, (N, M), 20, 1e-2, 40.0, 3, X0, Y0)
    oidx, _ = als_ref.recommend_full(Xo, Yo, users)
    oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
    assert abs(score - oscore) < 5e-4 and score > 0.9, (score, oscore)
    assert len(rec.eval_scores) == 3
    # evaluate() runs on the device end to end; the host loop over the reference's compute_ap gives the same number
    from recommender_amd import metrics
    recs, _ = rec.recommend(users, k=200)
    assert score == pytest.approx(metrics.mean_average_precision(recs, users, tr, te, 3.0, 200), rel=1e-12)
    # rmse(): predictions over the stored test entries against their ratings (utils.py:58-59), on the device
    tu, ti = te.nonzero()
    want = metrics.rmse(rec.predict(tu, ti), np.asarray(te[tu, ti]).ravel())
    assert rec.rmse() == pytest.approx(want, rel=1e-6)


def test_c1_full_training_metrics_to_three_decimals(ra):
    """The smallest benchmark configuration at its own size (ml-latest-small shaped synthetic: 610 x 9 724,
    100 836 ratings split 80/20, f=20, 15 iterations, alpha=40, lamb=0.01 as ALS.py:457):
    RecommenderALS.train() against the float64 oracle run from the same seed-42 init.
    Bars: factors 1e-4 relative after the 15th iteration; MAP@200 (the reference's eval.csv
    score), precision@10 and RMSE on the held-out entries identical to 3 decimals."""
    from recommender_amd import data, metrics
    N, M, f, lamb, alpha, iters = 610, 9724, 20, 0.01, 40.0, 15
    keys, r = data.synth_pairs(N, M, 100836, seed=1234, device="cpu")
    keys, r = keys.numpy(), r.numpy().astype(np.float64)
    rng = np.random.default_rng(0)
    test = rng.random(len(keys)) < 0.2
    def mat(mask):
        m = sps.csr_matrix((r[mask], (keys[mask] // M, keys[mask] % M)), shape=(N, M)); m.sort_indices(); return m
    tr, te = mat(~test), mat(test)
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M,
                            als_kwargs={"f": f, "lamb": lamb, "alpha": alpha, "epochs": iters})
    rec.input_data((tr, te))
    rec.train(evaluate=False)
    X0, Y0 = als_ref.init_factors(N, M, f, "ref")
    trt = sps.csr_matrix(tr.T); trt.sort_indices()
    Xo, Yo = X0, Y0
    for _ in range(iters):
        Xo = c_oracle.half_sweep(tr.indptr, tr.indices, tr.data, Yo, Xo, lamb, alpha, nthreads=8)
        Yo = c_oracle.half_sweep(trt.indptr, trt.indices, trt.data, Xo, Yo, lamb, alpha, nthreads=8)
    X = rec.X.cpu().numpy().astype(np.float64); Y = rec.Y.cpu().numpy().astype(np.float64)
    ex = np.abs(X - Xo).max() / np.abs(Xo).max(); ey = np.abs(Y - Yo).max() / np.abs(Yo).max()
    assert ex < 1e-4 and ey < 1e-4, (ex, ey)
    users = metrics.users_to_test(tr)
    score = rec.evaluate(users)
    oidx, _ = als_ref.recommend_full(Xo, Yo, users)
    oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
    assert abs(score - oscore) < 5e-4, (score, oscore)
    # precision@10 over the same users, training items filtered
    gidx, _ = rec.recommend(users, k=10, filter_rated=True)
    def prec(idx_rows, filt):
        hits = []
        for k, u in enumerate(users):
            trn = tr.indices[tr.indptr[u]:tr.indptr[u + 1]]
            row = [i for i in idx_rows[k] if not (filt and i in set(trn))][:10]
            tst = set(te.indices[te.indptr[u]:te.indptr[u + 1]])
            hits.append(sum(1 for i in row if i in tst) / 10.0)
        return float(np.mean(hits))
    assert abs(prec(gidx, False) - prec(oidx, True)) < 5e-4
    # RMSE of the predicted preference on the held-out entries (target p = 1)
    tu, ti = te.nonzero()
    rg = metrics.rmse(rec.predict(tu, ti), np.ones(len(tu)))
    ro = als_ref.rmse(np.einsum("ij,ij->i", Xo[tu], Yo[ti]), np.ones(len(tu)))
    assert abs(rg - ro) < 5e-4, (rg, ro)


def _c1_split():
    from recommender_amd import data
    N, M = 610, 9724
    keys, r = data.synth_pairs(N, M, 100836, seed=1234, device="cpu")
    keys, r = keys.numpy(), r.numpy().astype(np.float64)
    rng = np.random.default_rng(0)
    test = rng.random(len(keys)) < 0.2

    def mat(mask):
        m = sps.csr_matrix((r[mask], (keys[mask] // M, keys[mask] % M)), shape=(N, M)); m.sort_indices(); return m
    return N, M, mat(~test), mat(test)


def _precision_at_10(idx_rows, users, tr, te, filt):
    hits = []
    for k, u in enumerate(users):
        trn = set(tr.indices[tr.indptr[u]:tr.indptr[u + 1]])
        row = [i for i in idx_rows[k] if not (filt and i in trn)][:10]
        tst = set(te.indices[te.indptr[u]:te.indptr[u + 1]])
        hits.append(sum(1 for i in row if i in tst) / 10.0)
    return float(np.mean(hits))


CG_STEPS_DEFAULT = 10          # als.default_cg_steps(20)


@pytest.mark.parametrize("cg_steps", [CG_STEPS_DEFAULT, 2 * CG_STEPS_DEFAULT])
def test_c1_cg_training_metrics_to_three_decimals(ra, cg_steps):
    """The conjugate-gradient solver at metric level, on the smallest benchmark configuration at its own size: 15 iterations with
    `cg_steps` warm-started steps per row (10 is the default of RecommenderALS and of the benchmark's CG lines;
    3 steps miss MAP@200 by 5e-3 and 6 by 2e-3 here, 8 miss precision@10 by 6e-4)
    against the EXACT float64 solve of reclibwh/core/ALS.py:82-86 run from the same init.  The truncated iteration
    does not reproduce the exact factors (no factor bound here); what must hold is what a user of the model sees:
    MAP@200, precision@10 and held-out RMSE identical to 3 decimals."""
    from recommender_amd import metrics
    f, lamb, alpha, iters = 20, 0.01, 40.0, 15
    N, M, tr, te = _c1_split()
    rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M,
                            als_kwargs={"f": f, "lamb": lamb, "alpha": alpha, "epochs": iters, "solver": "cg",
                                        "cg_steps": cg_steps})
    rec.input_data((tr, te))
    rec.train(evaluate=False)
    X0, Y0 = als_ref.init_factors(N, M, f, "ref")
    trt = sps.csr_matrix(tr.T); trt.sort_indices()
    Xo, Yo = X0, Y0
    for _ in range(iters):
        Xo = c_oracle.half_sweep(tr.indptr, tr.indices, tr.data, Yo, Xo, lamb, alpha, nthreads=8)
        Yo = c_oracle.half_sweep(trt.indptr, trt.indices, trt.data, Xo, Yo, lamb, alpha, nthreads=8)
    users = metrics.users_to_test(tr)
    score = rec.evaluate(users)
    oidx, _ = als_ref.recommend_full(Xo, Yo, users)
    oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
    gidx, _ = rec.recommend(users, k=10, filter_rated=True)
    pg, po = _precision_at_10(gidx, users, tr, te, False), _precision_at_10(oidx, users, tr, te, True)
    tu, ti = te.nonzero()
    rg = metrics.rmse(rec.predict(tu, ti), np.ones(len(tu)))
    ro = als_ref.rmse(np.einsum("ij,ij->i", Xo[tu], Yo[ti]), np.ones(len(tu)))
    print("cg_steps=%d: MAP@200 %.5f vs %.5f, precision@10 %.5f vs %.5f, RMSE %.5f vs %.5f"
          % (cg_steps, score, oscore, pg, po, rg, ro))
    assert abs(score - oscore) < 5e-4, (score, oscore)
    assert abs(pg - po) < 5e-4, (pg, po)
    assert abs(rg - ro) < 5e-4, (rg, ro)


def _mid_split(N, M, nnz):
    from recommender_amd import data
    keys, r = data.synth_pairs(N, M, nnz, seed=1234, device="cpu")
    keys, r = keys.numpy(), r.numpy().astype(np.float64)
    test = np.random.default_rng(0).random(len(keys)) < 0.2

    def mat(mask):
        m = sps.csr_matrix((r[mask], (keys[mask] // M, keys[mask] % M)), shape=(N, M)); m.sort_indices(); return m
    return mat(~test), mat(test)


def _metrics_against_exact(ra, f, N, M, nnz, variants):
    """15 iterations from the seed-42 init with each of `variants` (als_kwargs) against the EXACT float64 solve of
    reclibwh/core/ALS.py:82-86 from the same init: MAP@200 (the reference's eval.csv score), precision@10 with the
    training items filtered and RMSE on the held-out entries, each within 5e-4."""
    from recommender_amd import metrics
    lamb, alpha, iters = 0.01, 40.0, 15
    tr, te = _mid_split(N, M, nnz)
    X0, Y0 = als_ref.init_factors(N, M, f, "ref")
    trt = sps.csr_matrix(tr.T); trt.sort_indices()
    Xo, Yo = X0, Y0
    nt = c_oracle.max_threads()
    for _ in range(iters):
        Xo = c_oracle.half_sweep(tr.indptr, tr.indices, tr.data, Yo, Xo, lamb, alpha, nthreads=nt)
        Yo = c_oracle.half_sweep(trt.indptr, trt.indices, trt.data, Xo, Yo, lamb, alpha, nthreads=nt)
    users = metrics.users_to_test(tr)
    oidx, _ = als_ref.recommend_full(Xo, Yo, users)
    oscore = als_ref.map_at_k(oidx, (te.indptr, te.indices, te.data), (tr.indptr, tr.indices, tr.data), users, 3.0, 200)
    po = _precision_at_10(oidx, users, tr, te, True)
    tu, ti = te.nonzero()
    ro = als_ref.rmse(np.einsum("ij,ij->i", Xo[tu], Yo[ti]), np.ones(len(tu)))
    out = {}
    for name, kw in variants.items():
        rec = ra.RecommenderALS(mode="train", n_users=N, n_items=M,
                                als_kwargs=dict({"f": f, "lamb": lamb, "alpha": alpha, "epochs": iters}, **kw))
        rec.input_data((tr, te))
        rec.train(evaluate=False)
        score = rec.evaluate(users)
        gidx, _ = rec.recommend(users, k=10, filter_rated=True)
        pg = _precision_at_10(gidx, users, tr, te, False)
        rg = metrics.rmse(rec.predict(tu, ti), np.ones(len(tu)))
        ferr = np.abs(rec.X.cpu().numpy() - Xo).max() / np.abs(Xo).max()
        print("f=%d %s: MAP@200 %.5f vs %.5f, precision@10 %.5f vs %.5f, RMSE %.5f vs %.5f, factor error %.2e"
              % (f, name, score, oscore, pg, po, rg, ro, ferr))
        assert abs(score - oscore) < 5e-4, (name, score, oscore)
        assert abs(pg - po) < 5e-4, (name, pg, po)
        assert abs(rg - ro) < 5e-4, (name, rg, ro)
        out[name] = (rec, ferr)
    return out


def test_f128_training_metrics_to_three_decimals(ra):
    """f = 128 at metric level on a mid-size synthetic (8 000 x 4 000, 500 000 ratings split 80/20) both oracles can walk:
    the exact solver with every short row in its entry-space form (factors also within 1e-4 after the 15th iteration),
    and the conjugate-gradient solver on the formed matrix at its default step count for this f (22; 16 steps miss
    precision@10 by 1.6e-3 here)."""
    from recommender_amd.als import default_cg_steps
    assert default_cg_steps(128) == 22 and default_cg_steps(20) == 10
    out = _metrics_against_exact(ra, 128, 8000, 4000, 500000,
                                 {"exact, short rows in entry space": {"short_rows": True},
                                  "cg on the formed matrix, default steps": {"solver": "cg"}})
    rec, ferr = out["exact, short rows in entry space"]
    assert rec.engine.short == {"users": True, "items": True} and ferr < 1e-4
    assert out["cg on the formed matrix, default steps"][0].engine.cg_steps == 22


def test_f64_training_metrics_to_three_decimals(ra):
    """The benchmark's one-GPU configuration (f = 64, exact solve, library defaults) at metric level: 15 iterations on
    the 8 000 x 4 000 synthetic - the fixed factors have at least 16 rows per factor, so the default accumulate is
    fp16 x 2, as on the 25 M-entry matrix - against the exact float64 solve from the same init: MAP@200, precision@10 and
    RMSE within 5e-4, factors within 1e-4 after the 15th iteration."""
    out = _metrics_against_exact(ra, 64, 8000, 4000, 500000, {"exact, library defaults": {}})
    rec, ferr = out["exact, library defaults"]
    assert rec.engine.solver == "cholesky" and ferr < 1e-4


def test_f256_lock_step_cg_training_metrics_to_three_decimals(ra):
    """f = 256 - the factor count of the 4 B-entry configuration, where the lock-step conjugate-gradient form is the
    solver - at metric level on a 4 000 x 2 000 synthetic with 250 000 ratings: 15 iterations at the default step
    count for this f (43; 36 steps miss precision@10 by 6e-4 here, 42 hold) against the exact float64 solve."""
    from recommender_amd.als import default_cg_steps
    assert default_cg_steps(256) == 43
    out = _metrics_against_exact(ra, 256, 4000, 2000, 250000, {"lock-step cg, default steps": {}})
    eng = out["lock-step cg, default steps"][0].engine
    assert eng.solver == "cg" and eng.cg_steps == 43


def test_exclusions_indexed_by_user_id_on_the_device(ra):
    """The "drop rated items" filter (RecommenderService.py:16-29) from the training CSR as it lies on the device:
    the kernels read row user_ids[u] of it, for any block of users in any order - the same ranking as from host
    arrays, as from the two-call path, and as from a CSR cut to the block by hand."""
    rng = np.random.default_rng(44)
    N, M, f, k = 700, 40000, 64, 50
    X = torch.from_numpy(rng.normal(size=(N, f)).astype(np.float32)).cuda()
    Y = torch.from_numpy(rng.normal(size=(M, f)).astype(np.float32)).cuda()
    rows = [np.sort(rng.choice(M, size=int(c), replace=False)).astype(np.int32) for c in rng.integers(0, 300, size=N)]
    rows[5] = np.zeros(0, np.int32)
    ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    col = np.concatenate(rows)
    users = torch.from_numpy(rng.permutation(N)[:333].astype(np.int64)).cuda()
    dev = (torch.from_numpy(ptr).cuda(), torch.from_numpy(col).cuda())
    i0, v0 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=(ptr, col), block_users=128)        # host arrays
    i1, v1 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=dev, block_users=100)               # device tensors
    i2, v2 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=dev, fused=False)                   # two-call path
    assert torch.equal(i0, i1) and torch.equal(v0, v1) and torch.equal(i0, i2) and torch.equal(v0, v2)
    uh = users.cpu().numpy()
    for j in range(0, len(uh), 7):
        assert not np.intersect1d(i0[j].cpu().numpy(), rows[uh[j]]).size
    # the block CSR built by hand (row j = the exclusions of users[j]) through topk_rows gives the same
    S = ra.scores(X, Y, users)
    bp = np.zeros(len(uh) + 1, dtype=np.int64); np.cumsum([len(rows[u]) for u in uh], out=bp[1:])
    bc = np.concatenate([rows[u] for u in uh])
    i3, v3 = ra.topk_rows(S.clone(), k, (torch.from_numpy(bp).cuda(), torch.from_numpy(bc).cuda()))
    i4, v4 = ra.topk_rows(S.clone(), k, dev, users)
    assert torch.equal(i3, i0) and torch.equal(i4, i0) and torch.equal(v3, v0) and torch.equal(v4, v0)
    with pytest.raises(ValueError):
        ra.ops.recommend_topk(X, Y, users, k, exclude_csr=(ptr[:300], col))        # a user id beyond the CSR


def _two_call(ra, X, Y, users, k, ex):
    return ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex, fused=False)


@pytest.mark.parametrize("f,k", [(32, 100), (20, 7), (64, 1024), (21, 10), (128, 100)])
def test_fused_topk_equals_two_call_path(ra, f, k):
    """The fused top-k (threshold from the first 16 384 items, the rest filtered in the score
    kernel's registers in two growing ranges, candidate rows cut back to k in between; f=21 takes
    the element-wise fp32 kernel) returns exactly what scores + top-k on the materialised matrix returns:
    same items in the same order (ties by ascending id: factors are quantised so that equal
    scores abound), same score bits, with and without per-user exclusions."""
    rng = np.random.default_rng(f + k)
    nu, M = 150, 200000
    X = torch.from_numpy(rng.integers(-2, 3, size=(nu, f)).astype(np.float32) / 4).cuda()
    Y = torch.from_numpy(rng.integers(-2, 3, size=(M, f)).astype(np.float32) / 4).cuda()
    users = torch.arange(nu, device="cuda")
    i0, v0 = _two_call(ra, X, Y, users, k, None)
    i1, v1 = ra.ops.recommend_topk(X, Y, users, k)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    # exclusions: every user drops its own best 30 items (so the threshold must respect them) plus random ones
    best = i0[:, :30].cpu().numpy()
    rows = [np.unique(np.concatenate([best[u], rng.integers(0, M, size=50)])).astype(np.int32) for u in range(nu)]
    ptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    ex = (ptr, np.concatenate(rows))
    j0, w0 = _two_call(ra, X, Y, users, k, ex)
    j1, w1 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex)
    assert torch.equal(j0, j1) and torch.equal(w0, w1)
    for u in range(0, nu, 37):
        assert not np.intersect1d(j1[u].cpu().numpy(), rows[u]).size


def test_fused_topk_overflow_falls_back(ra):
    """Adversarial item order: the first 65 536 items all score below the rest, so every item of
    the last range beats the threshold, the candidate lists overflow and the block is redone by the
    two-call path - the result is still exact."""
    rng = np.random.default_rng(3)
    nu, M, f, k = 70, 100000, 16, 50
    X = torch.from_numpy(np.abs(rng.normal(size=(nu, f))).astype(np.float32)).cuda()
    Yh = np.abs(rng.normal(size=(M, f))).astype(np.float32)
    Yh[:65536] *= 1e-3
    Y = torch.from_numpy(Yh).cuda()
    users = torch.arange(nu, device="cuda")
    i0, v0 = _two_call(ra, X, Y, users, k, None)
    i1, v1 = ra.ops.recommend_topk(X, Y, users, k)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)


def test_heavy_users_overflow_alone(ra):
    """Users whose excluded (rated) items are exactly their best-scoring ones, 20 000 of them: every such item of a
    later range beats the threshold and takes a candidate slot before it is dropped, so these users' lists overflow.
    They alone are redone by the two-call path (the library marks their rows); their neighbours in the same block
    keep the fused result, and everything equals the two-call path bit for bit."""
    rng = np.random.default_rng(12)
    nu, M, f, k = 300, 100000, 64, 100
    Xh = rng.normal(size=(nu, f)).astype(np.float32); Yh = rng.normal(size=(M, f)).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); Y = torch.from_numpy(Yh).cuda()
    heavy = [7, 150, 299]
    rows = [np.sort(rng.choice(M, size=int(c), replace=False)).astype(np.int32) for c in rng.integers(0, 200, size=nu)]
    for u in heavy:
        sc = Yh.astype(np.float64) @ Xh[u].astype(np.float64)
        rows[u] = np.sort(np.argsort(-sc)[:20000]).astype(np.int32)
    ptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    ex = (ptr, np.concatenate(rows))
    users = torch.arange(nu, device="cuda")
    from recommender_amd._lib import lib
    # the library call itself: an overflow count, the three heavy rows marked with item -1, every other row final
    ws = torch.empty(lib.als_recommend_topk_workspace_bytes(nu, M, k), dtype=torch.uint8, device="cuda")
    dex = ra.ops.device_exclusions(ex, "cuda")
    idx = torch.empty((nu, k), dtype=torch.int32, device="cuda"); val = torch.empty((nu, k), dtype=torch.float32, device="cuda")
    over = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.als_recommend_topk_f32(X.data_ptr(), f, users.data_ptr(), nu, Y.data_ptr(), f, M, f, k, dex[0].data_ptr(),
                                    dex[1].data_ptr(), 1, idx.data_ptr(), val.data_ptr(), over.data_ptr(), ws.data_ptr(),
                                    ws.numel(), None)
    torch.cuda.synchronize()
    assert rc == 0 and int(over.item()) > 0
    marked = torch.nonzero(idx[:, 0] < 0).flatten().cpu().tolist()
    assert marked == heavy and bool((idx[heavy] == -1).all())
    j0, w0 = _two_call(ra, X, Y, users, k, ex)
    keep = [u for u in range(nu) if u not in heavy]
    assert torch.equal(idx[keep], j0[keep]) and torch.equal(val[keep], w0[keep])
    # the wrapper redoes the marked users only and returns the two-call result for everyone
    j1, w1 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex)
    assert torch.equal(j0, j1) and torch.equal(w0, w1)
    for u in heavy:
        assert not np.intersect1d(j1[u].cpu().numpy(), rows[u]).size


def test_device_transpose_equals_scipy(ra):
    rng = np.random.default_rng(2)
    U = sps.random(300, 170, density=0.05, random_state=7, format="csr", dtype=np.float64)
    U.data = rng.integers(1, 11, size=U.nnz) * 0.5
    U.sort_indices()
    t = ra.DeviceCSR.from_scipy(U, "cuda").transpose()
    Ut = sps.csr_matrix(U.T); Ut.sort_indices()
    assert t.shape == Ut.shape and t.nnz == Ut.nnz
    assert np.array_equal(t.rowptr.cpu().numpy(), Ut.indptr)
    assert np.array_equal(t.col.cpu().numpy(), Ut.indices)
    assert np.array_equal(t.val.cpu().numpy(), Ut.data.astype(np.float32))


@pytest.mark.parametrize("nu,M,f,k", [(1, 16385, 4, 1), (3, 16384, 8, 16), (5, 16416, 64, 100), (2, 65537, 128, 1024),
                                      (129, 20000, 36, 1024), (7, 300000, 100, 200), (33, 70000, 132, 50),
                                      (4, 1024, 16, 1024), (130, 262145, 32, 7)])
def test_fused_topk_edge_shapes(ra, nu, M, f, k):
    """Range boundaries of the fused path (one item past the first range, exactly the first range, a
    first range shorter than 16 k, k = 1024, a single user, the float32 kernel at f = 132), without
    exclusions and with all but 2k+5 items of every user excluded: same bits as the two-call path."""
    rng = np.random.default_rng(nu + M + f + k)
    X = torch.from_numpy(rng.integers(-3, 4, size=(nu + 3, f)).astype(np.float32) / 4).cuda()
    Y = torch.from_numpy(rng.integers(-3, 4, size=(M, f)).astype(np.float32) / 4).cuda()
    users = torch.arange(nu, device="cuda") + 2
    i0, v0 = _two_call(ra, X, Y, users, k, None)
    i1, v1 = ra.ops.recommend_topk(X, Y, users, k)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    keep = [np.sort(rng.choice(M, size=min(M, 2 * k + 5), replace=False)) for _ in range(nu + 3)]
    rows = [np.setdiff1d(np.arange(M), kp).astype(np.int32) for kp in keep]
    ptr = np.zeros(nu + 4, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    ex = (ptr, np.concatenate(rows))
    j0, w0 = _two_call(ra, X, Y, users, k, ex)
    j1, w1 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex)
    assert torch.equal(j0, j1) and torch.equal(w0, w1)
