"""The oracle (oracle/als_ref.py, oracle/als_ref.c) against golden vectors minted from the
reference's own ALSRef / utils / eval_utils (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import als_ref, c_oracle


def _csr(g, p="U"):
    return g[p + "_indptr"], g[p + "_indices"], g[p + "_data"], tuple(g[p + "_shape"])


def test_g1_single_step(golden):
    g = golden("g1_single_step.npz")
    ptr, ind, dat, _ = _csr(g)
    X = als_ref.half_sweep_exact(ptr, ind, dat, g["Yinit"], g["Xinit"], float(g["lamb"]), float(g["alpha"]))
    np.testing.assert_allclose(X, g["Xnew"], rtol=1e-11, atol=1e-13)
    Xc = c_oracle.half_sweep(ptr, ind, dat, g["Yinit"], g["Xinit"], float(g["lamb"]), float(g["alpha"]))
    np.testing.assert_allclose(Xc, g["Xnew"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("K", [8, 20, 32, 64])
@pytest.mark.parametrize("conv", ["ref", "tf"])
def test_g2_train(golden, K, conv):
    g = golden("g2_powerlaw_train.npz")
    ptr, ind, dat, shape = _csr(g)
    tag = "K%d_%s" % (K, conv)
    X0, Y0 = als_ref.init_factors(shape[0], shape[1], K, conv)
    assert np.array_equal(X0, g[tag + "_X0"]) and np.array_equal(Y0, g[tag + "_Y0"])
    X, Y, trace, hist = als_ref.train_exact(ptr, ind, dat, shape, K, float(g["lamb"]), float(g["alpha"]), 3, X0, Y0)
    for s in range(3):
        np.testing.assert_allclose(hist[s][0], g["%s_X%d" % (tag, s + 1)], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(hist[s][1], g["%s_Y%d" % (tag, s + 1)], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(trace, g[tag + "_trace"], rtol=1e-9)


@pytest.mark.parametrize("K", [20, 64])
def test_g2_c_oracle_one_iteration(golden, K):
    g = golden("g2_powerlaw_train.npz")
    ptr, ind, dat, shape = _csr(g)
    tag = "K%d_ref" % K
    X1 = c_oracle.half_sweep(ptr, ind, dat, g[tag + "_Y0"], g[tag + "_X0"], 0.01, 40.0, nthreads=2)
    np.testing.assert_allclose(X1, g[tag + "_X1"], rtol=1e-8, atol=1e-10)
    tptr, tind, tdat = als_ref.csr_transpose(ptr, ind, dat, shape[1])
    Y1 = c_oracle.half_sweep(tptr, tind, tdat, X1, g[tag + "_Y0"], 0.01, 40.0, nthreads=2)
    np.testing.assert_allclose(Y1, g[tag + "_Y1"], rtol=1e-7, atol=1e-9)
    # float32 variant of the same arithmetic stays within float32 rounding of the float64 result
    X1f = c_oracle.half_sweep(ptr, ind, dat, g[tag + "_Y0"], g[tag + "_X0"], 0.01, 40.0, dtype=np.float32)
    assert np.abs(X1f - g[tag + "_X1"]).max() / np.abs(g[tag + "_X1"]).max() < 2e-3


def test_g3_edge_cases(golden):
    g = golden("g3_edge_cases.npz")
    ptr, ind, dat, shape = _csr(g)
    lamb, alpha = float(g["lamb"]), float(g["alpha"])
    for impl in (als_ref.half_sweep_exact, c_oracle.half_sweep):
        X = impl(ptr, ind, dat, g["Y0"], g["X0"], lamb, alpha)
        np.testing.assert_allclose(X, g["Xnew"], rtol=1e-8, atol=1e-10)
        assert np.all(X[0] == 0) and np.all(X[-1] == 0)         # empty rows solve to exactly 0
        tptr, tind, tdat = als_ref.csr_transpose(ptr, ind, dat, shape[1])
        Y = impl(tptr, tind, tdat, X, g["Y0"], lamb, alpha)
        np.testing.assert_allclose(Y, g["Ynew"], rtol=1e-7, atol=1e-9)
        assert np.all(Y[7] == 0)                                   # empty column


def test_g4_metrics(golden):
    g = golden("g4_metrics.npz")
    for k in range(g["rec"].shape[0]):
        rec = g["rec"][k][g["rec"][k] >= 0]; te = g["test"][k][g["test"][k] >= 0]; tr = g["train"][k][g["train"][k] >= 0]
        assert als_ref.compute_ap(rec, te, tr) == pytest.approx(g["ap"][k], abs=1e-15)
        assert als_ref.compute_auc(rec, te, tr) == pytest.approx(g["auc"][k], abs=1e-15)
    assert als_ref.rmse(g["rmse_x"], g["rmse_y"]) == pytest.approx(float(g["rmse"]), abs=1e-15)


def test_g5_foldin(golden):
    g = golden("g5_foldin.npz")
    ptr, ind, dat, _ = _csr(g)
    X = als_ref.half_sweep_exact(ptr, ind, dat, g["Y0"], g["X0"], float(g["lamb"]), float(g["alpha"]))
    np.testing.assert_allclose(X, g["Xnew"], rtol=1e-6, atol=1e-9)     # lamb=1e-6: ill conditioned
    assert np.array_equal(X[0], X[1])
    # visiting a subset of rows leaves the others untouched (fold-in, ALS.py:375-395)
    X2 = als_ref.half_sweep_exact(ptr, ind, dat, g["Y0"], g["X0"], float(g["lamb"]), float(g["alpha"]), rows=[1])
    assert np.array_equal(X2[0], g["X0"][0]) and np.array_equal(X2[1], X[1])


def test_g6_padded_batch(golden):
    g = golden("g6_padded_batch.npz")
    ptr, ind, dat, _ = _csr(g)
    rp, yp = als_ref.padded_batch(ptr, ind, dat, g["users"], int(g["pad_val"]))
    assert np.array_equal(rp, g["rp"]) and np.array_equal(yp, g["yp"])


def test_batched_f32_matches_exact_at_reference_test_settings(golden):
    """tests/core/test_ALS.py:18-53 asserts the TF path allclose ALSRef at K=5, alpha=1, lamb=1e-2;
    the float32 restatement of the TF path must meet the same bar (no phantom contribution:
    the pad row is zero here, as an out-of-range GPU gather returns)."""
    g = golden("g1_single_step.npz")
    ptr, ind, dat, shape = _csr(g)
    Ylayer = np.vstack([g["Yinit"], np.zeros((1, g["Yinit"].shape[1]))])
    X, diff = als_ref.half_sweep_batched_f32(ptr, ind, dat, shape[1], Ylayer, g["Xinit"], float(g["lamb"]),
                                             float(g["alpha"]), batchsize=shape[0])
    assert np.allclose(X, g["Xnew"], rtol=1e-4, atol=1e-5)
    assert diff == pytest.approx(np.mean(np.abs(g["Xinit"] - g["Xnew"])), rel=1e-4)


@pytest.mark.parametrize("K", [8, 20, 32])
def test_cg_oracle_pinned_through_its_limit(golden, K):
    """half_sweep_cg restates a published algorithm the reference does not contain; what pins it
    to the reference is that, run to convergence, it must land on ALSRef's exact half-sweep
    (golden vectors), and that fewer steps move monotonically towards it."""
    g = golden("g2_powerlaw_train.npz")
    ptr, ind, dat, shape = _csr(g)
    tag = "K%d_ref" % K
    X0, Y0, X1 = g[tag + "_X0"], g[tag + "_Y0"], g[tag + "_X1"]
    errs = []
    for steps in (1, 3, K, 3 * K):
        X = als_ref.half_sweep_cg(ptr, ind, dat, Y0, X0, float(g["lamb"]), float(g["alpha"]), steps)
        errs.append(np.abs(X - X1).max() / np.abs(X1).max())
    assert errs[-1] < 1e-9
    assert errs[0] > errs[1] > errs[2] >= errs[3]


def test_cg_oracle_edge_cases(golden):
    g = golden("g3_edge_cases.npz")
    ptr, ind, dat, shape = _csr(g)
    K = g["Y0"].shape[1]
    X = als_ref.half_sweep_cg(ptr, ind, dat, g["Y0"], g["X0"], 0.01, 40.0, 4 * K)
    np.testing.assert_allclose(X, g["Xnew"], rtol=1e-7, atol=1e-9)
    assert np.all(X[0] == 0) and np.all(X[-1] == 0)          # rows without entries: exactly 0
