"""GPU parity tests of the conjugate-gradient half-sweep (any f <= 256) and of the wide Gram
(128 < f <= 256), through the C ABI.

The reference has no CG solver (it inverts every row's system, reclibwh/core/ALS.py:82,172), so
the checker is oracle/als_ref.py::half_sweep_cg - the published algorithm restated in float64 -
which tests/test_oracle.py pins to the golden vectors of the reference's ALSRef through its
limit (enough steps reproduce the exact solve).  Bars: the HIP path (float32) against the oracle
at the SAME step count, 2e-4 of the largest factor entry (three steps of float32 CG carry a few
more roundings than one solve); against the exact golden solve, 1e-4 once the iteration has
converged.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import als_ref  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd
    return recommender_amd


def _dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _csr(g, p="U"):
    return g[p + "_indptr"], g[p + "_indices"], g[p + "_data"], tuple(int(x) for x in g[p + "_shape"])


def _random_csr(rng, N, M, avg, maxlen):
    lens = np.minimum(rng.zipf(1.6, size=N) + rng.integers(0, 2 * avg, size=N), maxlen).astype(np.int64)
    lens[rng.integers(0, N, size=max(N // 50, 1))] = 0
    ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    return ptr, ind, dat


def _sweep_cg(ra, ptr, ind, dat, shape, Yfix, Xold, lamb, alpha, steps, chunk_nnz=2048, row_ids=None, form="auto"):
    csr = ra.DeviceCSR(ptr, ind, dat, shape, "cuda", chunk_nnz=chunk_nnz, row_ids=row_ids)
    Y = _dev(Yfix); X = _dev(Xold)
    gc = ra.gram(Y, lamb)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = ra.half_sweep_cg(csr, Y, gc, X, alpha, steps, info=info, form=form)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return X.cpu().numpy().astype(np.float64), float(d.item())


# ----------------------------------------------------------------------------- wide Gram
@pytest.mark.parametrize("f", [129, 160, 192, 200, 256])
@pytest.mark.parametrize("n", [1, 100, 4097, 70001])
def test_wide_gram_matches_numpy(ra, f, n):
    rng = np.random.default_rng(f * 1000 + n)
    Y = rng.normal(0, 1 / np.sqrt(f), size=(n, f)).astype(np.float32)
    gc = ra.gram(_dev(Y), 0.5)
    torch.cuda.synchronize()
    G = gc.gram.cpu().numpy()
    ref = Y.astype(np.float64).T @ Y.astype(np.float64)
    assert np.abs(G - ref).max() <= 3e-6 * np.sqrt(n) * max(np.abs(ref).max(), 1.0)
    assert np.array_equal(G, G.T)


def test_wide_gram_partials_equal_single_call(ra):
    rng = np.random.default_rng(6)
    n, f = 2 * 65536 + 333, 192
    Y = _dev(rng.normal(0, 0.2, size=(n, f)).astype(np.float32))
    whole = ra.gram(Y, 0.5)
    parts = [ra.gram_partials(Y, 0, 65536), ra.gram_partials(Y, 65536, n)]
    sharded = ra.gram_finalize(torch.cat(parts, 0).contiguous(), f, 0.5)
    torch.cuda.synchronize()
    assert torch.equal(whole.gram, sharded.gram)


# ----------------------------------------------------------------------------- CG vs the oracle
@pytest.mark.parametrize("f,form", [(8, "auto"), (20, "matrix"), (20, "lock_step"), (30, "matrix"), (30, "lock_step"),
                                    (64, "auto"), (64, "lock_step"), (100, "auto"), (128, "matrix"), (128, "lock_step"),
                                    (192, "auto"), (256, "lock_step"), (130, "matrix"), (160, "matrix"), (192, "matrix"),
                                    (200, "matrix"), (256, "matrix")])
@pytest.mark.parametrize("steps", [1, 3])
def test_cg_matches_oracle_same_steps(ra, f, steps, form):
    """All implementations of the iteration: on the formed matrix in one wave's registers (f <= 128)
    or spread over a 4-wave workgroup (128 < f <= 256), one gather pass each, and in lock step
    over all rows (any f, steps + 1 gather passes)."""
    rng = np.random.default_rng(500 + f)
    N, M = 600, 900
    ptr, ind, dat = _random_csr(rng, N, M, 30, 850)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = als_ref.half_sweep_cg(ptr, ind, dat, Y0, X0, 0.01, 40.0, steps)
    X, d = _sweep_cg(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, steps, form=form)
    assert _rel(X, ref) < 2e-4
    assert d == pytest.approx(np.abs(X0.astype(np.float64) - ref).sum(), rel=1e-4)
    X2, d2 = _sweep_cg(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, steps, form=form)   # bit-for-bit reproducible
    assert np.array_equal(X, X2) and d == d2


@pytest.mark.parametrize("form", ["matrix", "lock_step"])
@pytest.mark.parametrize("chunk_nnz", [64, 128, 2048])
def test_cg_edge_cases_and_long_rows(ra, golden, chunk_nnz, form):
    """Empty rows (exactly 0), single-entry rows, one long row cut into chunks: the chunked sum is
    within rounding of the whole-row sum, and with enough steps CG lands on the reference's exact
    golden solution."""
    g = golden("g3_edge_cases.npz")
    ptr, ind, dat, shape = _csr(g)
    K = g["Y0"].shape[1]
    ref3 = als_ref.half_sweep_cg(ptr, ind, dat, g["Y0"], g["X0"], 0.01, 40.0, 3)
    X3, _ = _sweep_cg(ra, ptr, ind, dat, shape, g["Y0"], g["X0"], 0.01, 40.0, 3, chunk_nnz=chunk_nnz, form=form)
    assert _rel(X3, ref3) < 2e-4
    assert np.all(X3[0] == 0) and np.all(X3[-1] == 0)
    Xn, _ = _sweep_cg(ra, ptr, ind, dat, shape, g["Y0"], g["X0"], 0.01, 40.0, 3 * K, chunk_nnz=chunk_nnz, form=form)
    assert _rel(Xn, g["Xnew"]) < 1e-4


@pytest.mark.parametrize("K", [8, 20, 32, 64])
def test_cg_converges_to_golden_exact(ra, golden, K):
    """2K steps of float32 CG reproduce ALSRef's exact half-sweep (golden vectors) to 1e-4."""
    g = golden("g2_powerlaw_train.npz")
    ptr, ind, dat, shape = _csr(g)
    tag = "K%d_ref" % K
    X1, _ = _sweep_cg(ra, ptr, ind, dat, shape, g[tag + "_Y0"], g[tag + "_X0"], 0.01, 40.0, 2 * K)
    assert _rel(X1, g[tag + "_X1"]) < 1e-4


@pytest.mark.parametrize("f,form", [(96, "auto"), (192, "matrix"), (192, "auto"), (256, "matrix")])
def test_cg_row_ids_and_shards(ra, f, form):
    """row_ids route CSR rows to arbitrary factor rows (others untouched); two row shards swept
    separately equal the whole sweep bit for bit; long rows (chunk_nnz = 256) included."""
    rng = np.random.default_rng(42)
    N, M = 1500, 2000
    ptr, ind, dat = _random_csr(rng, N, M, 25, 1900)
    Y0 = rng.normal(0, 0.1, size=(M, f)).astype(np.float32); X0 = rng.normal(0, 0.1, size=(N, f)).astype(np.float32)
    whole, _ = _sweep_cg(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, 3, chunk_nnz=256, form=form)
    ref = als_ref.half_sweep_cg(ptr, ind, dat, Y0, X0, 0.01, 40.0, 3)
    assert _rel(whole, ref) < 2e-4
    cut = 611
    a, _ = _sweep_cg(ra, ptr[: cut + 1], ind[: ptr[cut]], dat[: ptr[cut]], (cut, M), Y0, X0[:cut], 0.01, 40.0, 3, chunk_nnz=256,
                     form=form)
    b, _ = _sweep_cg(ra, ptr[cut:] - ptr[cut], ind[ptr[cut]:], dat[ptr[cut]:], (N - cut, M), Y0, X0[cut:], 0.01, 40.0, 3,
                     chunk_nnz=256, form=form)
    assert np.array_equal(np.vstack([a, b]), whole)
    ids = rng.permutation(N + 7)[:N]
    Xbig = rng.normal(0, 0.1, size=(N + 7, f)).astype(np.float32)
    Xbig[ids] = X0
    out, _ = _sweep_cg(ra, ptr, ind, dat, (N, M), Y0, Xbig, 0.01, 40.0, 3, chunk_nnz=256, row_ids=ids, form=form)
    assert np.array_equal(out[ids], whole)
    rest = np.setdiff1d(np.arange(N + 7), ids)
    assert np.array_equal(out[rest], Xbig[rest].astype(np.float64))


def test_engine_cg_f256_tracks_oracle(ra):
    """Three ALS iterations at f = 256 (CG only).  Every half-sweep is checked against the oracle
    running the same algorithm on the SAME inputs (the engine's own previous factors): an
    unconverged CG iterate amplifies float32 rounding of its inputs, so free-running float32 and
    float64 trajectories drift apart by ~1e-2 within a few iterations although each step agrees to
    ~1e-5."""
    from recommender_amd.als import ALSEngine
    rng = np.random.default_rng(8)
    N, M, f, lamb, alpha, steps = 1500, 1000, 256, 0.05, 10.0, 3
    ptr, ind, dat = _random_csr(rng, N, M, 20, 900)
    tptr, tind, tdat = als_ref.csr_transpose(ptr, ind, dat, M)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X = _dev(X0); Y = _dev(Y0)
    eng = ALSEngine(ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda"), ra.DeviceCSR(tptr, tind, tdat, (M, N), "cuda"), X, Y,
                    lamb, alpha, solver="cg", cg_steps=steps)
    assert eng.solver == "cg"
    Xp, Yp = X0.astype(np.float64), Y0.astype(np.float64)
    for it in range(3):
        eng.step()
        Xg, Yg = X.cpu().numpy().astype(np.float64), Y.cpu().numpy().astype(np.float64)
        Xn = als_ref.half_sweep_cg(ptr, ind, dat, Yp, Xp, lamb, alpha, steps)
        Yn = als_ref.half_sweep_cg(tptr, tind, tdat, Xg, Yp, lamb, alpha, steps)
        assert _rel(Xg, Xn) < 2e-4 and _rel(Yg, Yn) < 2e-4, (it, _rel(Xg, Xn), _rel(Yg, Yn))
        tr = np.mean(np.abs(Xp - Xn)) + np.mean(np.abs(Yp - Yn))
        assert eng.last_delta() == pytest.approx(tr, rel=1e-3)
        Xp, Yp = Xg, Yg
    assert int(eng.info.item()) == 0
    with pytest.raises(ValueError):
        ALSEngine(eng.csr_users, eng.csr_items, X, Y, lamb, alpha, solver="cholesky")


@pytest.mark.parametrize("f,form", [(20, "auto"), (64, "matrix"), (64, "lock_step"), (127, "auto"), (200, "lock_step"),
                                    (200, "matrix")])
def test_cg_rows_of_stored_zeros_get_zero(ra, f, form):
    """Stored zeros are dropped (ALS.py:79): a row whose only entries are zeros has the exact solution
    0 and gets it in every CG form, as in the oracle - not cg_steps steps towards it; rows that mix zeros
    and ratings ignore the zeros."""
    rng = np.random.default_rng(f)
    N, M = 60, 500
    ptr, ind, dat = _random_csr(rng, N, M, 30, 300)
    lens = np.diff(ptr)
    rows0 = [r for r in np.flatnonzero(lens > 0)[:6]]                       # these rows: nothing but stored zeros
    for r in rows0:
        dat[ptr[r]:ptr[r + 1]] = 0.0
    dat[rng.integers(0, len(dat), size=len(dat) // 10)] = 0.0                # and zeros sprinkled over the others
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = als_ref.half_sweep_cg(ptr, ind, dat.astype(np.float64), Y0.astype(np.float64), X0.astype(np.float64), 0.1, 10.0, 3)
    X, _ = _sweep_cg(ra, ptr, ind, dat, (N, M), Y0, X0, 0.1, 10.0, 3, form=form)
    assert np.all(X[rows0] == 0.0) and np.all(ref[rows0] == 0.0)
    assert _rel(X, ref) < 2e-4
