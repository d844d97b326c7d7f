"""GPU tests of the short-row ("entry space") form of the exact half-sweep (csrc/als_short_rows.hip).

Rows with fewer stored entries than factors are solved as an L x L system on the whitened factor table instead of
the f x f one.  Algebraically the same x as ALSRef._run_single_step (reclibwh/core/ALS.py:67-88); the bar is the
same as for the f x f kernels: 1e-4 relative against the float64 oracle.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle  # noqa: E402

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


def _ragged(rng, n_rows, n_cols, lengths, zeros=0.0):
    """CSR with the given row lengths, sorted distinct columns, ratings in {0.5, ..., 5}; a share `zeros` of the
    entries is a stored zero."""
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(lengths)
    col = np.concatenate([np.sort(rng.choice(n_cols, size=int(L), replace=False)) for L in lengths] + [np.zeros(0, np.int64)])
    val = rng.integers(1, 11, size=col.shape[0]).astype(np.float32) * 0.5
    if zeros:
        val[rng.random(col.shape[0]) < zeros] = 0.0
    return ptr, col.astype(np.int32), val


def _run(ra, ptr, col, val, shape, Y, X0, lamb, alpha, short, chunk_nnz=2048, row_ids=None):
    csr = ra.DeviceCSR(ptr, col, val, shape, "cuda", chunk_nnz=chunk_nnz, row_ids=row_ids)
    Yd, Xd = _dev(Y), _dev(X0)
    gc = ra.gram(Yd, lamb)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    if short:
        ra.whiten(Yd, gc, info=info)
    d = ra.half_sweep(csr, Yd, gc, Xd, alpha, info=info, short_rows=short)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return Xd.cpu().numpy().astype(np.float64), float(d.item())


@pytest.mark.parametrize("f", [36, 64, 100, 128])
def test_whitened_table_matches_float64(ra, f):
    rng = np.random.default_rng(f)
    n, lamb = 5000, 0.01
    Y = (rng.normal(0, 1 / np.sqrt(f), size=(n, f)) * rng.uniform(0.3, 3.0, size=f)).astype(np.float32)
    Yd = _dev(Y)
    gc = ra.gram(Yd, lamb)
    wz = ra.whiten(Yd, gc).cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    G = gc.gram.cpu().numpy().astype(np.float64) + lamb * np.eye(f)
    R = np.linalg.cholesky(G).T
    W = Y.astype(np.float64) @ np.linalg.inv(R)
    Z = Y.astype(np.float64) @ np.linalg.inv(G)
    assert _rel(wz[:, :f], W) < 2e-6
    assert _rel(wz[:, f:], Z) < 2e-6


@pytest.mark.parametrize("f", [36, 64, 100, 128])
def test_every_row_length_against_the_float64_oracle(ra, f):
    """Rows of every length from 0 to past the short-row limit (so both forms run in one call), stored zeros
    included: each within 1e-4 of the float64 solve, the reported change to 1e-6."""
    rng = np.random.default_rng(100 + f)
    n_cols, lamb, alpha = 4000, 0.01, 40.0
    lengths = np.concatenate([np.arange(0, 150), rng.integers(0, 140, size=400), [300, 2500]])
    rng.shuffle(lengths)
    n_rows = lengths.shape[0]
    ptr, col, val = _ragged(rng, n_rows, n_cols, lengths, zeros=0.03)
    Y = rng.normal(0, 1 / np.sqrt(f), size=(n_cols, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(n_rows, f)).astype(np.float32)
    want = c_oracle.half_sweep(ptr, col, val.astype(np.float64), Y.astype(np.float64), X0.astype(np.float64), lamb, alpha)
    got, d = _run(ra, ptr, col, val, (n_rows, n_cols), Y, X0, lamb, alpha, True)
    assert _rel(got, want) < 1e-4, _rel(got, want)
    per_row = np.abs(got - want).max(axis=1) / np.abs(want).max()
    assert per_row.max() < 1e-4
    assert abs(d - np.abs(X0.astype(np.float64) - want).sum()) < 1e-5 * np.abs(X0 - want).sum()
    plain, _ = _run(ra, ptr, col, val, (n_rows, n_cols), Y, X0, lamb, alpha, False)
    assert _rel(got, plain) < 5e-5
    lim = ra.short_rows_max_entries(f)
    longer = lengths > lim
    assert np.array_equal(got[longer], plain[longer])           # rows past the limit take the f x f form: same bits


def test_repeat_runs_shards_and_row_ids_agree_bit_for_bit(ra):
    rng = np.random.default_rng(7)
    f, n_rows, n_cols, lamb, alpha = 128, 3000, 6000, 0.01, 40.0
    lengths = np.minimum(np.maximum(rng.lognormal(3.5, 1.0, size=n_rows).astype(np.int64), 0), 900)
    ptr, col, val = _ragged(rng, n_rows, n_cols, lengths)
    Y = rng.normal(0, 1 / np.sqrt(f), size=(n_cols, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(n_rows, f)).astype(np.float32)
    a, da = _run(ra, ptr, col, val, (n_rows, n_cols), Y, X0, lamb, alpha, True)
    b, db = _run(ra, ptr, col, val, (n_rows, n_cols), Y, X0, lamb, alpha, True)
    assert np.array_equal(a, b) and da == db
    # two row shards = the whole
    cut = 1234
    lo = _run(ra, ptr[:cut + 1], col[:ptr[cut]], val[:ptr[cut]], (cut, n_cols), Y, X0[:cut], lamb, alpha, True)[0]
    hi = _run(ra, ptr[cut:] - ptr[cut], col[ptr[cut]:], val[ptr[cut]:], (n_rows - cut, n_cols), Y, X0[cut:], lamb, alpha, True)[0]
    assert np.array_equal(np.concatenate([lo, hi]), a)
    # a permuted row_ids map writes the same rows elsewhere
    perm = rng.permutation(n_rows)
    Xp = np.zeros_like(X0); Xp[perm] = X0
    c, _ = _run(ra, ptr, col, val, (n_rows, n_cols), Y, Xp, lamb, alpha, True, row_ids=perm)
    assert np.array_equal(c[perm], a)


def test_trained_factors_hold_the_bar(ra):
    """Factors after three exact iterations (correlated columns, a spread of scales): both forms against float64."""
    from recommender_amd import data
    f, N, M, lamb, alpha = 128, 6000, 3000, 0.01, 40.0
    U, Ut = data.synth_matrix(N, M, 300000, seed=11, device="cpu")
    p, c, v = (t.numpy() for t in U); pt, ct, vt = (t.numpy() for t in Ut)
    rng = np.random.default_rng(3)
    X = rng.normal(0, 1 / np.sqrt(f), size=(N, f)); Y = rng.normal(0, 1 / np.sqrt(f), size=(M, f))
    for _ in range(3):
        X = c_oracle.half_sweep(p, c, v.astype(np.float64), Y, X, lamb, alpha, nthreads=8)
        Y = c_oracle.half_sweep(pt, ct, vt.astype(np.float64), X, Y, lamb, alpha, nthreads=8)
    Y32 = Y.astype(np.float32); X32 = X.astype(np.float32)
    want = c_oracle.half_sweep(p, c, v.astype(np.float64), Y32.astype(np.float64), X32.astype(np.float64), lamb, alpha, nthreads=8)
    got, _ = _run(ra, p, c, v, (N, M), Y32, X32, lamb, alpha, True)
    plain, _ = _run(ra, p, c, v, (N, M), Y32, X32, lamb, alpha, False)
    e_short, e_plain = _rel(got, want), _rel(plain, want)
    assert e_short < 1e-4 and e_plain < 1e-4, (e_short, e_plain)
