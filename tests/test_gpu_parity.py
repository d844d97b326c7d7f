"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors minted from the reference's ALSRef.

Tolerances.  The HIP path computes in float32 (as the reference's TensorFlow path does,
reclibwh/core/ALS.py:160-199), the oracle in float64.  The reference's own acceptance test
(tests/core/test_ALS.py:53) is np.allclose(float32 path, float64 ALSRef), i.e. rtol 1e-5 /
atol 1e-8, at K=5, alpha=1, lamb=1e-2; the north-star bar is 1e-4 relative on the factors.
Each test states the bound it uses.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import als_ref, c_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd
    return recommender_amd


def _dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def _csr(g, p="U"):
    return g[p + "_indptr"], g[p + "_indices"], g[p + "_data"], tuple(int(x) for x in g[p + "_shape"])


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _sweep(ra, ptr, ind, dat, shape, Yfix, Xold, lamb, alpha, chunk_nnz=2048, row_ids=None, opts=None):
    csr = ra.DeviceCSR(ptr, ind, dat, shape, "cuda", chunk_nnz=chunk_nnz, row_ids=row_ids)
    Y = _dev(Yfix); X = _dev(Xold)
    gc = ra.gram(Y, lamb)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = ra.half_sweep(csr, Y, gc, X, alpha, info=info, opts=opts)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return X.cpu().numpy().astype(np.float64), float(d.item())


# ----------------------------------------------------------------------------- gram
@pytest.mark.parametrize("f", [5, 8, 16, 20, 32, 50, 64, 100, 128])
@pytest.mark.parametrize("n", [1, 3, 100, 1023, 1025, 70001])
def test_gram_matches_numpy(ra, f, n):
    rng = np.random.default_rng(f * 1000 + n)
    Y = rng.normal(0, 1 / np.sqrt(f), size=(n, f)).astype(np.float32)
    gc = ra.gram(_dev(Y), 0.01)
    torch.cuda.synchronize()
    G = gc.gram.cpu().numpy()
    ref = Y.astype(np.float64).T @ Y.astype(np.float64)
    # float32 accumulation over n rows: error ~ sqrt(n) * eps * |sum|
    assert np.abs(G - ref).max() <= 3e-6 * np.sqrt(n) * max(np.abs(ref).max(), 1.0)
    assert np.array_equal(G, G.T)                    # exactly symmetric


def test_gram_partials_equal_single_call(ra):
    """Gram assembled from row shards (the multi-device path) == single call, bit for bit."""
    rng = np.random.default_rng(5)
    n, f = 3 * 65536 + 777, 64
    Y = _dev(rng.normal(0, 0.2, size=(n, f)).astype(np.float32))
    whole = ra.gram(Y, 0.5)
    parts = [ra.gram_partials(Y, 0, 65536), ra.gram_partials(Y, 65536, 3 * 65536), ra.gram_partials(Y, 3 * 65536, n)]
    sharded = ra.gram_finalize(torch.cat(parts, 0).contiguous(), f, 0.5)
    torch.cuda.synchronize()
    assert torch.equal(whole.gram, sharded.gram) and torch.equal(whole.tiles, sharded.tiles)


# ----------------------------------------------------------------------------- golden vectors
def test_g1_reference_unit_test(ra, golden):
    """Replica of tests/core/test_ALS.py:18-53 with the HIP kernel in place of run_als_step;
    same assertion: np.allclose (rtol 1e-5, atol 1e-8) against ALSRef._run_single_step."""
    g = golden("g1_single_step.npz")
    ptr, ind, dat, shape = _csr(g)
    X, d = _sweep(ra, ptr, ind, dat, shape, g["Yinit"], g["Xinit"], float(g["lamb"]), float(g["alpha"]))
    # With 3 items and K=5 the Gram is rank deficient (cond(A) ~ 5e2): on this seeded draw the
    # float32 restatement of the TensorFlow path itself is 1.5e-5 away from ALSRef, so the bar is
    # allclose OR at least as close as that restatement.
    Ylayer = np.vstack([g["Yinit"], np.zeros((1, g["Yinit"].shape[1]))])
    Xtf, _ = als_ref.half_sweep_batched_f32(ptr, ind, dat, shape[1], Ylayer, g["Xinit"], float(g["lamb"]),
                                            float(g["alpha"]), batchsize=shape[0])
    err = np.abs(X - g["Xnew"])
    bar = np.abs(Xtf - g["Xnew"]).max()
    assert np.all(err <= np.maximum(1e-8 + 1e-5 * np.abs(g["Xnew"]), bar)), \
        "max |X - ALSRef| = %.3e (float32 restatement of the TF path: %.3e; np.allclose alone would allow %.3e at that entry)" \
        % (err.max(), bar, 1e-8 + 1e-5 * np.abs(g["Xnew"]).flat[err.argmax()])
    assert d / X.size == pytest.approx(np.mean(np.abs(g["Xinit"].astype(np.float32) - g["Xnew"])), rel=1e-4)


@pytest.mark.parametrize("K", [8, 20, 32, 64])
def test_g2_first_iteration(ra, golden, K):
    """One X half-sweep and one Y half-sweep from the seed-42 init; factors within 1e-4
    relative (max-abs over the matrix / max-abs of the reference)."""
    g = golden("g2_powerlaw_train.npz")
    ptr, ind, dat, shape = _csr(g)
    tag = "K%d_ref" % K
    X1, _ = _sweep(ra, ptr, ind, dat, shape, g[tag + "_Y0"], g[tag + "_X0"], 0.01, 40.0)
    assert _rel(X1, g[tag + "_X1"]) < 1e-4
    tptr, tind, tdat = als_ref.csr_transpose(ptr, ind, dat, shape[1])
    Y1, _ = _sweep(ra, tptr, tind, tdat, (shape[1], shape[0]), g[tag + "_X1"], g[tag + "_Y0"], 0.01, 40.0)
    assert _rel(Y1, g[tag + "_Y1"]) < 1e-4


@pytest.mark.parametrize("chunk_nnz", [64, 128, 2048])
def test_g3_edge_cases(ra, golden, chunk_nnz):
    """Empty rows, single-entry rows, one long row (split into chunks when chunk_nnz is small),
    an empty column."""
    g = golden("g3_edge_cases.npz")
    ptr, ind, dat, shape = _csr(g)
    X, _ = _sweep(ra, ptr, ind, dat, shape, g["Y0"], g["X0"], 0.01, 40.0, chunk_nnz=chunk_nnz)
    assert _rel(X, g["Xnew"]) < 1e-4
    assert np.all(X[0] == 0) and np.all(X[-1] == 0)
    tptr, tind, tdat = als_ref.csr_transpose(ptr, ind, dat, shape[1])
    Y, _ = _sweep(ra, tptr, tind, tdat, (shape[1], shape[0]), g["Xnew"], g["Y0"], 0.01, 40.0, chunk_nnz=chunk_nnz)
    # 12 user rows < K=16: the Gram of X is rank deficient and the systems are ill conditioned
    # (the reference notes the same effect, tests/core/test_ALS.py:27-29); float32 cannot reach
    # 1e-4 here, so the bar is the error of the float32 build of the oracle itself (x2).
    Yf32 = c_oracle.half_sweep(tptr, tind, tdat, g["Xnew"], g["Y0"], 0.01, 40.0, dtype=np.float32)
    assert _rel(Y, g["Ynew"]) < max(1e-4, 2 * _rel(Yf32, g["Ynew"])), \
        "relative error %.3e (float32 build of the oracle: %.3e; bar 1e-4 or twice that)" % (_rel(Y, g["Ynew"]), _rel(Yf32, g["Ynew"]))
    assert np.all(Y[7] == 0)


def test_g5_foldin_rows_identical(ra, golden):
    """Two rows with identical entries get bitwise identical factors (test_ALS.py:88-101), and
    row_ids route CSR rows to arbitrary factor rows leaving the others untouched."""
    g = golden("g5_foldin.npz")
    ptr, ind, dat, shape = _csr(g)
    X0 = np.vstack([g["X0"], g["X0"], g["X0"]])           # 6 rows; update rows 4 and 1 only
    X, _ = _sweep(ra, ptr, ind, dat, shape, g["Y0"], X0, 1e-2, 40.0, row_ids=[4, 1])
    assert np.array_equal(X[4], X[1])
    assert np.array_equal(X[[0, 2, 3, 5]], X0.astype(np.float32)[[0, 2, 3, 5]].astype(np.float64))
    ref = als_ref.half_sweep_exact(ptr, ind, dat, g["Y0"], g["X0"], 1e-2, 40.0)
    assert _rel(X[4], ref[0]) < 1e-4


# ----------------------------------------------------------------------------- larger, vs the C oracle
def _random_csr(rng, N, M, avg, maxlen):
    lens = np.minimum(rng.zipf(1.6, size=N) + rng.integers(0, 2 * avg, size=N), maxlen).astype(np.int64)
    lens[rng.integers(0, N, size=max(N // 50, 1))] = 0
    ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    return ptr, ind, dat


@pytest.mark.parametrize("f", [20, 50, 64, 96, 128])
def test_random_matrix_vs_c_oracle(ra, f):
    rng = np.random.default_rng(77 + f)
    N, M = 3000, 5000
    ptr, ind, dat = _random_csr(rng, N, M, 40, 4500)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, 0.01, 40.0, dtype=np.float64, nthreads=8)
    X, d = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0)
    assert _rel(X, ref) < 1e-4
    assert d == pytest.approx(np.abs(X0.astype(np.float64) - ref).sum(), rel=1e-4)
    # bit-for-bit reproducible, and independent of how long rows are scheduled relative to others
    X2, d2 = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0)
    assert np.array_equal(X, X2) and d == d2


def test_row_shards_equal_whole(ra):
    """Sweeping two row shards separately (what two devices do) gives the same bits as one call."""
    rng = np.random.default_rng(99)
    N, M, f = 2000, 3000, 64
    ptr, ind, dat = _random_csr(rng, N, M, 30, 2900)
    Y0 = rng.normal(0, 0.2, size=(M, f)).astype(np.float32); X0 = rng.normal(0, 0.2, size=(N, f)).astype(np.float32)
    whole, _ = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0)
    cut = 777
    p1 = ptr[: cut + 1]; p2 = ptr[cut:] - ptr[cut]
    a, _ = _sweep(ra, p1, ind[: ptr[cut]], dat[: ptr[cut]], (cut, M), Y0, X0[:cut], 0.01, 40.0)
    b, _ = _sweep(ra, p2, ind[ptr[cut]:], dat[ptr[cut]:], (N - cut, M), Y0, X0[cut:], 0.01, 40.0)
    assert np.array_equal(np.vstack([a, b]), whole)


def test_row_chunks_with_smaller_ratings_accumulate_at_the_whole_matrix_scale(ra):
    """The fp16 x 2 accumulate takes its operand scale from a bound of the ratings.  A chunk of rows whose own largest
    rating is far below the matrix's (here 0.5 against 900) must still use the matrix's bound - row_range hands it down -
    so that the chunk's rows come out bit-identical to the same rows of one call."""
    rng = np.random.default_rng(17)
    N, M, f = 1500, 4000, 64
    ptr, ind, dat = _random_csr(rng, N, M, 30, 2900)
    cut = 700
    dat = dat.copy(); dat[: ptr[cut]] = 0.5; dat[ptr[cut]:] = rng.choice([1.0, 30.0, 900.0], size=int(ptr[-1] - ptr[cut]))
    Y = _dev(rng.normal(0, 0.2, size=(M, f))); X0 = rng.normal(0, 0.2, size=(N, f)).astype(np.float32)
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda")
    assert csr.max_rating == 900.0 and csr.row_range(0, cut).max_rating == 900.0
    gc = ra.gram(Y, 0.01)
    whole = _dev(X0); ra.half_sweep(csr, Y, gc, whole, 0.1)
    parts = _dev(X0)
    ra.half_sweep(csr.row_range(0, cut), Y, gc, parts[:cut], 0.1)
    ra.half_sweep(csr.row_range(cut, N), Y, gc, parts[cut:], 0.1)
    assert torch.equal(parts, whole)


def test_invalid_arguments_raise(ra):
    rng = np.random.default_rng(1)
    ptr = np.array([0, 2, 3], dtype=np.int64); ind = np.array([0, 1, 5], dtype=np.int32)
    dat = np.ones(3, dtype=np.float32)
    with pytest.raises(ValueError):
        ra.DeviceCSR(ptr, ind, dat, (2, 4), "cuda")          # column 5 out of range
    with pytest.raises(ValueError):
        ra.DeviceCSR(ptr, ind, np.array([1.0, -0.5, 2.0], dtype=np.float32), (2, 6), "cuda")   # negative rating
    csr = ra.DeviceCSR(ptr, ind, dat, (2, 6), "cuda")
    Y = _dev(rng.normal(size=(6, 8))); X = _dev(rng.normal(size=(2, 8)))
    gc = ra.gram(Y, 0.1)
    with pytest.raises(ValueError):
        ra.half_sweep(csr, Y, gc, _dev(rng.normal(size=(2, 4))), 1.0)      # factor count mismatch
    with pytest.raises(ValueError):
        ra.half_sweep(csr, Y[:4], gc, X, 1.0)                              # too few opposite rows
    with pytest.raises(ValueError):
        ra.gram(_dev(rng.normal(size=(4, 300))), 0.1)                      # f beyond the supported range
    with pytest.raises(ValueError):
        ra.half_sweep(ra.DeviceCSR(ptr, ind, dat, (2, 6), "cuda"), _dev(rng.normal(size=(6, 200))), ra.gram(_dev(rng.normal(size=(6, 200))), 0.1),
                      _dev(rng.normal(size=(2, 200))), 1.0)                   # exact solve ends at f = 128


@pytest.mark.parametrize("f", [20, 64, 100, 128])
def test_accumulate_modes_agree_and_both_hold_parity(ra, f):
    """The fp16 x 2 accumulate (the default when the CSR's largest rating is known and the fixed factor has at least 16
    rows per factor), the bf16 x 3 accumulate and the float32 fma-chain accumulate are three roundings of the same
    sums: each within 1e-4 of the float64 oracle, within 2e-5 of each other, and each bit-for-bit reproducible.  Long
    rows (chunk_nnz = 128) and ragged 32-entry blocks included.  A deliberately loose rating bound (1000 times the
    largest rating: 5 of fp16's bits given away) must not cost the bar either."""
    rng = np.random.default_rng(300 + f)
    N, M = 1200, 2500
    ptr, ind, dat = _random_csr(rng, N, M, 45, 2400)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, 0.01, 40.0, dtype=np.float64, nthreads=8)
    out = {}
    for mode in ("fp32", "bf16x3", "fp16x2"):       # a per-call option: nothing to restore afterwards
        o = ra.ops.sweep_opts(accumulate=mode)
        a, _ = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, chunk_nnz=128, opts=o)
        b, _ = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, chunk_nnz=128, opts=o)
        assert np.array_equal(a, b)
        assert _rel(a, ref) < 1e-4, (mode, _rel(a, ref))
        out[mode] = a
    assert _rel(out["bf16x3"], out["fp32"]) < 2e-5 and _rel(out["fp16x2"], out["fp32"]) < 2e-5
    dflt, _ = _sweep(ra, ptr, ind, dat, (N, M), Y0, X0, 0.01, 40.0, chunk_nnz=128)
    # no options = the documented default: fp16 x 2 where the kernels have a split form (f > 16) and M >= 16 f
    assert np.array_equal(dflt, out["fp16x2" if M >= 16 * f else "bf16x3"])
    if f > 16:
        csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=128)
        csr.max_rating *= 1000.0
        Y = _dev(Y0); X = _dev(X0)
        ra.half_sweep(csr, Y, ra.gram(Y, 0.01), X, 40.0, opts=ra.ops.sweep_opts(accumulate="fp16x2"))
        torch.cuda.synchronize()
        assert _rel(X.cpu().numpy().astype(np.float64), ref) < 1e-4


def test_integration_stub_from_the_docs_runs(ra):
    """The ctypes binding printed in the integration guide at the repository root (operator level: plan, Gram cache, half-sweep),
    executed as printed, reproduces the oracle's half-sweep."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def run_epoch" in b)
    stub = stub.replace('"recommender_amd/libals_hip.so"', repr(os.path.join(root, "recommender_amd", "libals_hip.so")))
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    lib = ns["lib"]
    rng = np.random.default_rng(12)
    N, M, f, lamb, alpha = 400, 3000, 20, 0.01, 40.0
    ptr, ind, dat = _random_csr(rng, N, M, 30, 2500)         # one row longer than a chunk
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    plan = ns["make_plan"](np.ascontiguousarray(ptr, dtype=np.int64))
    Y, X = _dev(Y0), _dev(X0)
    gram = torch.empty((f, f), device="cuda")
    tiles = torch.empty(lib.als_gram_tiles_bytes(f), dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(lib.als_gram_workspace_bytes(M, f), 256), dtype=torch.uint8, device="cuda")
    ns["update_cache"](Y, lamb, gram, tiles, ws)
    ws2 = torch.empty(max(lib.als_half_sweep_workspace_bytes(N, plan[3], f), 256), dtype=torch.uint8, device="cuda")
    absdiff = torch.zeros(1, dtype=torch.float64, device="cuda")
    mean_change = ns["run_epoch"](_dev(ptr, torch.int64), _dev(ind, torch.int32), _dev(dat), plan, Y, tiles, X, alpha,
                                  absdiff, ws2)
    torch.cuda.synchronize()
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, lamb, alpha, dtype=np.float64, nthreads=4)
    assert _rel(X.cpu().numpy().astype(np.float64), ref) < 1e-4
    assert float(mean_change.item()) == pytest.approx(np.abs(X0.astype(np.float64) - ref).mean(), rel=1e-4)

    # the same binding with short rows in their entry-space form (f = 128) and the rating bound
    N, M, f = 600, 4000, 128
    ptr, ind, dat = _random_csr(rng, N, M, 40, 2500)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    hp = np.ascontiguousarray(ptr, dtype=np.int64)
    plan = ns["make_plan"](hp)
    bounds = ns["make_short_bounds"](hp)
    Y, X = _dev(Y0), _dev(X0)
    gram = torch.empty((f, f), device="cuda")
    tiles = torch.empty(lib.als_gram_tiles_bytes(f), dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(lib.als_gram_workspace_bytes(M, f), 256), dtype=torch.uint8, device="cuda")
    ns["update_cache"](Y, lamb, gram, tiles, ws)
    WZ = torch.empty((M, 2 * f), device="cuda")
    wws = torch.empty(lib.als_whiten_workspace_bytes(f), dtype=torch.uint8, device="cuda")
    ns["whiten"](Y, lamb, gram, WZ, wws)
    val = _dev(dat)
    opts = ns["short_row_opts"](val, WZ, bounds)
    ws2 = torch.empty(max(lib.als_half_sweep_workspace_bytes(N, plan[3], f), 256), dtype=torch.uint8, device="cuda")
    absdiff.zero_()
    ns["run_epoch"](_dev(ptr, torch.int64), _dev(ind, torch.int32), val, plan, Y, tiles, X, alpha, absdiff, ws2, opts)
    torch.cuda.synchronize()
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, lamb, alpha, dtype=np.float64, nthreads=4)
    assert _rel(X.cpu().numpy().astype(np.float64), ref) < 1e-4
