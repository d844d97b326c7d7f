"""The persistent stream kernel for 64 < f <= 128 (row_kernel="stream") against the one-wave kernel and the oracle.

Both kernels cut, multiply and solve in the same order, so their factors and |delta| must be the same bits; the
oracle comparison (float64 restatement of reclibwh/core/ALS.py:70-118) bounds both.  Shapes cover rows longer than
chunk_nnz (few chunk items stay on the one-wave kernel, many go through the stream kernel too), empty rows, fewer rows than resident waves, and f < 128
(masked row halves).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _csr(rng, N, M, avg, maxlen):
    lens = np.minimum(rng.zipf(1.6, size=N) + rng.integers(0, 2 * avg, size=N), maxlen).astype(np.int64)
    lens[rng.integers(0, N, size=max(N // 50, 1))] = 0
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    return ptr, ind, dat


@pytest.mark.parametrize("f", [128, 96, 100, 68])
@pytest.mark.parametrize("shape", [(3000, 5000, 2048), (1200, 2500, 128), (5, 400, 2048), (700, 300, 64)])
def test_stream_equals_wave_and_oracle(f, shape):
    import recommender_amd as ra
    from oracle import c_oracle

    N, M, chunk = shape
    rng = np.random.default_rng(1000 * f + N)
    ptr, ind, dat = _csr(rng, N, M, 40, M - 1)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, 0.01, 40.0, dtype=np.float64, nthreads=8)
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=chunk)
    Y = torch.from_numpy(Y0).cuda()
    gc = ra.gram(Y, 0.01)
    out = {}
    for rk in ("wave", "stream"):
        for cg in (0, 3):
            res = []
            for _ in range(2):
                X = torch.from_numpy(X0).cuda()
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                o = ra.ops.sweep_opts(row_kernel=rk)
                if cg:
                    d = ra.ops.half_sweep_cg(csr, Y, gc, X, 40.0, cg, info=info, form="matrix", opts=o)
                else:
                    d = ra.half_sweep(csr, Y, gc, X, 40.0, info=info, opts=o)
                torch.cuda.synchronize()
                assert int(info.item()) == 0
                res.append((X.cpu().numpy(), float(d.item())))
            assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1], "two runs differ"
            out[(rk, cg)] = res[0]
    scale = np.abs(ref).max()
    assert np.abs(out[("stream", 0)][0] - ref).max() / scale < 1e-4      # the tolerance the factors are held to
    for cg in (0, 3):
        assert np.array_equal(out[("stream", cg)][0], out[("wave", cg)][0])
        assert out[("stream", cg)][1] == out[("wave", cg)][1]


@pytest.mark.parametrize("f", [128, 100])
def test_stream_takes_the_chunk_items_when_there_are_many(f):
    """With several chunk items per resident wavefront (here ~5 000 of 64 entries each, from 150 rows of about 2 100
    entries) the long rows' chunks go through the stream kernel as well, which stores their partial sums instead of
    solving: the same bits as the one-wave kernel for factors and |delta|, exact solve and conjugate gradients."""
    import recommender_amd as ra
    from oracle import c_oracle

    rng = np.random.default_rng(77 + f)
    N, M, chunk = 1300, 6000, 64
    lens = rng.integers(0, 60, size=N).astype(np.int64)
    long_rows = rng.choice(N, size=150, replace=False)
    lens[long_rows] = rng.integers(2040, 2120, size=150)
    lens[long_rows[:4]] = [2048, 2049, 2047, 2080]            # whole chunks, one entry over, one under, a half block over
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, 0.01, 40.0, dtype=np.float64, nthreads=8)
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=chunk)
    assert csr.n_items - (N - csr.n_long) >= 4 * 4 * torch.cuda.get_device_properties(0).multi_processor_count
    Y = torch.from_numpy(Y0).cuda()
    gc = ra.gram(Y, 0.01)
    out = {}
    for rk in ("wave", "stream"):
        for cg in (0, 3):
            X = torch.from_numpy(X0).cuda()
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            o = ra.ops.sweep_opts(row_kernel=rk)
            if cg:
                d = ra.ops.half_sweep_cg(csr, Y, gc, X, 40.0, cg, info=info, form="matrix", opts=o)
            else:
                d = ra.half_sweep(csr, Y, gc, X, 40.0, info=info, opts=o)
            torch.cuda.synchronize()
            assert int(info.item()) == 0
            out[(rk, cg)] = (X.cpu().numpy(), float(d.item()))
    assert np.abs(out[("stream", 0)][0] - ref).max() / np.abs(ref).max() < 1e-4
    for cg in (0, 3):
        assert np.array_equal(out[("stream", cg)][0], out[("wave", cg)][0])
        assert out[("stream", cg)][1] == out[("wave", cg)][1]


def test_stream_block_boundaries_padded_rows_and_row_ids():
    """Row lengths at the 32-entry block boundaries (0, 1, 31, 32, 33, 63, 64, 65, ...), a matrix without any entry,
    factor tables with a padded leading dimension, and rows written through row_ids (the fold-in's mapping): the
    stream kernel returns the one-wave kernel's bits every time."""
    import recommender_amd as ra

    rng = np.random.default_rng(77)
    f, M, ld = 128, 900, 160
    lens = np.array([0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 0, 0, 200, 2, 32, 64] * 60, dtype=np.int64)
    N = len(lens)
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(0, 11, size=ptr[-1]) * 0.5).astype(np.float32)          # stored zeros included
    Ys = torch.from_numpy(rng.normal(0, 0.1, size=(M, ld)).astype(np.float32)).cuda()
    Y = Ys[:, :f]
    gc = ra.gram(Y, 0.01)
    X0 = rng.normal(0, 0.1, size=(3 * N, ld)).astype(np.float32)
    row_ids = rng.permutation(3 * N)[:N].astype(np.int64)
    cases = {
        "plain": ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda"),
        "row_ids": ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", row_ids=row_ids),
        "empty": ra.DeviceCSR(np.zeros(N + 1, dtype=np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), (N, M), "cuda"),
    }
    for name, csr in cases.items():
        for cg in (0, 3):
            got = {}
            for rk in ("wave", "stream"):
                X = torch.from_numpy(X0).cuda()[:, :f]
                o = ra.ops.sweep_opts(row_kernel=rk)
                if cg:
                    d = ra.ops.half_sweep_cg(csr, Y, gc, X, 40.0, cg, form="matrix", opts=o)
                else:
                    d = ra.half_sweep(csr, Y, gc, X, 40.0, opts=o)
                got[rk] = (X.cpu().numpy(), float(d.item()))
            assert np.array_equal(got["wave"][0], got["stream"][0]), (name, cg)
            assert got["wave"][1] == got["stream"][1], (name, cg)
            if name == "empty":
                assert not got["stream"][0][:N].any()


def test_stream_rejects_what_it_cannot_run():
    import recommender_amd as ra

    rng = np.random.default_rng(3)
    ptr, ind, dat = _csr(rng, 200, 300, 10, 100)
    for f in (64, 20):
        Y = torch.from_numpy(rng.normal(size=(300, f)).astype(np.float32)).cuda()
        X = torch.zeros(200, f, device="cuda")
        csr = ra.DeviceCSR(ptr, ind, dat, (200, 300), "cuda")
        with pytest.raises(ValueError, match="STREAM"):
            ra.half_sweep(csr, Y, ra.gram(Y, 0.1), X, 40.0, opts=ra.ops.sweep_opts(row_kernel="stream"))
    f = 128
    Y = torch.from_numpy(rng.normal(size=(300, f)).astype(np.float32)).cuda()
    X = torch.zeros(200, f, device="cuda")
    csr = ra.DeviceCSR(ptr, ind, dat, (200, 300), "cuda")
    with pytest.raises(ValueError, match="STREAM"):
        ra.half_sweep(csr, Y, ra.gram(Y, 0.1), X, 40.0, opts=ra.ops.sweep_opts(row_kernel="stream", accumulate="fp32"))


@pytest.mark.parametrize("f", [64, 128])
def test_a_non_finite_row_nobody_references_changes_nothing(f):
    """Lanes past a row's (or a 32-entry block's) end gather SOME factor row with weight 0.  That row must be one the
    CSR row references itself: with factor row 0 set to NaN and no CSR row touching item 0, every kernel (one wave per
    row, the stream kernel, the short-row form) must return what it returns for a finite row 0, bit for bit."""
    import recommender_amd as ra
    rng = np.random.default_rng(31 + f)
    n_rows, n_cols, lamb, alpha = 700, 3000, 0.01, 40.0
    lens = rng.integers(0, 200, size=n_rows); lens[:3] = [1, 33, 2500]
    ptr = np.zeros(n_rows + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    col = np.concatenate([np.sort(rng.choice(np.arange(1, n_cols), size=int(L), replace=False)) for L in lens]).astype(np.int32)
    val = (rng.integers(1, 11, size=col.shape[0]) * 0.5).astype(np.float32)
    Y = rng.normal(0, 1 / np.sqrt(f), size=(n_cols, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(n_rows, f)).astype(np.float32)
    csr = ra.DeviceCSR(ptr, col, val, (n_rows, n_cols), "cuda")

    def run(y0, **kw):
        Yd = torch.from_numpy(Y).cuda(); Yd[0] = y0
        Xd = torch.from_numpy(X0).cuda()
        gc = ra.gram(Yd[1:], lamb)                   # the Gram of the rows that exist (row 0 is nobody's)
        short = kw.pop("short", False)
        if short:
            ra.whiten(Yd[1:], gc)
            gc.whitened = torch.cat([torch.full_like(gc.whitened[:1], float("nan")), gc.whitened], 0)
        ra.half_sweep(csr, Yd, gc, Xd, alpha, opts=ra.ops.sweep_opts(**kw) if kw else None, short_rows=short)
        torch.cuda.synchronize()
        return Xd.cpu().numpy()

    kinds = [dict(row_kernel="wave")] + ([dict(row_kernel="stream"), dict(short=True)] if f == 128 else [])
    for kw in kinds:
        finite, nan = run(0.25, **dict(kw)), run(float("nan"), **dict(kw))
        assert np.all(np.isfinite(nan)), kw
        assert np.array_equal(finite, nan), kw


@pytest.mark.parametrize("f", [128, 100])
def test_reserved_compute_units_change_nothing_but_the_grid(f):
    """als_sweep_opts_t.reserve_cus: the stream kernel leaves compute units free for other streams (the RCCL exchange
    of a multi-GPU run).  Rows are shared out by the grid size and solved independently: with 16 units reserved, and
    with all but one (four resident wavefronts walk every row), factors and |delta| are the bits of the full grid -
    whole rows and the long rows' chunk items alike."""
    import recommender_amd as ra

    rng = np.random.default_rng(31 + f)
    N, M, chunk = 2600, 5000, 64
    lens = rng.integers(0, 90, size=N).astype(np.int64)
    long_rows = rng.choice(N, size=160, replace=False)
    lens[long_rows] = rng.integers(2000, 2100, size=160)          # ~5 000 chunk items: they take the stream kernel too
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    Y = torch.from_numpy(rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)).cuda()
    X0 = torch.from_numpy(rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)).cuda()
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=chunk)
    gc = ra.gram(Y, 0.01)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    out = []
    for reserve in (0, 16, cus - 1, 10 * cus):
        X = X0.clone()
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        d = ra.half_sweep(csr, Y, gc, X, 40.0, info=info, opts=ra.ops.sweep_opts(row_kernel="stream", reserve_cus=reserve))
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        out.append((X, float(d.item())))
    for X, d in out[1:]:
        assert torch.equal(X, out[0][0]) and d == out[0][1]
    with pytest.raises(ValueError):
        ra.ops.sweep_opts(reserve_cus=-1)
