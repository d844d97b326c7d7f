"""als_half_sweep_csr_static_f32 - the half-sweep for callers that cannot allocate device memory - against
als_half_sweep_csr_f32 (the same call with a caller-provided workspace): bit-for-bit equal factors and |delta| on two
calls in a row, the second with a smaller input, so that nothing the first call left in the library's built-in scratch
buffer can leak into the second.  Both matrices hold rows longer than chunk_nnz, i.e. the scratch carries chunk
partials, and rows without entries.  Runs last (file name) and also passes alone in a fresh process."""
import ctypes

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


def _matrix(rng, N, M, avg, maxlen):
    lens = np.minimum(rng.zipf(1.5, size=N) + rng.integers(0, 2 * avg, size=N), maxlen).astype(np.int64)
    lens[rng.integers(0, N, size=max(N // 40, 1))] = 0
    lens[N // 2] = maxlen
    ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    return ptr, ind, dat


def _both(ra, rng, N, M, f, chunk_nnz):
    ptr, ind, dat = _matrix(rng, N, M, 30, min(M - 1, 700))
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=chunk_nnz)
    assert csr.n_slots > 0 and csr.n_long > 0
    Y = torch.from_numpy(Y0).cuda()
    gc = ra.gram(Y, 0.01)
    out = []
    for static in (False, True):
        X = torch.from_numpy(X0).cuda()
        d = torch.zeros(1, dtype=torch.float64, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ra.half_sweep(csr, Y, gc, X, 40.0, absdiff=d, info=info, static_scratch=static)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        out.append((X.cpu().numpy(), float(d.item())))
    ref = c_oracle.half_sweep(ptr, ind, dat, Y0, X0, 0.01, 40.0, dtype=np.float64, nthreads=4)
    return out, ref


@pytest.mark.parametrize("f", [64, 128, 20])
def test_static_scratch_equals_workspace_call_twice_in_a_row(ra, f):
    rng = np.random.default_rng(4100 + f)
    # first call: the larger input; second: fewer rows, fewer slots, other contents
    for (N, M, chunk) in ((1500, 2000, 128), (300, 900, 64)):      # 13.4 MB, then 6.6 MB of scratch at f = 128
        (ws_out, st_out), ref = _both(ra, rng, N, M, f, chunk)
        assert np.array_equal(ws_out[0], st_out[0])              # factors: the same bits
        assert ws_out[1] == st_out[1]                            # sum |delta|: the same bits
        err = np.abs(st_out[0] - ref).max() / np.abs(ref).max()
        assert err < 1e-4, err                                   # and the oracle's bar, as for every half-sweep


def test_static_scratch_rejects_what_does_not_fit(ra):
    from recommender_amd._lib import lib
    limit = lib.als_static_scratch_bytes()
    assert limit == 16 << 20
    assert lib.als_static_scratch_init() == 0
    n_rows = 8_000_000                                           # 32 MB of per-row |delta| alone
    assert lib.als_half_sweep_workspace_bytes(n_rows, 0, 64) > limit
    ptr = np.arange(n_rows + 1, dtype=np.int64) % 2              # never read: the size check comes first
    rc = lib.als_half_sweep_csr_static_f32(None, None, None, n_rows, None, None, n_rows, None, 0, 0, 2048, None, 10, 64,
                                           None, 64, ctypes.c_float(1.0), None, 64, None, None, None, None)
    assert rc == 3                                               # ALS_ERR_WORKSPACE
    msg = lib.als_last_error_string().decode()
    assert "scratch" in msg and str(limit) in msg
    del ptr


def test_static_call_allocates_nothing_and_replays_from_a_graph(ra):
    """What the entry point exists for: the call leaves the device's free memory and torch's allocator untouched, and
    it can be recorded into a HIP graph (no allocation, no synchronisation inside) whose replay returns the eager
    call's bits."""
    from recommender_amd._lib import lib
    rng = np.random.default_rng(77)
    N, M, f = 900, 1500, 128
    ptr, ind, dat = _matrix(rng, N, M, 30, 600)
    X0 = torch.from_numpy(rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)).cuda()
    Y = torch.from_numpy(rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)).cuda()
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=128)
    gc = ra.gram(Y, 0.01)
    assert lib.als_static_scratch_init() == 0
    d = torch.zeros(1, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    X = X0.clone()
    ra.half_sweep(csr, Y, gc, X, 40.0, absdiff=d, info=info, static_scratch=True)        # first use on this device done
    torch.cuda.synchronize()
    eager, eager_d = X.clone(), float(d.item())
    X.copy_(X0)
    torch.cuda.synchronize()
    free0, alloc0, reserved0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    ra.half_sweep(csr, Y, gc, X, 40.0, absdiff=d, info=info, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.cuda.memory_allocated() == alloc0 and torch.cuda.memory_reserved() == reserved0
    assert torch.equal(X, eager) and float(d.item()) == eager_d
    # recorded once, replayed twice from the initial factors
    X.copy_(X0)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            ra.half_sweep(csr, Y, gc, X, 40.0, absdiff=d, info=info, static_scratch=True)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        X.copy_(X0); d.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(X, eager) and float(d.item()) == eager_d
    assert int(info.item()) == 0
