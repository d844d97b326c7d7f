"""Two ranks on ONE GPU (gloo for the collectives, the HIP kernels for the compute): the sharded
engine must give the single-process factors bit for bit.  This rehearses the multi-GPU code path
(row shards, in-place all-gather, sharded Gram partials summed in row order) with the real
operators on a one-GPU box; RCCL itself needs one device per rank and is exercised by bench.py."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, N, M, f, balance, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)       # a stuck rank names the call it is stuck in
    sys.path.insert(0, ROOT)
    import recommender_amd as ra
    from recommender_amd import data
    from recommender_amd.als import ALSEngine, balanced_ranges, init_factors
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        group = torch.distributed.group.WORLD
    U, Ut = data.synth_matrix(N, M, 40 * N, seed=3, device="cpu")
    if balance == "nnz":
        ub = balanced_ranges(U[0].numpy(), world); ib = balanced_ranges(Ut[0].numpy(), world)
    else:
        ub = np.linspace(0, N, world + 1).astype(np.int64); ib = np.linspace(0, M, world + 1).astype(np.int64)

    def shard(csr, b0, b1, ncols):
        p, c, v = csr
        return ra.DeviceCSR(p[b0:b1 + 1] - p[b0], c[p[b0]:p[b1]], v[p[b0]:p[b1]], (b1 - b0, ncols), dev, chunk_nnz=256)

    cu = shard(U, int(ub[rank]), int(ub[rank + 1]), M)
    ci = shard(Ut, int(ib[rank]), int(ib[rank + 1]), N)
    X0, Y0 = init_factors(N, M, f, "tf")
    X = torch.from_numpy(X0.astype(np.float32)).to(dev); Y = torch.from_numpy(Y0.astype(np.float32)).to(dev)
    # (item sweep in overlapped row chunks as well: what the engine does by itself for large item factors)
    eng = ALSEngine(cu, ci, X, Y, 0.01, 40.0, [int(b) for b in ub], [int(b) for b in ib], group,
                    item_overlap_chunks=2 if world > 1 else None)
    trace = eng.fit(2)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "w%d_r%d_%s.npz" % (world, rank, balance)), X=X.cpu().numpy(), Y=Y.cpu().numpy(),
             trace=np.array(trace))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("balance", ["rows", "nnz"])
def test_two_ranks_one_gpu_equal_single_process(tmp_path, balance):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    # 140 000 users -> 3 Gram partials of 65 536 rows: with world = 1 the Gram is one call, with
    # world = 2 ... still below 2*world partials, so also use 280 000 to take the sharded branch
    for N in (140000, 280000):
        M, f = 3000, 32
        out = str(tmp_path / ("n%d" % N)); os.makedirs(out)
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=_worker, args=(0, 1, 0, N, M, f, balance, out)); p.start(); p.join(300)
        assert p.exitcode == 0
        mp.spawn(_worker, args=(2, _free_port(), N, M, f, balance, out), nprocs=2, join=True)
        ref = np.load(os.path.join(out, "w1_r0_%s.npz" % balance))
        assert np.all(np.isfinite(ref["X"])) and np.all(np.isfinite(ref["Y"]))
        for r in range(2):
            got = np.load(os.path.join(out, "w2_r%d_%s.npz" % (r, balance)))
            assert np.array_equal(got["X"], ref["X"]) and np.array_equal(got["Y"], ref["Y"])
            np.testing.assert_allclose(got["trace"], ref["trace"], rtol=1e-6)


def _rccl_single_rank(rank, port):
    """RCCL itself, with the one rank a one-GPU box allows: the collective calls of the engine (in-place
    all-gather, list outputs that alias the input, a side stream, all_reduce, barrier) are accepted and
    leave the data intact."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = torch.distributed.group.WORLD
    from recommender_amd import als
    T = torch.arange(40, dtype=torch.float32, device=dev).reshape(10, 4).clone(); ref = T.clone()
    als._all_gather_row_blocks(T, [0], [10], 0, g)
    als._all_gather_row_blocks(T, [3], [4], 0, g)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        als._all_gather_row_blocks(T, [2], [5], 0, g)
        ev = side.record_event()
    torch.cuda.current_stream().wait_event(ev)
    # both forms for shards of different sizes, as far as one rank can walk them: the list form (receives land in the
    # tensor's own views) and the padded form (kept buffer + copy-out), the latter twice so that the buffer is reused
    als._all_gather_row_blocks(T, [1], [6], 0, g, uneven="list")
    als._all_gather_padded(T, [2], [5], 0, g)
    als._all_gather_padded(T, [0], [3], 0, g)
    assert len(als._PADDED_RECV) == 1 and next(iter(als._PADDED_RECV.values()))[1].shape[0] == 5
    with pytest.raises(ValueError):
        als._all_gather_row_blocks(T, [0], [10], 0, g, uneven="ring")
    d = torch.ones(2, dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(d, group=g)
    torch.distributed.barrier()
    torch.cuda.synchronize()
    assert torch.equal(T, ref) and float(d.sum()) == 2.0
    torch.distributed.destroy_process_group()


def test_rccl_single_rank_collectives():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    mp.spawn(_rccl_single_rank, args=(_free_port(),), nprocs=1, join=True)
