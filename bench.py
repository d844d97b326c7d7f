#!/usr/bin/env python3
"""Benchmark of the implicit-ALS hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ALS iteration (Gram(Y) -> X half-sweep -> Gram(X) -> Y half-sweep) over a
synthetic MovieLens-25M-shaped implicit matrix, f=64 (BASELINE.json configs[1]).  With N
GPUs the matrix is N user blocks of 162 541 users each over the same 59 047 items (weak
scaling): rank r sweeps user block r and an nnz-balanced 1/N of the items, factors are
all-gathered over RCCL after each half-sweep.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2_ml25m_syn")
    ap.add_argument("--factors", type=int, default=None)
    ap.add_argument("--solver", default="cholesky", choices=["cholesky", "cg"],
                    help="cholesky: exact solve (f <= 128, the default workload); cg: fixed-step conjugate gradient")
    ap.add_argument("--cg-steps", type=int, default=3)
    ap.add_argument("--cg-form", default="auto", choices=["auto", "matrix", "lock_step"],
                    help="cg: on the formed matrix (one gather pass) or all rows in lock step (cg_steps+1 passes); "
                         "auto = matrix for f <= 128, lock_step above")
    ap.add_argument("--accumulate", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="exact solver: how the per-row normal matrix is accumulated (see include/als_hip.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work per half-sweep sample")
    return ap.parse_args()


def build_shards(wl, world, rank, dev):
    """This rank's shards of the weak-scaled matrix: (user CSR block, item CSR shard, bounds)."""
    from recommender_amd import data
    from recommender_amd.als import balanced_ranges
    nb, M = wl["n_users"], wl["n_items"]
    N = nb * world
    item_counts = torch.zeros(M, dtype=torch.int64, device=dev)
    blocks = []
    for b in range(world):
        keys, r = data.synth_pairs(nb, M, wl["nnz"], seed=1234 + b, device=dev, item_seed=1234)
        u = torch.div(keys, M, rounding_mode="floor"); i = keys - u * M
        item_counts += torch.bincount(i, minlength=M)
        if b == rank:
            U = data.csr_from_sorted_keys(keys, r, nb, M)
        blocks.append((u + b * nb, i, r))
        del keys
    item_ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(item_counts.cpu().numpy(), out=item_ptr[1:])
    ib = balanced_ranges(item_ptr, world)
    i0, i1 = int(ib[rank]), int(ib[rank + 1])
    ks, rs = [], []
    for (u, i, r) in blocks:
        m = (i >= i0) & (i < i1)
        ks.append((i[m] - i0) * N + u[m]); rs.append(r[m])
    del blocks
    tk, order = torch.sort(torch.cat(ks))
    Ut = data.csr_from_sorted_keys(tk, torch.cat(rs)[order], i1 - i0, N)
    ub = [nb * r for r in range(world + 1)]
    return U, Ut, ub, [int(x) for x in ib], N, M, int(item_ptr[-1])


def cpu_baseline(U, Ut, X0, Y0, wl, seconds):
    """The C restatement of the reference's per-row exact solve (oracle/als_ref.c, float64) on
    the host cores, on an evenly strided row sample of the SAME matrix sized to ~`seconds`
    of work per half-sweep.  Returns nnz/s."""
    from oracle import c_oracle
    threads = c_oracle.max_threads()
    f, lamb, alpha = X0.shape[1], wl["lamb"], wl["alpha"]

    def sample(csr, stride):
        ptr, col, val = (t.cpu().numpy() for t in csr)
        rows = np.arange(0, ptr.shape[0] - 1, stride)
        lens = (ptr[rows + 1] - ptr[rows])
        sp = np.zeros(rows.shape[0] + 1, dtype=np.int64); np.cumsum(lens, out=sp[1:])
        idx = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows]) if rows.size else np.zeros(0, np.int64)
        return sp, col[idx], val[idx].astype(np.float64), rows

    def timed(csr, fixed, own, stride):
        sp, c, v, rows = sample(csr, stride)
        G = c_oracle.gram(fixed)
        t = time.perf_counter()
        c_oracle.half_sweep(sp, c, v, fixed, own[rows], lamb, alpha, nthreads=threads, Y2=G)
        return time.perf_counter() - t, int(sp[-1]), rows.shape[0]

    X64, Y64 = X0.astype(np.float64), Y0.astype(np.float64)
    # probe with a sparse sample, then size the real sample for ~`seconds`
    tu, nu, _ = timed(U, Y64, X64, 400)
    ti, ni, _ = timed(Ut, X64, Y64, 400)
    su = max(1, int(400 * tu / seconds)); si = max(1, int(400 * ti / seconds))
    tu, nu, ru = timed(U, Y64, X64, su)
    ti, ni, ri = timed(Ut, X64, Y64, si)
    return {"value": (nu + ni) / (tu + ti), "unit": "nnz/s", "cores": threads, "kind": "port",
            "sample": "oracle/als_ref.c float64: every %dth user row (%d rows, %d nnz, %.1f s) + every %dth item "
                      "row (%d rows, %d nnz, %.1f s) of the same matrix, one half-sweep each, Gram excluded"
                      % (su, ru, nu, tu, si, ri, ni, ti)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # ALS_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo collectives, to walk the multi-rank code
    # path on a one-GPU box (its timings mean nothing)
    rehearsal = bool(os.environ.get("ALS_BENCH_REHEARSAL"))
    if rehearsal:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(60, exit=True)      # a stuck rehearsal names where it is stuck
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1:
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)
        group = torch.distributed.group.WORLD

    from recommender_amd import data, ops
    from recommender_amd._lib import lib
    from recommender_amd.als import ALSEngine, init_factors
    import ctypes

    ops.set_accumulate_mode(args.accumulate)
    wl = dict(data.WORKLOADS[args.workload])
    if args.factors:
        wl["f"] = args.factors
    f = wl["f"]
    U, Ut, ub, ib, N, M, nnz = build_shards(wl, world, rank, dev)
    if wl.get("init") == "device":      # large shards: same distribution, drawn on the device
        g = torch.Generator(device=dev); g.manual_seed(42)
        X = torch.randn((N, f), generator=g, device=dev) / np.sqrt(f)
        Y = torch.randn((M, f), generator=g, device=dev) / np.sqrt(f)
        Xh = Yh = None
    else:
        Xh, Yh = init_factors(N, M, f, "tf")
        X = torch.from_numpy(Xh.astype(np.float32)).to(dev)
        Y = torch.from_numpy(Yh.astype(np.float32)).to(dev)
    if wl.get("solver") and args.solver == "cholesky" and f > 128:
        args.solver = wl["solver"]
    csr_u = ops.DeviceCSR(U[0], U[1], U[2], (ub[rank + 1] - ub[rank], M), dev)
    # with N user blocks an item row is N times longer: longer chunks keep the partial-sum traffic of the
    # split rows (11 KB per chunk at f=64) small against the gather
    csr_i = ops.DeviceCSR(Ut[0], Ut[1], Ut[2], (ib[rank + 1] - ib[rank], N), dev,
                          chunk_nnz=2048 * (1 if world == 1 else 4))
    eng = ALSEngine(csr_u, csr_i, X, Y, wl["lamb"], wl["alpha"], ub, ib, group, solver=args.solver,
                    cg_steps=args.cg_steps)
    cg = args.solver == "cg"
    lockstep = cg and (args.cg_form == "lock_step" or (args.cg_form == "auto" and f > 128))
    if cg and args.cg_form != "auto":
        import functools
        ops.half_sweep_cg = functools.partial(ops.half_sweep_cg, form=args.cg_form)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step()
    # taps: events around every launch of the gather/accumulate kernel, on the launch stream
    # (the user sweep is several launches when its all-gather is overlapped chunk by chunk)
    launches_per_step = (len(eng._parts()) if world > 1 else 1) + 1
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * launches_per_step * args.steps)]
    for e in pool:
        e.record()          # materialise the handles
    sync()
    taps = {"users": [], "items": []}

    def tap(which):
        e0, e1 = pool.pop(), pool.pop()
        if lockstep:  # several kernels per half-sweep: events around the whole call, on the launch stream
            e0.record()
        else:       # exact path: the library records the events around the one gather/accumulate/solve kernel
            lib.als_profile_tap_next_sweep(ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))
        taps[which].append((e0, e1))

    def tap_end(which):
        if lockstep:
            taps[which][-1][1].record()

    eng.post_sweep_hook = tap_end
    eng.pre_sweep_hook = tap
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    sync()
    dt = time.perf_counter() - t0
    eng.pre_sweep_hook = eng.post_sweep_hook = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if int(eng.info.item()) != 0 and not os.environ.get("ALS_BENCH_IGNORE_INFO"):   # (timing-only ablation builds set it)
        raise SystemExit("non positive definite row systems: %d" % int(eng.info.item()))

    # dominant kernel: the per-row gather/accumulate(+solve) kernel; algorithmic bytes per launch =
    # nnz*(4 col + 4 val + 4f gathered row) + n_rows*(8 rowptr + 4f written row)
    ku = sum(e0.elapsed_time(e1) for e0, e1 in taps["users"]) / args.steps
    ki = sum(e0.elapsed_time(e1) for e0, e1 in taps["items"]) / args.steps
    if lockstep:
        # per half-sweep: cg_steps+1 gather passes over the entries, plus the per-row vectors each pass
        # reads and writes (x, r, p, z and the GEMM's operand and result): see DESIGN.md
        sp = args.cg_steps
        row_vecs = 4 + 7 * (sp - 1) + 6 + 2 * (sp + 1)
        bytes_u = (sp + 1) * csr_u.nnz * (8 + 4 * f) + csr_u.shape[0] * (8 + 4 * f * row_vecs)
        bytes_i = (sp + 1) * csr_i.nnz * (8 + 4 * f) + csr_i.shape[0] * (8 + 4 * f * row_vecs)
    else:
        bytes_u = csr_u.nnz * (8 + 4 * f) + csr_u.shape[0] * (8 + 4 * f)
        bytes_i = csr_i.nnz * (8 + 4 * f) + csr_i.shape[0] * (8 + 4 * f)
    achieved = (bytes_u + bytes_i) / ((ku + ki) * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload + "_f%d" % f + ("_cg" if cg else ""), {}).get("hbm_bytes_per_launch")
            if args.accumulate != "bf16x3":
                traffic = None
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "ALS nnz/sec (2*nnz per iteration; iterations/sec in iters_per_s)",
            "value": 2.0 * nnz * args.steps / dt, "unit": "nnz/s", "iters_per_s": args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d users x %d items, %d nnz, f=%d, lamb=%g, alpha=%g, %s"
                                   % (args.workload, N, M, nnz, f, wl["lamb"], wl["alpha"],
                                      "%d conjugate-gradient steps per row" % args.cg_steps if cg
                                      else "exact Cholesky solve, %s accumulate" % args.accumulate),
                       "parallelism": "dp%d (user/item row shards, all-gather of factors)" % world},
            "roofline": {"bound": "hbm", "kernel": "cg half-sweep (cg_items_kernel passes + Gram GEMM)" if lockstep
                         else "%s (per-row gather/accumulate + %s)" % ("wide_row_items_kernel" if f > 128 else "sweep_items_kernel",
                                                                       "CG on the formed matrix" if cg else "solve"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_ms": (ku + ki) / 2,
                         "launch_ms": {"user_sweep": ku, "item_sweep": ki},
                         "algorithmic_bytes_per_launch": (bytes_u + bytes_i) / 2},
        }
        if world == 1 and not args.no_cpu_baseline and Xh is not None:
            out["cpu_baseline"] = cpu_baseline(U, Ut, Xh, Yh, wl, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
