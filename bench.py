#!/usr/bin/env python3
"""Benchmark of the implicit-ALS hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ALS iteration (Gram(Y) -> X half-sweep -> Gram(X) -> Y half-sweep) over a
synthetic MovieLens-25M-shaped implicit matrix, f=64 (the benchmark's one-GPU configuration).  With N
GPUs the matrix is N user blocks of 162 541 users each over the same 59 047 items (weak
scaling): rank r sweeps user block r and an nnz-balanced 1/N of the items, factors are
all-gathered over RCCL after each half-sweep.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

Besides the headline fields that line carries an `also` list: the same measurement (fewer steps)
at f=128 on the same matrix, exact and conjugate-gradient, and on `c3_shard_syn` - per GPU one
eighth of the 10 M x 1 M, 1 B-entry, f=128 matrix (the benchmark's eight-GPU configuration); with N GPUs the
matrix is N such blocks with user shards cut by stored entries, so N=8 is that configuration.
`--no-also` leaves them out; `--workload/--factors/--solver` move the headline itself.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3      # dense float32 matrix-core peak (256 CUs x 256 flops per cycle x 2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 matrix-core peak
SIMDS, CLOCK_HZ = 1024, 2.4e9     # 256 CUs x 4 SIMDs, each with one matrix pipe; peak engine clock


def algorithmic_flops(nnz, rows, f):
    """float32-equivalent flops of one exact half-sweep: per stored entry the rank-1 update of the symmetric f x f
    normal matrix (f (f+1) / 2 multiply-adds) and of b (f), per row the Cholesky factorisation (f^3 / 3 flops) and the
    two triangular solves (2 f^2)."""
    return nnz * (f * (f + 1) + 2 * f) + rows * (f ** 3 / 3.0 + 2.0 * f * f)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2_ml25m_syn")
    ap.add_argument("--factors", type=int, default=None)
    ap.add_argument("--solver", default="cholesky", choices=["cholesky", "cg"],
                    help="cholesky: exact solve (f <= 128, the default workload); cg: fixed-step conjugate gradient")
    ap.add_argument("--cg-steps", type=int, default=None,
                    help="conjugate-gradient steps per row (default: max(10, ceil(f / 6)), the count at which training "
                         "metrics equal the exact solve's to 3 decimals; 3 is quoted beside it)")
    ap.add_argument("--cg-form", default="auto", choices=["auto", "matrix", "lock_step"],
                    help="cg: on the formed matrix (one gather pass) or all rows in lock step (cg_steps+1 passes); "
                         "auto = matrix for f <= 128, lock_step above")
    ap.add_argument("--accumulate", default="default", choices=["default", "fp16x2", "bf16x3", "fp32"],
                    help="exact solver: how the per-row normal matrix is accumulated (see include/als_hip.h; default = the "
                         "library's choice: fp16x2 where the fixed factor has at least 16 rows per factor, else bf16x3)")
    ap.add_argument("--row-kernel", default="auto", choices=["auto", "wave", "stream"],
                    help="64 < f <= 128: one wavefront per row, or one resident wavefront per SIMD streaming its rows")
    ap.add_argument("--short-rows", default="auto", choices=["auto", "on", "off"],
                    help="exact solver, 32 < f <= 128: rows with fewer stored entries than factors solved in their "
                         "entry-space form (auto: where it pays, see recommender_amd/als.py::short_rows_pay)")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="multi-GPU, 64 < f <= 128: compute units the persistent row kernel leaves free for the RCCL "
                         "exchange that runs beside it (default: the engine's choice, 0 = none; see --gap-ticks)")
    ap.add_argument("--gap-ticks", type=int, default=None,
                    help="multi-GPU: length of the spin kernel behind the event that holds the next row chunk's sweep back until "
                         "the exchange of the previous one has been released (default: the engine's 20000; 0 = event only; "
                         "-1 = no gap at all)")
    ap.add_argument("--overlap-chunks", type=int, default=None,
                    help="row chunks of the user sweep whose exchange runs beside the next chunk's sweep (default: 4 with "
                         "several GPUs, 1 on one; on one GPU a value > 1 measures what the chunking itself costs)")
    ap.add_argument("--tap-every", type=int, default=1,
                    help="the per-row kernels of every n-th timed step are bracketed by events (launch durations of the roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="only the headline measurement")
    ap.add_argument("--also-steps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work per half-sweep sample")
    return ap.parse_args()


def build_shards(wl, world, rank, dev):
    """This rank's shards of the matrix: (user CSR shard, item CSR shard, user bounds, item bounds, N, M, nnz).
    The matrix is `blocks` user blocks of wl["n_users"] users over the same items: blocks = wl["blocks"] if the
    workload fixes it (the same matrix for any number of GPUs: strong scaling), else `world` (one block per GPU: weak
    scaling).  Block b holds wl["nnz"] entries, or - with wl["block_jitter"] and several blocks - that many times
    1 + jitter * (a fixed pattern with zero mean), so that the blocks differ as real user populations do.  User
    shards are contiguous ranges with about equal entries (wl["balance_users"], always when blocks != world), or the
    blocks themselves."""
    from recommender_amd import data
    from recommender_amd.als import balanced_ranges
    nb, M = wl["n_users"], wl["n_items"]
    n_blocks = int(wl.get("blocks", world))
    N = nb * n_blocks
    jit = float(wl.get("block_jitter", 0.0))
    pattern = [((b * 5) % n_blocks) / max(n_blocks - 1, 1) * 2.0 - 1.0 for b in range(n_blocks)]      # a permutation of [-1, 1]
    mean = sum(pattern) / n_blocks
    item_counts = torch.zeros(M, dtype=torch.int64, device=dev)
    user_counts = []
    blocks = []
    for b in range(n_blocks):
        nnz_b = int(round(wl["nnz"] * (1.0 + jit * (pattern[b] - mean)))) if n_blocks > 1 else wl["nnz"]
        keys, r = data.synth_pairs(nb, M, nnz_b, seed=1234 + b, device=dev, item_seed=1234)
        u = torch.div(keys, M, rounding_mode="floor"); i = keys - u * M
        item_counts += torch.bincount(i, minlength=M)
        user_counts.append(torch.bincount(u, minlength=nb))
        blocks.append((u + b * nb, i, r))
        del keys
    item_ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(item_counts.cpu().numpy(), out=item_ptr[1:])
    ib = balanced_ranges(item_ptr, world)
    if (wl.get("balance_users") and world > 1) or n_blocks != world:
        user_ptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(torch.cat(user_counts).cpu().numpy(), out=user_ptr[1:])
        ub = [int(x) for x in balanced_ranges(user_ptr, world)]
    else:
        ub = [nb * r for r in range(world + 1)]
    u0, u1 = ub[rank], ub[rank + 1]
    ks, rs = [], []
    for (u, i, r) in blocks:                       # blocks are in user order, each sorted by (user, item)
        m = (u >= u0) & (u < u1)
        if bool(m.any()):
            ks.append((u[m] - u0) * M + i[m]); rs.append(r[m])
    U = data.csr_from_sorted_keys(torch.cat(ks), torch.cat(rs), u1 - u0, M)
    i0, i1 = int(ib[rank]), int(ib[rank + 1])
    ks, rs = [], []
    for (u, i, r) in blocks:
        m = (i >= i0) & (i < i1)
        ks.append((i[m] - i0) * N + u[m]); rs.append(r[m])
    del blocks
    tk, order = torch.sort(torch.cat(ks))
    Ut = data.csr_from_sorted_keys(tk, torch.cat(rs)[order], i1 - i0, N)
    return U, Ut, ub, [int(x) for x in ib], N, M, int(item_ptr[-1])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(U, Ut, X0, Y0, wl, seconds):
    """The C restatement of the reference's per-row exact solve (oracle/als_ref.c, float64) on
    the host cores, on an evenly strided row sample of the SAME matrix sized to ~`seconds`
    of work per half-sweep.  Returns nnz/s.  Beside it, once, the float32 restatement of the
    reference's TensorFlow path on padded batches of 100 rows (oracle/als_ref.py::half_sweep_batched_f32,
    reclibwh/core/ALS.py:160-233) on the C1-shaped matrix, timed on numpy's threads."""
    from oracle import als_ref, c_oracle
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
    out = {"value": (nu + ni) / (tu + ti), "unit": "nnz/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "host_cpus": os.cpu_count(),
           "sample": "oracle/als_ref.c float64: every %dth user row (%d rows, %d nnz, %.1f s) + every %dth item "
                     "row (%d rows, %d nnz, %.1f s) of the same matrix, one half-sweep each, Gram excluded"
                     % (su, ru, nu, tu, si, ri, ni, ti)}
    # the TF path's batched float32 form on the C1-shaped matrix (610 x 9724, f=20): one full iteration
    try:
        from recommender_amd import data
        from recommender_amd.als import init_factors
        c1 = data.WORKLOADS["c1_ml_small_syn"]
        U1, Ut1 = data.synth_matrix(c1["n_users"], c1["n_items"], c1["nnz"], seed=1234, device="cpu")
        Xi, Yi = init_factors(c1["n_users"] + 1, c1["n_items"] + 1, c1["f"], "tf")
        p, c, v = (t.numpy() for t in U1)
        pt, ct, vt = (t.numpy() for t in Ut1)
        t = time.perf_counter()
        X1, _ = als_ref.half_sweep_batched_f32(p, c, v, c1["n_items"], Yi, Xi, c1["lamb"], c1["alpha"], batchsize=100)
        als_ref.half_sweep_batched_f32(pt, ct, vt, c1["n_users"], X1, Yi, c1["lamb"], c1["alpha"], batchsize=100)
        dt = time.perf_counter() - t
        out["batched_f32"] = {"value": 2.0 * c1["nnz"] / dt, "unit": "nnz/s", "kind": "port",
                              "threads": "numpy/BLAS default (%d host cpus visible)" % os.cpu_count(),
                              "sample": "oracle half_sweep_batched_f32 (padded batches of 100 rows, float32, np.linalg.inv): one "
                                        "iteration of c1_ml_small_syn 610 x 9724, %d nnz, f=20, %.1f s" % (c1["nnz"], dt)}
    except Exception as exc:        # the second baseline must never cost the headline line
        out["batched_f32"] = {"error": repr(exc)}
    return out


class Ctx:
    pass


def measure(ctx, workload, f, solver, cg_steps, cg_form, accumulate, row_kernel, steps, warmup, shards=None,
            short_rows="auto", reserve_cus=None, gap_ticks=None, tap_every=1, overlap_chunks=None):
    """Builds (or reuses) the shards of `workload`, runs warmup + `steps` timed iterations and returns
    (result dict, shards, host init or None).  Every per-row kernel of every half-sweep is bracketed by
    events the library records on the launch stream (als_sweep_opts_t.tap_begin/tap_end)."""
    from recommender_amd import data, ops
    from recommender_amd.als import ALSEngine, init_factors
    world, rank, dev, group = ctx.world, ctx.rank, ctx.dev, ctx.group
    wl = dict(data.WORKLOADS[workload])
    if f:
        wl["f"] = f
    f = wl["f"]
    if shards is None:
        shards = build_shards(wl, world, rank, dev)
    U, Ut, ub, ib, N, M, nnz = shards
    if wl.get("init") == "device":      # large shards: same distribution, drawn on the device
        g = torch.Generator(device=dev); g.manual_seed(42)
        X = torch.randn((N, f), generator=g, device=dev) / np.sqrt(f)
        Y = torch.randn((M, f), generator=g, device=dev) / np.sqrt(f)
        Xh = Yh = None
    else:
        Xh, Yh = init_factors(N, M, f, "tf")
        X = torch.from_numpy(Xh.astype(np.float32)).to(dev)
        Y = torch.from_numpy(Yh.astype(np.float32)).to(dev)
    if f > 128:
        solver = "cg"
    if cg_steps is None:
        from recommender_amd.als import default_cg_steps
        cg_steps = default_cg_steps(f)
    # entries per chunk item of a long row: the library's choice for this f (ALS_BENCH_CHUNK_NNZ: experiments)
    base_chunk = int(os.environ.get("ALS_BENCH_CHUNK_NNZ", "0")) or ops.default_chunk_nnz(f)
    csr_u = ops.DeviceCSR(U[0], U[1], U[2], (ub[rank + 1] - ub[rank], M), dev, chunk_nnz=base_chunk)
    csr_i = ops.DeviceCSR(Ut[0], Ut[1], Ut[2], (ib[rank + 1] - ib[rank], N), dev, chunk_nnz=base_chunk)
    eng = ALSEngine(csr_u, csr_i, X, Y, wl["lamb"], wl["alpha"], ub, ib, group, solver=solver,
                    cg_steps=cg_steps, cg_form=cg_form, short_rows={"auto": "auto", "on": True, "off": False}[short_rows]
                    if solver != "cg" else False, stream_reserve_cus=reserve_cus, overlap_chunks=overlap_chunks,
                    **({} if gap_ticks is None else {"exchange_gap_ticks": gap_ticks}))
    short = dict(eng.short)
    cg = solver == "cg"
    lockstep = cg and (cg_form == "lock_step" or (cg_form == "auto" and f > 128))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    base_opts = dict(accumulate=accumulate, row_kernel=row_kernel)
    eng.sweep_opts = lambda which: ops.sweep_opts(**base_opts)
    for _ in range(warmup):
        eng.step()
    launches_per_step = eng.launches_per_step()
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * launches_per_step * steps)]
    for e in pool:
        e.record()          # materialise the handles
    sync()
    taps = {"users": [], "items": []}

    # Event records cost the stream a moment each (four per step; measured: 0.4 % of the 2 ms step of the 25 M-entry
    # matrix); --tap-every n taps only every n-th step of the timed region
    state = {"step": 0}
    tapped_steps = [s for s in range(steps) if s % tap_every == 0]

    def tapped(which):
        if state["step"] % tap_every:
            return ops.sweep_opts(**base_opts)
        e0, e1 = pool.pop(), pool.pop()
        taps[which].append((e0, e1))
        return ops.sweep_opts(tap=(e0, e1), **base_opts)

    eng.sweep_opts = tapped
    if world > 1:
        eng.exchange_taps = {"x_wait": [], "y_gather": []}
    sync()
    t0 = time.perf_counter()
    for s_i in range(steps):
        state["step"] = s_i
        eng.step()
    sync()
    dt = time.perf_counter() - t0
    eng.sweep_opts = None
    exposed = None
    if world > 1:
        # what the main stream spends waiting for exchanges per step: the part of the user exchange still in flight
        # when the user sweep is queued, and the item exchange (nothing overlaps it); max over ranks
        mine = [sum(a.elapsed_time(b) for a, b in eng.exchange_taps[k]) / steps for k in ("x_wait", "y_gather")]
        t = torch.tensor(mine, dtype=torch.float64, device="cpu" if ctx.rehearsal else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        exposed = {"user_exchange_wait_ms": float(t[0]), "item_exchange_ms": float(t[1]),
                   "exposed_exchange_ms": float(t[0] + t[1]), "note": "max over ranks, per iteration"}
        eng.exchange_taps = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if ctx.rehearsal else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if int(eng.info.item()) != 0 and not os.environ.get("ALS_BENCH_IGNORE_INFO"):   # (timing-only ablation builds set it)
        raise SystemExit("non positive definite row systems: %d" % int(eng.info.item()))

    # dominant kernels: the per-row kernels of a half-sweep (gather/accumulate + solve, the long rows' combine
    # and solve); algorithmic bytes per half-sweep = nnz*(4 col + 4 val + 4f gathered row) + n_rows*(8 rowptr
    # + 4f written row)
    ku = sum(e0.elapsed_time(e1) for e0, e1 in taps["users"]) / len(tapped_steps)
    ki = sum(e0.elapsed_time(e1) for e0, e1 in taps["items"]) / len(tapped_steps)
    if lockstep:
        # per half-sweep: cg_steps+1 gather passes over the entries, plus the per-row vectors each pass
        # reads and writes (x, r, p, z and the GEMM's operand and result)
        sp = cg_steps
        row_vecs = 4 + 7 * (sp - 1) + 6 + 2 * (sp + 1)
        bytes_u = (sp + 1) * csr_u.nnz * (8 + 4 * f) + csr_u.shape[0] * (8 + 4 * f * row_vecs)
        bytes_i = (sp + 1) * csr_i.nnz * (8 + 4 * f) + csr_i.shape[0] * (8 + 4 * f * row_vecs)
    else:
        bytes_u = csr_u.nnz * (8 + 4 * f) + csr_u.shape[0] * (8 + 4 * f)
        bytes_i = csr_i.nnz * (8 + 4 * f) + csr_i.shape[0] * (8 + 4 * f)
    achieved = (bytes_u + bytes_i) / ((ku + ki) * 1e-3) / 1e9
    res = {
        "workload": "%s: %d users x %d items, %d nnz, f=%d, lamb=%g, alpha=%g, %s"
                    % (workload, N, M, nnz, f, wl["lamb"], wl["alpha"],
                       "%d conjugate-gradient steps per row (%s)" % (cg_steps, "lock step" if lockstep else "formed matrix")
                       if cg else "exact Cholesky solve, %s accumulate" % ("fp16x2" if accumulate == "default" else accumulate)),
        "nnz": nnz, "f": f, "solver": solver, "cg_steps": cg_steps if cg else None, "steps": steps, "seconds": dt,
        "value": 2.0 * nnz * steps / dt, "ms_per_step": dt / steps * 1e3,
        "kernel": ("cg half-sweep (cg_items_kernel passes + Gram GEMM)" if lockstep else
                   "per-row kernels of the half-sweep (%s: gather/accumulate + %s; the long rows' combine and solve "
                   "included)" % ("wide_row_items_kernel" if f > 128 else
                                  "sweep_stream_kernel for whole rows (row_kernel=%s), sweep_items_kernel for long rows' chunks"
                                  % row_kernel if f > 64 and row_kernel != "wave" and accumulate != "fp32" else
                                  "sweep_items_kernel, row_kernel=%s" % row_kernel,
                                  "CG on the formed matrix" if cg else "solve")),
        "achieved": achieved, "launch_ms": {"user_sweep": ku, "item_sweep": ki}, "tapped_steps": len(tapped_steps),
        "bytes_per_launch": (bytes_u + bytes_i) / 2,
        "user_rows": csr_u.shape[0], "item_rows": csr_i.shape[0],
        "short_rows": short_rows_note(short, csr_u, csr_i, f),
    }
    if exposed is not None:
        exposed["stream_reserve_cus"] = eng.stream_reserve_cus
        exposed["exchange_gap_ticks"] = eng.exchange_gap_ticks
        res["exchange"] = exposed
    if "blocks" in wl:
        res["scaling"] = strong_scaling(workload, f, solver, world, dt / steps * 1e3)
    if not cg:
        fl = algorithmic_flops(csr_u.nnz, csr_u.shape[0], f) + algorithmic_flops(csr_i.nnz, csr_i.shape[0], f)
        res["mfma"] = mfma_view(fl / 2, (bytes_u + bytes_i) / 2, (ku + ki) / 2)
        res["mfma"]["matrix_pipe_busy"] = {"user_sweep": pipe_busy(matrix_pipe_cycles(csr_u, f, short["users"], accumulate), ku),
                                           "item_sweep": pipe_busy(matrix_pipe_cycles(csr_i, f, short["items"], accumulate), ki)}
    del eng, csr_u, csr_i, X, Y
    return res, shards, (Xh, Yh, wl)


def short_rows_note(short, csr_u, csr_i, f):
    """Which half-sweeps solved their short rows in the entry-space form, and how many rows that was."""
    from recommender_amd import ops
    lim = ops.short_rows_max_entries(f)
    nb = lim // 16
    out = {"max_entries": lim}
    for which, csr in (("users", csr_u), ("items", csr_i)):
        n = int(csr.short_bounds[-1] - csr.short_bounds[nb]) if lim else 0
        out[which] = {"entry_space_form": bool(short.get(which)), "rows_within_limit": n, "rows": csr.shape[0]}
    return out


def tile_products(n):
    """16 x 16 tile products of a right-looking blocked Cholesky on n block rows: panel solves + trailing updates."""
    return n * (n - 1) // 2 + (n - 1) * n * (n + 1) // 6


def matrix_pipe_cycles(csr, f, short, accumulate="default"):
    """Matrix-pipe cycles the exact half-sweep over `csr` issues (counted from the row lengths, not measured):
    a bf16 MFMA (16x16x32 or 16x16x16) holds a SIMD's pipe 16 cycles, a float32 16x16x4 one 32.
    f x f form: per 32-entry block T tiles x 3 (fp16 x 2) or 6 (bf16 x 3) piece products; per row the blocked Cholesky's tile products
    (f > 64: 6 bf16 MFMAs each, else 4 float32 ones).  Entry-space form (rows of at most 16 j entries, j block rows):
    T(j) tiles x 6 products x f/32 slices, then the j-block Cholesky.  f <= 32 (float32 accumulate) is not counted."""
    if f <= 32 or accumulate == "fp32":
        return None
    # piece products per tile and 32 entries: three (fp16 x 2: the default where the fixed factor has at least 16 rows
    # per factor) or six (bf16 x 3)
    pp = 3 if accumulate == "fp16x2" or (accumulate == "default" and csr.shape[1] >= 16 * f and csr.max_rating > 0) else 6
    vw = 4 if f <= 64 else 8
    T = vw * (vw + 1) // 2
    lens = (csr.rowptr[1:] - csr.rowptr[:-1])
    lim = 0
    cyc = 0.0
    if short:
        from recommender_amd import ops
        lim = ops.short_rows_max_entries(f)
        ks = 2 if f <= 64 else 4
        b = csr.short_bounds
        for j in range(1, lim // 16 + 1):
            rows = int((b[j - 1] if j > 1 else b[-1]) - b[j])
            tj = j * (j + 1) // 2
            solve = tile_products(j) * (6 * 16 if j >= 8 else 4 * 32)
            cyc += rows * (tj * pp * ks * 16 + solve)
    long_rows = lens > lim
    blocks = float(torch.div(lens[long_rows] + 31, 32, rounding_mode="floor").sum().item())
    n_long = float(long_rows.sum().item())
    cyc += blocks * T * pp * 16 + n_long * tile_products(vw) * (6 * 16 if vw >= 8 else 4 * 32)
    return cyc


def pipe_busy(cycles_per_launch, launch_ms):
    """Fraction of the chip's matrix-pipe time the launch's MFMAs occupy at the peak clock."""
    if cycles_per_launch is None:
        return None
    return cycles_per_launch / (SIMDS * CLOCK_HZ * launch_ms * 1e-3)


def measure_gram(dev):
    """MFMA utilisation of the Gram (als_gram_f32) on resident factor tables: float32-equivalent flops of the
    symmetric tiles against the float32 matrix roof (the figure of earlier rounds), and the matrix-pipe busy
    fraction of the bf16 x 3 form it runs in for 32 < f <= 128 (T tiles x 6 products x 16 cycles per 32 rows)."""
    from recommender_amd import ops
    out = []
    for n, f in [(59047, 64), (162541, 64), (59047, 128), (162541, 128), (1000000, 128)]:
        Y = torch.randn(n, f, device=dev) * 0.1
        gc = ops.gram(Y, 0.01)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ops.gram(Y, 0.01, out=gc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        vw = 4 if f <= 64 else 8
        T = vw * (vw + 1) // 2
        flops = n * T * 512.0
        out.append({"rows": n, "f": f, "ms": ms, "mfma_util_f32_equivalent": flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                    "matrix_pipe_busy": pipe_busy(n / 32.0 * T * 6 * 16, ms), "read_GBs": n * f * 4 / ms / 1e6})
        del Y, gc
    return out


def measure_topk(dev, users=65536, items=1000000, f=128, k=100):
    """The serving configuration: fused top-k of `users` users against `items` items in blocks of 4096 users, all
    resident.  Two forms: against the item table prepared once per model (als_items_prepare_f32: two fp16 pieces per
    value, three piece products: floor = 6 users items f flops at the dense fp16 peak) - what RecommenderALS.recommend
    runs - and against the float32 factor (six bf16 piece products per block: floor = 12 users items f flops)."""
    from recommender_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((users, f), generator=g, device=dev) / f ** 0.5
    Y = torch.randn((items, f), generator=g, device=dev) / f ** 0.5
    u = torch.arange(users, device=dev)

    def timed(fn, reps=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    ops.prepare_items(Y)
    prep_ms, P = timed(lambda: ops.prepare_items(Y), 3)
    out = {"workload": "c5: fused top-%d of %d users x %d items, f=%d, blocks of 4096 users, no exclusions" % (k, users, items, f)}
    for name, T, products in (("prepared_items", P, 3), ("float32_items", Y, 6)):
        ops.recommend_topk(X, T, u[:4096], k)
        torch.cuda.synchronize()
        ms, _ = timed(lambda: ops.recommend_topk(X, T, u, k))
        floor_ms = 2.0 * products * users * items * f / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3
        out[name] = {"ms": ms, "users_per_s": users / ms * 1e3,
                     "bound": "mfma (%s pipe, %d piece products)" % ("fp16" if products == 3 else "bf16", products),
                     "floor_ms": floor_ms, "frac_of_floor": floor_ms / ms,
                     "fp32_equivalent_TFLOPs": 2.0 * users * items * f / ms / 1e9}
    out["prepared_items"]["prepare_ms_once_per_model"] = prep_ms
    out["ms"] = out["prepared_items"]["ms"]
    out["users_per_s"] = out["prepared_items"]["users_per_s"]
    return out


def strong_scaling(workload, f, solver, world, ms_per_step):
    """A workload of fixed size (wl["blocks"]) is the same matrix for any number of GPUs: efficiency = T_1 / (g T_g).
    On one GPU this run IS T_1; on g > 1 GPUs T_1 comes from the single-GPU measurement kept under profiles/
    (profiles/scaling_t1.json, written from a --gpus 1 run of the same workload), or is unknown."""
    out = {"kind": "strong", "gpus": world, "ms_per_step": ms_per_step}
    if world == 1:
        out.update(T1_ms=ms_per_step, T1_source="this run", efficiency=1.0)
        return out
    key = "%s_f%d_%s" % (workload, f, solver)
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "scaling_t1.json")))[key]
        out.update(T1_ms=rec["ms_per_step"], T1_source="profiles/scaling_t1.json[%s] (%s)" % (key, rec.get("source", "")),
                   efficiency=rec["ms_per_step"] / (world * ms_per_step))
    except Exception:
        out.update(T1_ms=None, T1_source="no single-GPU measurement of %s on record" % key, efficiency=None)
    return out


def mfma_view(flops_per_launch, bytes_per_launch, launch_ms):
    """The same launches against the float32 matrix-core roof: above 157.3 / 8 = 19.7 flops per byte the exact
    half-sweep is bound by arithmetic, not by HBM (f = 128: 38 flops per byte on MovieLens-like rows, 58 on rows of
    about 100 entries, where the f^3 / 3 of the solve weighs as much as the accumulation)."""
    # `frac` is float32-EQUIVALENT flops over the float32 matrix roof: the bf16 x 3 products run on a pipe 16 times
    # faster, so it is not a utilisation - `matrix_pipe_busy` (added by the callers) is: MFMA cycles issued / pipe time
    tf = flops_per_launch / (launch_ms * 1e-3) / 1e12
    return {"bound_by": "mfma" if flops_per_launch / bytes_per_launch > MFMA_F32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else "hbm",
            "flops_per_byte": flops_per_launch / bytes_per_launch,
            "algorithmic_flops_per_launch": flops_per_launch, "achieved_TFLOPs_f32_equivalent": tf,
            "peak_TFLOPs_f32_matrix": MFMA_F32_PEAK_TFLOPS, "frac": tf / MFMA_F32_PEAK_TFLOPS}


def measure_rank_share(ctx, wl_name, solver, cg_steps, accumulate, row_kernel, steps, warmup, state=None,
                       short_rows="auto"):
    """What ONE rank of the `of_world`-GPU run of the full matrix executes per iteration, on one GPU: the user
    half-sweep over its n_users x n_items block, and the item half-sweep over its n_items / of_world items against ALL
    n_users * of_world users - rows of_world times longer than in the one-block matrix, gathered from the whole user
    table (5 GB at f = 128).  The all-gathers between the half-sweeps are not part of it (a one-GPU box).  The Gram of
    the user table is taken over the rank's share of its rows, as in the sharded run."""
    from recommender_amd import data, ops
    dev = ctx.dev
    wl = dict(data.WORKLOADS[wl_name])
    W, nb, M, f = wl["of_world"], wl["n_users"], wl["n_items"], wl["f"]
    N, mi = nb * W, M // W
    if cg_steps is None:
        from recommender_amd.als import default_cg_steps
        cg_steps = default_cg_steps(f)
    if state is None:
        keys, r = data.synth_pairs(nb, M, wl["nnz"], seed=1234, device=dev, item_seed=1234)
        U = data.csr_from_sorted_keys(keys, r, nb, M)
        del keys, r
        # the rank's items over all users: user activity log-normal over the N users, item popularity over its mi items
        keys, r = data.synth_pairs(N, mi, wl["nnz"], seed=4321, device=dev)
        tk, tr = data.transpose_keys(keys, r, N, mi)
        del keys, r
        Ut = data.csr_from_sorted_keys(tk, tr, mi, N)
        del tk, tr
        torch.cuda.empty_cache()
        state = (ops.DeviceCSR(U[0], U[1], U[2], (nb, M), dev, chunk_nnz=ops.default_chunk_nnz(f)),
                 ops.DeviceCSR(Ut[0], Ut[1], Ut[2], (mi, N), dev, chunk_nnz=ops.default_chunk_nnz(f)))
    csr_u, csr_i = state
    g = torch.Generator(device=dev); g.manual_seed(42)
    X = torch.randn((N, f), generator=g, device=dev) / np.sqrt(f)
    Y = torch.randn((M, f), generator=g, device=dev) / np.sqrt(f)
    cache_Y, cache_X = ops.GramCache(f, wl["lamb"], dev), ops.GramCache(f, wl["lamb"], dev)
    ops.gram(X, wl["lamb"], out=cache_X)            # the other ranks' rows of X do not change in this model
    dX = torch.zeros(1, dtype=torch.float64, device=dev); dY = torch.zeros(1, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    cg = solver == "cg"
    base = dict(accumulate=accumulate, row_kernel=row_kernel)
    taps = {"users": [], "items": []}
    # the decision of the W-rank run: totals over the whole matrix (this rank's user block is one of W alike; its
    # item rows are one W-th of the items)
    from recommender_amd.als import short_rows_pay
    lim = ops.short_rows_max_entries(f)
    short = {"users": False, "items": False}
    if not cg and lim and short_rows != "off":
        nsu = int(csr_u.short_bounds[-1] - csr_u.short_bounds[lim // 16]) * W
        nsi = int(csr_i.short_bounds[-1] - csr_i.short_bounds[lim // 16]) * W
        short = {"users": short_rows == "on" or short_rows_pay(nsu, N, M, f),
                 "items": short_rows == "on" or short_rows_pay(nsi, M, N, f)}

    def sweep(csr, opp, cache, own, d, which, timed):
        opts = ops.sweep_opts(**base)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            taps[which].append((e0, e1))
            opts = ops.sweep_opts(tap=(e0, e1), **base)
        if cg:
            ops.half_sweep_cg(csr, opp, cache, own, wl["alpha"], cg_steps, absdiff=d, info=info, opts=opts)
        else:
            ops.half_sweep(csr, opp, cache, own, wl["alpha"], absdiff=d, info=info, opts=opts, short_rows=short[which])

    def step(timed):
        ops.gram(Y, wl["lamb"], out=cache_Y)
        if short["users"]:
            ops.whiten(Y, cache_Y, info=info)
        sweep(csr_u, Y, cache_Y, X[:nb], dX, "users", timed)
        ops.gram_partials(X, 0, nb)                 # this rank's share of the user table's Gram partials
        if short["items"]:
            ops.whiten(X, cache_X, info=info)
        sweep(csr_i, X, cache_X, Y[:mi], dY, "items", timed)

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(info.item()) != 0 and not os.environ.get("ALS_BENCH_IGNORE_INFO"):
        raise SystemExit("non positive definite row systems: %d" % int(info.item()))
    ku = sum(e0.elapsed_time(e1) for e0, e1 in taps["users"]) / steps
    ki = sum(e0.elapsed_time(e1) for e0, e1 in taps["items"]) / steps
    bu = csr_u.nnz * (8 + 4 * f) + csr_u.shape[0] * (8 + 4 * f)
    bi = csr_i.nnz * (8 + 4 * f) + csr_i.shape[0] * (8 + 4 * f)
    res = {"workload": "%s: one rank's share of %d users x %d items on %d GPUs, f=%d: user half-sweep %d x %d (%d nnz), item "
                       "half-sweep %d x %d (%d nnz); %s; exchanges not included"
                       % (wl_name, N, M, W, f, nb, M, csr_u.nnz, mi, N, csr_i.nnz,
                          "%d conjugate-gradient steps per row (formed matrix)" % cg_steps if cg else
                          "exact Cholesky solve, %s accumulate" % ("fp16x2" if accumulate == "default" else accumulate)),
           "f": f, "solver": solver, "cg_steps": cg_steps if cg else None, "steps": steps,
           "nnz_per_s": (csr_u.nnz + csr_i.nnz) * steps / dt, "ms_per_step": dt / steps * 1e3,
           "launch_ms": {"user_sweep": ku, "item_sweep": ki},
           "achieved_GBs": (bu + bi) / ((ku + ki) * 1e-3) / 1e9,
           "frac": (bu + bi) / ((ku + ki) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "frac_per_launch": {"user_sweep": bu / (ku * 1e-3) / 1e9 / HBM_PEAK_GBS, "item_sweep": bi / (ki * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "algorithmic_bytes_per_launch": (bu + bi) / 2,
           "traffic": pmc_traffic("%s_f%d" % (wl_name, f)) if not cg else None,
           "rows_this_rank": {"user": csr_u.shape[0], "item": csr_i.shape[0]},
           "short_rows": short_rows_note(short, csr_u, csr_i, f)}
    if not cg:
        fl = algorithmic_flops(csr_u.nnz, csr_u.shape[0], f) + algorithmic_flops(csr_i.nnz, csr_i.shape[0], f)
        res["mfma"] = mfma_view(fl / 2, (bu + bi) / 2, (ku + ki) / 2)
        res["mfma"]["matrix_pipe_busy"] = {"user_sweep": pipe_busy(matrix_pipe_cycles(csr_u, f, short["users"], accumulate), ku),
                                           "item_sweep": pipe_busy(matrix_pipe_cycles(csr_i, f, short["items"], accumulate), ki)}
    del X, Y
    return res, state


def pmc_traffic(key):
    """HBM bytes per launch from the PMC passes kept under profiles/ (a separate rocprofv3 run of the same
    command: FETCH_SIZE x 2 + WRITE_SIZE); None if that configuration was not profiled."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(tpath)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    ctx = Ctx()
    # ALS_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo collectives, to walk the multi-rank code
    # path on a one-GPU box (its timings mean nothing)
    ctx.rehearsal = bool(os.environ.get("ALS_BENCH_REHEARSAL"))
    if ctx.rehearsal:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("ALS_BENCH_REHEARSAL_TIMEOUT", "240")), exit=True)
    torch.cuda.set_device(local)
    ctx.dev = torch.device("cuda", local)
    ctx.world, ctx.rank, ctx.group = world, rank, None
    backend = None
    if world > 1:
        backend = "gloo" if ctx.rehearsal else "nccl"
        if ctx.rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=ctx.dev)
        ctx.group = torch.distributed.group.WORLD

    from recommender_amd import data as _data
    if "of_world" in _data.WORKLOADS.get(args.workload, {}):       # one rank's share of a sharded run, by itself
        if world != 1:
            raise SystemExit("%s models one rank of a sharded run on one GPU: --gpus 1" % args.workload)
        r, _ = measure_rank_share(ctx, args.workload, args.solver, args.cg_steps, args.accumulate, args.row_kernel,
                                  args.steps, args.warmup, short_rows=args.short_rows)
        print(json.dumps({"metric": "ALS nnz/sec (2*nnz per iteration; iterations/sec in iters_per_s)",
                          "value": r["nnz_per_s"], "unit": "nnz/s", "iters_per_s": 1e3 / r["ms_per_step"], "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": r["workload"]},
                          "roofline": {"bound": "hbm", "achieved": r["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": r["frac"], "frac_per_launch": r["frac_per_launch"], "traffic": None,
                                       "launch_ms": r["launch_ms"],
                                       "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                                       "mfma_view": r.get("mfma"), "short_rows": r["short_rows"]}}), flush=True)
        return
    head, shards, (Xh, Yh, wl) = measure(ctx, args.workload, args.factors, args.solver, args.cg_steps, args.cg_form,
                                         args.accumulate, args.row_kernel, args.steps, args.warmup,
                                         short_rows=args.short_rows, reserve_cus=args.reserve_cus, gap_ticks=args.gap_ticks,
                                         tap_every=max(1, args.tap_every), overlap_chunks=args.overlap_chunks)
    f = head["f"]
    cg = head["solver"] == "cg"

    also = []
    if not args.no_also:
        plan = []
        if args.workload == "c2_ml25m_syn" and not args.factors and args.solver == "cholesky":
            plan.append(("c2_ml25m_syn", 128, "cholesky", shards, 0))
            plan.append(("c2_ml25m_syn", 128, "cg", shards, -1))           # -1: the default count for this f
            plan.append(("c3_shard_syn", None, "cholesky", None, 0))
            plan.append(("c3_shard_syn", None, "cg", "prev", -1))
            plan.append(("c3_shard_syn", None, "cg", "prev", 3))       # the truncation other ALS libraries default to
            # the whole 1 B-entry matrix: T_1 on one GPU, the strong-scaled run on several
            plan.append(("c3_full_syn", None, "cholesky", None, 0))
        prev = None
        for (wname, ff, solver, sh, cgs) in plan:
            try:
                r, prev, _ = measure(ctx, wname, ff, solver, cgs if cgs > 0 else args.cg_steps, "auto", args.accumulate, args.row_kernel,
                                     args.also_steps, 2, shards=prev if sh == "prev" else sh, short_rows=args.short_rows,
                                     reserve_cus=args.reserve_cus, gap_ticks=args.gap_ticks)
                also.append({"workload": r["workload"], "f": r["f"], "solver": r["solver"], "cg_steps": r["cg_steps"],
                             "steps": r["steps"],
                             "nnz_per_s": r["value"], "ms_per_step": r["ms_per_step"], "launch_ms": r["launch_ms"],
                             "achieved_GBs": r["achieved"], "frac": r["achieved"] / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": r["bytes_per_launch"], "kernel": r["kernel"],
                             "traffic": pmc_traffic("%s_f%d%s" % (wname, r["f"], "" if solver != "cg" else
                                                                  "_cg" if cgs <= 0 else "_cg%d" % cgs)),
                             "rows_this_rank": {"user": r["user_rows"], "item": r["item_rows"]},
                             "short_rows": r["short_rows"]})
                for k in ("exchange", "scaling"):
                    if k in r:
                        also[-1][k] = r[k]
                if "mfma" in r:
                    also[-1]["mfma"] = r["mfma"]
            except Exception as exc:            # a failing extra point must not cost the headline line
                also.append({"workload": wname, "f": ff, "solver": solver, "cg_steps": cgs or None, "error": repr(exc)})
                prev = None
            torch.cuda.empty_cache()

        if plan and world == 1:
            # the eight-GPU configuration as one of its ranks runs it (long item rows, the whole user table)
            prev = None                     # drops the last reference to the c3 shards
            torch.cuda.empty_cache()
            state = None
            for (solver, cgs) in (("cholesky", 0), ("cg", args.cg_steps)):        # None: the default count for f
                try:
                    r, state = measure_rank_share(ctx, "c3_rank_syn", solver, cgs, args.accumulate, args.row_kernel,
                                                  args.also_steps, 2, state, short_rows=args.short_rows)
                    also.append(r)
                except Exception as exc:
                    also.append({"workload": "c3_rank_syn", "solver": solver, "error": repr(exc)})
            del state
            torch.cuda.empty_cache()
            # and the weak-scaled headline matrix as one of its eight ranks runs it (f=64: its user block, its 7 380
            # items over all 1.3 M users)
            try:
                r, state = measure_rank_share(ctx, "c2_rank_syn", "cholesky", None, args.accumulate, args.row_kernel,
                                              max(args.also_steps, 5), 2, None, short_rows=args.short_rows)
                also.append(r)
            except Exception as exc:
                also.append({"workload": "c2_rank_syn", "solver": "cholesky", "error": repr(exc)})
            del state
            torch.cuda.empty_cache()

    extra = {}
    if not args.no_also and world == 1 and args.workload == "c2_ml25m_syn" and not args.factors:
        for key, fn in (("gram", measure_gram), ("serving_topk", measure_topk)):
            try:
                extra[key] = fn(ctx.dev)
            except Exception as exc:
                extra[key] = {"error": repr(exc)}
            torch.cuda.empty_cache()

    ranks = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                "uuid": str(getattr(torch.cuda.get_device_properties(ctx.dev), "uuid", ""))}
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, mine)

    if rank == 0:
        tkey = args.workload + "_f%d" % f + ("_cg" if cg else "") + ("_fp32mode" if args.accumulate == "fp32" else "")
        traffic = pmc_traffic(tkey)
        out = {
            "metric": "ALS nnz/sec (2*nnz per iteration; iterations/sec in iters_per_s)",
            "value": head["value"], "unit": "nnz/s", "iters_per_s": args.steps / head["seconds"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if "scaling" in head else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["workload"],
                       "parallelism": "dp%d (user/item row shards, all-gather of factors)" % world},
            "roofline": {"bound": "hbm", "kernel": head["kernel"],
                         "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head["achieved"] / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": ("profiles/pmc_traffic.json[%s]: a separate rocprofv3 --pmc run of this "
                                            "command (FETCH_SIZE x 2 + WRITE_SIZE per launch), not measured in this run" % tkey)
                         if traffic is not None else None,
                         "avg_launch_ms": (head["launch_ms"]["user_sweep"] + head["launch_ms"]["item_sweep"]) / 2,
                         "launch_ms": head["launch_ms"],
                         "algorithmic_bytes_per_launch": head["bytes_per_launch"]},
        }
        if "mfma" in head:
            out["roofline"]["mfma_view"] = head["mfma"]
        out["roofline"]["short_rows"] = head["short_rows"]
        if world > 1:
            out["distributed"] = {"backend": backend, "world_size": torch.distributed.get_world_size(), "ranks": ranks}
        for k in ("exchange", "scaling"):
            if k in head:
                out[k if k == "exchange" else "strong_scaling"] = head[k]
        avg_ms = (head["launch_ms"]["user_sweep"] + head["launch_ms"]["item_sweep"]) / 2
        out["roofline"]["achieved_kind"] = "algorithmic bytes per launch / launch time (gathers served by caches count)"
        if traffic is not None:
            out["roofline"]["hbm_measured_GBs"] = traffic / (avg_ms * 1e-3) / 1e9
            out["roofline"]["hbm_measured_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if "mfma" in head:
            out["roofline"]["bound"] = head["mfma"]["bound_by"]      # from the flops per byte of this workload
        if also:
            out["also"] = also
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline and Xh is not None:
            out["cpu_baseline"] = cpu_baseline(shards[0], shards[1], Xh, Yh, wl, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
