"""Can a kernel of another stream run beside the persistent row kernel?  (one GPU; no collective involved)

The multi-GPU engine issues the all-gather of row chunk c on a side stream while the sweep of chunk c + 1 runs.
sweep_stream_kernel (64 < f <= 128) keeps one wavefront with all 512 registers on every SIMD, so a kernel that
becomes ready beside it should find no room until it ends - unless the sweep leaves compute units free
(als_sweep_opts_t.reserve_cus).  This probe measures exactly that on one GPU, with a copy kernel standing in for the
collective: user half-sweep of the 1.25 M x 1 M block (f = 128) on the main stream, a copy of `--mb` megabytes on a
side stream submitted `--delay-ms` after the sweep was queued, i.e. while its persistent kernel runs; reported per reserve_cus: the sweep's duration and the time from
the sweep's start to the copy's end (alone, the copy takes `copy_alone_ms`).  Two stand-ins: the runtime's own copy
kernel (a handful of registers: it fits into the 40 registers the stream kernel leaves on a SIMD) and
tools/ubench/probe_copy.hip with the footprint of an RCCL kernel (`--blocks` workgroups of 256 threads, 288 vector
registers, 16 KB of LDS: one workgroup fills a compute unit, as rcclGenericKernel does), built with the hipcc line in that file.

    python tools/overlap_probe.py [--reserve 0 4 8 16 32] [--mb 640] [--blocks 32]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", type=int, nargs="*", default=[0, 4, 8, 16, 32])
    ap.add_argument("--mb", type=int, default=640)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--threads", type=int, default=256, help="threads per workgroup of the stand-in (64, 128 or 256)")
    ap.add_argument("--factors", type=int, default=128,
                    help="factor count; at f <= 64 the sweep is the one-wave-per-row kernel with three waves per SIMD")
    ap.add_argument("--high-priority", action="store_true", help="the side stream is created with high priority")
    ap.add_argument("--gap-event", action="store_true",
                    help="third experiment: the main stream waits for an event the side stream records in front of its copy")
    ap.add_argument("--gap-cycles", type=int, default=0,
                    help="third experiment: a spin kernel of that many clock ticks on the main stream in front of the second sweep")
    ap.add_argument("--pattern-passes", type=int, default=4, help="third experiment: passes of the stand-in (its duration)")
    ap.add_argument("--held", type=int, nargs="*", default=[0, 32, 64],
                    help="second experiment: workgroups of the stand-in that are resident before the sweep starts")
    ap.add_argument("--hold-passes", type=int, default=8, help="... and how many passes over the buffer they make")
    ap.add_argument("--delay-ms", type=float, default=4.0,
                    help="host-side wait between queueing the sweep and the copy (the sweep takes 20-30 ms)")
    args = ap.parse_args()
    from recommender_amd import data, ops
    dev = torch.device("cuda", 0)
    wl = data.WORKLOADS["c3_shard_syn"]
    nb, M, f = wl["n_users"], wl["n_items"], args.factors
    keys, r = data.synth_pairs(nb, M, wl["nnz"], seed=1234, device=dev, item_seed=1234)
    U = data.csr_from_sorted_keys(keys, r, nb, M)
    del keys, r
    csr = ops.DeviceCSR(U[0], U[1], U[2], (nb, M), dev)
    g = torch.Generator(device=dev); g.manual_seed(42)
    X0 = torch.randn((nb, f), generator=g, device=dev) / f ** 0.5
    Y = torch.randn((M, f), generator=g, device=dev) / f ** 0.5
    cache = ops.gram(Y, wl["lamb"])
    can_short = ops.short_rows_max_entries(f) > 0
    if can_short:
        ops.whiten(Y, cache)
    src = torch.empty(args.mb << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream(device=dev, priority=-1 if args.high_priority else 0)
    main_s = torch.cuda.current_stream(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    import ctypes
    probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libprobe_copy.so"))
    probe.probe_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p]

    def copy(kind, blocks=None, passes=1):
        if kind == "runtime":
            dst.copy_(src)
        else:
            rc = probe.probe_copy_launch(src.data_ptr(), dst.data_ptr(), src.numel(), blocks or args.blocks, args.threads,
                                         passes, torch.cuda.current_stream(dev).cuda_stream)
            assert rc == 0, rc

    out = {"f": f, "high_priority_side_stream": args.high_priority, "copy_mb": args.mb, "blocks": args.blocks, "threads": args.threads, "delay_ms": args.delay_ms, "runs": []}
    for kind in ("runtime", "rccl_like"):
        copy(kind); torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record(); copy(kind); b.record(); torch.cuda.synchronize()
        out["copy_alone_ms_" + kind] = a.elapsed_time(b)
    for kind, short in (("runtime", True), ("rccl_like", True), ("rccl_like", False)):
        if short and not (can_short and f > 64):
            continue
        for res in args.reserve:
            best = None
            for _ in range(args.reps + 1):
                X = X0.clone()
                torch.cuda.synchronize()
                e0, e1, c1 = ev(), ev(), ev()
                e0.record(main_s)
                side.wait_event(e0)
                ops.half_sweep(csr, Y, cache, X, wl["alpha"], opts=ops.sweep_opts(reserve_cus=res), short_rows=short)
                e1.record(main_s)
                time.sleep(args.delay_ms * 1e-3)      # the copy is submitted while the persistent kernel is running
                with torch.cuda.stream(side):
                    copy(kind)
                    c1.record(side)
                torch.cuda.synchronize()
                t = (e0.elapsed_time(e1), e0.elapsed_time(c1))
                best = t if best is None or t[1] < best[1] else best      # the first pass warms up
            out["runs"].append({"copy": kind, "short_rows": short, "reserve_cus": res, "sweep_ms": best[0],
                                "copy_done_after_ms": best[1]})
            print(json.dumps(out["runs"][-1]), flush=True)
    # The other order: a kernel of another stream is already resident on `held` compute units when the sweep starts and
    # stays for several milliseconds (a collective that began first).  The stream kernel's workgroups that find no
    # unit start late; with rows handed out while the kernel runs they take fewer rows and the sweep loses about the
    # share of the chip that was held - fixed shares would make it wait for the late workgroups' whole share.
    for held in args.held:
        for short in (True, False):
            if short and not (can_short and f > 64):
                continue
            best = None
            for _ in range(args.reps + 1):
                X = X0.clone()
                torch.cuda.synchronize()
                e0, e1, h0, h1 = ev(), ev(), ev(), ev()
                with torch.cuda.stream(side):
                    h0.record(side)
                    if held:
                        copy("rccl_like", blocks=held, passes=args.hold_passes)
                    h1.record(side)
                time.sleep(0.0005)
                e0.record(main_s)
                ops.half_sweep(csr, Y, cache, X, wl["alpha"], opts=ops.sweep_opts(), short_rows=short)
                e1.record(main_s)
                torch.cuda.synchronize()
                t = (e0.elapsed_time(e1), h0.elapsed_time(h1))
                best = t if best is None or t[0] < best[0] else best
            out["runs"].append({"held_workgroups": held, "short_rows": short, "sweep_ms": best[0], "held_for_ms": best[1]})
            print(json.dumps(out["runs"][-1]), flush=True)
    # The engine's own pattern: everything is queued ahead of time - sweep of chunk c, an event, sweep of chunk c + 1
    # on the main stream; on the side stream the exchange of chunk c, which waits for that event.  Exchange and next
    # sweep become runnable at the same instant, at a kernel boundary, when the chip drains.
    for res in args.reserve:
        for short in (True, False):
            if short and not (can_short and f > 64):
                continue
            best = None
            for _ in range(args.reps + 1):
                X = X0.clone()
                torch.cuda.synchronize()
                e0, e1, e2, c1 = ev(), ev(), ev(), ev()
                e0.record(main_s)
                ops.half_sweep(csr, Y, cache, X, wl["alpha"], opts=ops.sweep_opts(), short_rows=short)
                e1.record(main_s)
                side.wait_event(e1)
                with torch.cuda.stream(side):
                    released = side.record_event() if args.gap_event else None
                    copy("rccl_like", passes=args.pattern_passes)
                    c1.record(side)
                if released is not None:
                    main_s.wait_event(released)             # main stream: goes on when the side stream has been released
                if args.gap_cycles:
                    torch.cuda._sleep(args.gap_cycles)      # main stream: a one-wave kernel that spins that long
                ops.half_sweep(csr, Y, cache, X, wl["alpha"], opts=ops.sweep_opts(reserve_cus=res), short_rows=short)
                e2.record(main_s)
                torch.cuda.synchronize()
                t = (e0.elapsed_time(e1), e1.elapsed_time(e2), e1.elapsed_time(c1))
                best = t if best is None or t[2] < best[2] else best
            out["runs"].append({"pattern": "sweep, event, sweep || exchange after the event", "short_rows": short,
                                "reserve_cus_second_sweep": res, "gap_cycles": args.gap_cycles, "gap_event": args.gap_event, "first_sweep_ms": best[0], "second_sweep_ms": best[1],
                                "exchange_done_after_event_ms": best[2]})
            print(json.dumps(out["runs"][-1]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
