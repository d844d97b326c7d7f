"""Can a kernel of another stream run beside the persistent row kernel?  (one GPU; no collective involved)

The multi-GPU engine issues the all-gather of row chunk c on a side stream while the sweep of chunk c + 1 runs.
sweep_stream_kernel (64 < f <= 128) keeps one wavefront with all 512 registers on every SIMD, so a kernel that
becomes ready beside it should find no room until it ends - unless the sweep leaves compute units free
(als_sweep_opts_t.reserve_cus).  This probe measures exactly that on one GPU, with a device-to-device copy kernel
standing in for the collective: user half-sweep of the 1.25 M x 1 M block (f = 128) on the main stream, a copy of
`--mb` megabytes on a side stream that may start when the sweep starts; reported per reserve_cus: the sweep's
duration and the time from the sweep's start to the copy's end (alone, the copy takes `copy_alone_ms`).

    python tools/overlap_probe.py [--reserve 0 4 8 16 32] [--mb 640]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", type=int, nargs="*", default=[0, 4, 8, 16, 32])
    ap.add_argument("--mb", type=int, default=640)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from recommender_amd import data, ops
    dev = torch.device("cuda", 0)
    wl = data.WORKLOADS["c3_shard_syn"]
    nb, M, f = wl["n_users"], wl["n_items"], wl["f"]
    keys, r = data.synth_pairs(nb, M, wl["nnz"], seed=1234, device=dev, item_seed=1234)
    U = data.csr_from_sorted_keys(keys, r, nb, M)
    del keys, r
    csr = ops.DeviceCSR(U[0], U[1], U[2], (nb, M), dev)
    g = torch.Generator(device=dev); g.manual_seed(42)
    X0 = torch.randn((nb, f), generator=g, device=dev) / f ** 0.5
    Y = torch.randn((M, f), generator=g, device=dev) / f ** 0.5
    cache = ops.gram(Y, wl["lamb"])
    ops.whiten(Y, cache)
    src = torch.empty(args.mb << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # the copy alone
    dst.copy_(src); torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record(); dst.copy_(src); b.record(); torch.cuda.synchronize()
    out = {"copy_mb": args.mb, "copy_alone_ms": a.elapsed_time(b), "runs": []}
    for short in (True, False):
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
                with torch.cuda.stream(side):
                    dst.copy_(src)
                    c1.record(side)
                torch.cuda.synchronize()
                t = (e0.elapsed_time(e1), e0.elapsed_time(c1))
                best = t if best is None or t[1] < best[1] else best      # the first pass warms up
            out["runs"].append({"short_rows": short, "reserve_cus": res, "sweep_ms": best[0], "copy_done_after_ms": best[1]})
            print(json.dumps(out["runs"][-1]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
