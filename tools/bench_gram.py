#!/usr/bin/env python3
"""MFMA utilisation of the Gram kernels: time als_gram_f32 on a resident factor matrix and
report achieved fp32 matrix-core TFLOP/s against the 157.3 TFLOP/s dense fp32 MFMA peak.
Counted flops are the ones the kernel issues on the matrix cores (symmetric tiles only):
n * T * 2*16*16*4/4 with T = VW(VW+1)/2 tiles for f <= 128, 16 tiles for each of the
NB(NB+1)/2 block pairs above."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommender_amd import ops  # noqa: E402

PEAK = 157.3


def tiles(f):
    if f <= 128:
        vw = 1 if f <= 16 else 2 if f <= 32 else 4 if f <= 64 else 8
        return vw * (vw + 1) // 2
    nb = (f + 63) // 64
    return nb * (nb + 1) // 2 * 16


def main():
    out = []
    for n, f in [(59047, 64), (162541, 64), (1000000, 64), (4000000, 64), (1000000, 128), (4000000, 128),
                 (1000000, 256)]:
        Y = torch.randn(n, f, device="cuda") * 0.1
        gc = ops.gram(Y, 0.01)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ops.gram(Y, 0.01, out=gc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = n * tiles(f) * 512.0
        out.append({"n": n, "f": f, "ms": ms, "mfma_tflops": flops / ms / 1e9, "mfma_util": flops / ms / 1e9 / PEAK,
                    "read_GBs": n * f * 4 / ms / 1e6})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
