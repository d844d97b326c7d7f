#!/usr/bin/env python3
"""Serving-path benchmark (BASELINE configs[4]): top-100 items for a block of users against
1 M items from f=128 factors, all resident.  Reports users/s and the split between the score
kernel alone writing every score (bf16-split products on the matrix cores; store bound) and the fused
path, which never stores them."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommender_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=8192)
    ap.add_argument("--items", type=int, default=1000000)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--block-users", type=int, default=4096)
    a = ap.parse_args()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.randn((a.users, a.factors), generator=g, device="cuda") / a.factors ** 0.5
    Y = torch.randn((a.items, a.factors), generator=g, device="cuda") / a.factors ** 0.5
    users = torch.arange(a.users, device="cuda")
    ops.recommend_topk(X, Y, users[: a.block_users], a.k, block_users=a.block_users)       # warm-up
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    idx, val = ops.recommend_topk(X, Y, users, a.k, block_users=a.block_users)
    e[1].record()
    rows = min(a.block_users, 1024)
    buf = torch.empty((rows, a.items), device="cuda")
    torch.cuda.synchronize()
    e[2].record()
    for s in range(0, a.users, rows):
        ops.scores(X, Y, users[s:s + rows], a.items, out=buf)
    e[3].record()
    torch.cuda.synchronize()
    total, gemm = e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])
    flops = 2.0 * a.users * a.items * a.factors
    print(json.dumps({"users": a.users, "items": a.items, "f": a.factors, "k": a.k, "ms_total": total,
                      "users_per_s": a.users / total * 1e3, "ms_scores_materialised": gemm,
                      "fused_tflops_fp32_equivalent": flops / total / 1e9,
                      "scores_written_gb_per_s": 4.0 * a.users * a.items / gemm / 1e6}), flush=True)


if __name__ == "__main__":
    main()
