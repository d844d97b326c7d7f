#!/usr/bin/env python3
"""Serving-path benchmark (the 10 M-user top-100 serving configuration): top-100 items for a block of users against
1 M items from f=128 factors, all resident.  Reports users/s and the split between the score
kernel alone writing every score (bf16-split products on the matrix cores; store bound) and the fused
path, which never stores them.  --filter-rated: every user's training items (a synthetic CSR with --avg-rated
entries a user, resident on the device, indexed by user id inside the kernels) are dropped from the ranking, as
RecommenderService.py:16-29 does after the fact."""
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
    ap.add_argument("--filter-rated", action="store_true")
    ap.add_argument("--avg-rated", type=int, default=100)
    ap.add_argument("--skip-materialised", action="store_true", help="do not time the score kernel writing every score")
    ap.add_argument("--prepared", action="store_true",
                    help="rank against the item table prepared once (als_items_prepare_f32: fp16 x 2 operands)")
    a = ap.parse_args()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.randn((a.users, a.factors), generator=g, device="cuda") / a.factors ** 0.5
    Y = torch.randn((a.items, a.factors), generator=g, device="cuda") / a.factors ** 0.5
    users = torch.arange(a.users, device="cuda")
    if a.prepared:
        Y = ops.prepare_items(Y)
    ex, ex_nnz = None, 0
    if a.filter_rated:
        from recommender_amd import data
        keys, r = data.synth_pairs(a.users, a.items, a.users * a.avg_rated, 77, "cuda")
        U = data.csr_from_sorted_keys(keys, r, a.users, a.items)
        ex, ex_nnz = (U[0], U[1]), int(U[1].numel())
        del keys, r
    ops.recommend_topk(X, Y, users[: a.block_users], a.k, exclude_csr=ex, block_users=a.block_users)       # warm-up
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    idx, val = ops.recommend_topk(X, Y, users, a.k, exclude_csr=ex, block_users=a.block_users)
    e[1].record()
    if ex is not None:          # no training item of a user among its recommendations (sampled users)
        torch.cuda.synchronize()
        for u in range(0, a.users, max(1, a.users // 200)):
            rated = ex[1][int(ex[0][u]):int(ex[0][u + 1])]
            assert not torch.isin(idx[u], rated).any(), "user %d was recommended an item it rated" % u
    rows = min(a.block_users, 1024)
    torch.cuda.synchronize()
    e[2].record()
    if not a.skip_materialised:
        buf = torch.empty((rows, a.items), device="cuda")
        for s in range(0, a.users, rows):
            ops.scores(X, Y, users[s:s + rows], a.items, out=buf)
    e[3].record()
    torch.cuda.synchronize()
    total, gemm = e[0].elapsed_time(e[1]), max(e[2].elapsed_time(e[3]), 1e-9)
    flops = 2.0 * a.users * a.items * a.factors
    print(json.dumps({"users": a.users, "items": a.items, "f": a.factors, "k": a.k, "ms_total": total,
                      "users_per_s": a.users / total * 1e3, "filter_rated": bool(a.filter_rated), "excluded_entries": ex_nnz,
                      "items_form": "prepared (fp16 x 2)" if a.prepared else "float32 (bf16 x 3)",
                      "ms_scores_materialised": None if a.skip_materialised else gemm,
                      "fused_tflops_fp32_equivalent": flops / total / 1e9,
                      "scores_written_gb_per_s": None if a.skip_materialised else 4.0 * a.users * a.items / gemm / 1e6}),
          flush=True)


if __name__ == "__main__":
    main()
