"""Randomised shake-out of the fused top-k against the two-call path (scores + exact row top-k) on real-valued
factors: random user/item/factor counts, k, user subsets, leading dimensions, per-user exclusions.
usage (on a GPU box): python tools/fuzz_topk.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import recommender_amd as ra
rng = np.random.default_rng(11)
bad = 0
for case in range(40):
    f = int(rng.choice([4, 8, 20, 33, 64, 100, 128, 200]))
    M = int(rng.choice([500, 16384, 16385, 40000, 70000, 300000]))
    nu = int(rng.integers(1, 300)); k = int(min(M, rng.choice([1, 10, 100, 200, 1024])))
    ld = f + int(rng.choice([0, 4]))
    Xs = torch.from_numpy(rng.normal(size=(nu + 50, ld)).astype(np.float32)).cuda()
    Ys = torch.from_numpy((rng.normal(size=(M, ld)) * rng.choice([1.0, 1e-3, 30.0])).astype(np.float32)).cuda()
    X, Y = Xs[:, :f], Ys[:, :f]
    users = torch.from_numpy(rng.permutation(nu + 50)[:nu]).cuda()
    ex = None
    if rng.random() < 0.6:
        rows = [np.sort(rng.choice(M, size=int(rng.integers(0, min(M - k, 300) + 1)), replace=False)).astype(np.int32) for _ in range(nu + 50)]
        ptr = np.zeros(nu + 51, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
        ex = (ptr, np.concatenate(rows) if ptr[-1] else np.zeros(0, np.int32))
    i0, v0 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex, fused=False)
    i1, v1 = ra.ops.recommend_topk(X, Y, users, k, exclude_csr=ex, block_users=int(rng.choice([64, 4096])))
    ok = torch.equal(i0, i1) and torch.equal(v0, v1)
    if f <= 128 and f % 4 == 0:            # the same against the prepared item table
        P = ra.prepare_items(Y)
        j0, w0 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex, fused=False)
        j1, w1 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex, block_users=int(rng.choice([64, 4096])))
        ok = ok and torch.equal(j0, j1) and torch.equal(w0, w1)
        # and the two arithmetics rank alike wherever scores are clearly apart: the k-th scores agree to rounding
        ok = ok and bool(((w0[:, -1] - v0[:, -1]).abs() <= 1e-5 * v0[:, 0].abs().clamp_min(1e-30)).all())
    bad += not ok
    print(case, "f", f, "M", M, "nu", nu, "k", k, "ld", ld, "ex", ex is not None, "ok" if ok else "MISMATCH", flush=True)
print("MISMATCHES", bad)
