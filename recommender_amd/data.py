"""Synthetic implicit-feedback matrices shaped like the MovieLens data the reference trains on
(no dataset ships with the reference: data/READme.md:1).

The generator is plain torch, runs on the GPU or the CPU, and is seeded, so every rank of a
multi-GPU job can rebuild any block of the global matrix by itself:

  * item popularity  ~ rank^-item_exp   (Zipf), ranks assigned by a seeded permutation
  * user activity    ~ log-normal(0, user_sigma)
  * (user, item) pairs are sampled independently, de-duplicated and sorted (canonical CSR)
  * ratings are uniform over {0.5, 1.0, ..., 5.0} - all > 0, as in MovieLens
"""
import numpy as np
import torch


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def synth_pairs(n_users, n_items, nnz_target, seed, device="cpu", item_exp=0.9, user_sigma=1.0, item_seed=None):
    """Sorted unique keys u * n_items + i (int64) and their ratings (float32).

    item_seed fixes which items are popular independently of `seed`, so that several user
    blocks generated with different seeds share one item-popularity profile."""
    device = torch.device(device)
    gi = _gen(seed if item_seed is None else item_seed, device)
    item_w = torch.arange(1, n_items + 1, dtype=torch.float64, device=device).pow(-item_exp)
    item_w = item_w[torch.randperm(n_items, generator=gi, device=device)]
    g = _gen(seed, device)
    user_w = torch.exp(user_sigma * torch.randn(n_users, generator=g, device=device, dtype=torch.float64))
    icdf = torch.cumsum(item_w / item_w.sum(), 0)
    ucdf = torch.cumsum(user_w / user_w.sum(), 0)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = int(nnz_target)
    # oversample a little, de-duplicate, repeat until the target is met (2-3 rounds)
    for _ in range(8):
        need = want - keys.numel()
        if need <= 0:
            break
        m = int(need * 1.15) + 1024
        u = torch.searchsorted(ucdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(max=n_users - 1)
        i = torch.searchsorted(icdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(max=n_items - 1)
        keys = torch.unique(torch.cat([keys, u * n_items + i]))
    if keys.numel() > want:  # drop a seeded random subset, keep sorted order
        keep = torch.randperm(keys.numel(), generator=g, device=device)[:want]
        keys = keys[torch.sort(keep).values]
    r = torch.randint(1, 11, (keys.numel(),), generator=g, device=device).to(torch.float32) * 0.5
    return keys, r


def csr_from_sorted_keys(keys, vals, n_rows, n_cols):
    """CSR arrays (int64 rowptr, int32 col, float32 val) from sorted keys row * n_cols + col."""
    rows = torch.div(keys, n_cols, rounding_mode="floor")
    cols = (keys - rows * n_cols).to(torch.int32)
    counts = torch.bincount(rows, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=keys.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, cols, vals


def transpose_keys(keys, vals, n_rows, n_cols):
    """Keys/values of the transposed matrix, sorted (col-major order of the original)."""
    rows = torch.div(keys, n_cols, rounding_mode="floor")
    cols = keys - rows * n_cols
    tkeys, order = torch.sort(cols * n_rows + rows)
    return tkeys, vals[order]


def synth_matrix(n_users, n_items, nnz_target, seed=1234, device="cpu", **kw):
    """(U, Ut): CSR arrays of the user x item matrix and of its transpose, on `device`."""
    keys, r = synth_pairs(n_users, n_items, nnz_target, seed, device, **kw)
    U = csr_from_sorted_keys(keys, r, n_users, n_items)
    tk, tr = transpose_keys(keys, r, n_users, n_items)
    Ut = csr_from_sorted_keys(tk, tr, n_items, n_users)
    return U, Ut


def to_scipy(csr, shape):
    import scipy.sparse as sps
    rowptr, col, val = (t.cpu().numpy() for t in csr)
    return sps.csr_matrix((val.astype(np.float64), col, rowptr), shape=shape)


# Named workloads (the benchmark configurations; dimensions are the GroupLens ones)
WORKLOADS = {
    # ml-latest-small shaped: 610 x 9724, 80 668 training ratings, f=20, 15 iterations
    "c1_ml_small_syn": dict(n_users=610, n_items=9724, nnz=80668, f=20, lamb=0.01, alpha=40.0, iters=15),
    # ml-25m shaped: 162 541 x 59 047, 25 000 095 ratings, f=64
    "c2_ml25m_syn": dict(n_users=162541, n_items=59047, nnz=25000095, f=64, lamb=0.01, alpha=40.0, iters=15),
    # one GPU's eighth of the 10 M x 1 M, 1 B-entry matrix (f=128): 1.25 M users over all 1 M items.  Both
    # factor tables (0.5 GB and 0.64 GB) exceed the 256 MB Infinity Cache, so the gathers come from HBM.
    # With several GPUs the blocks' entry counts differ by up to +-10 % (block_jitter) and the user shards are cut by
    # stored entries (balance_users), not at the block boundaries: ranks own different numbers of rows.
    "c3_shard_syn": dict(n_users=1250000, n_items=1000000, nnz=125000000, f=128, lamb=0.01, alpha=40.0, iters=15,
                         init="device", block_jitter=0.1, balance_users=True),
    # the whole 10 M x 1 M, 1 B-entry matrix (8 such blocks, entry counts jittered +-10 %), whatever the number of GPUs:
    # on one GPU it is T_1 of the strong-scaling efficiency T_1 / (g T_g); on g GPUs users and items are cut into g
    # shards of about equal stored entries
    "c3_full_syn": dict(n_users=1250000, n_items=1000000, nnz=125000000, f=128, lamb=0.01, alpha=40.0, iters=15,
                        init="device", block_jitter=0.1, balance_users=True, blocks=8),
    # the same construction a hundred times smaller (tests and one-GPU rehearsals of the strong-scaled run)
    "c3_full_small_syn": dict(n_users=12500, n_items=10000, nnz=1250000, f=128, lamb=0.01, alpha=40.0, iters=15,
                              init="device", block_jitter=0.1, balance_users=True, blocks=8),
    # what one rank of the eight-GPU run of that matrix executes: its 1.25 M users over all 1 M items, and its
    # 125 000 items over all 10 M users (item rows eight times longer, the whole 5 GB user table gathered from)
    "c3_rank_syn": dict(n_users=1250000, n_items=1000000, nnz=125000000, f=128, lamb=0.01, alpha=40.0, of_world=8),
    # the same for the weak-scaled 25 M-entry matrix (8 user blocks of 162 541 over the same 59 047 items, f=64): a rank's
    # user block, and its 7 380 items over all 1.3 M users (rows of 3 400 entries on average, one wavefront each)
    "c2_rank_syn": dict(n_users=162541, n_items=59047, nnz=25000095, f=64, lamb=0.01, alpha=40.0, of_world=8),
    # one GPU's eighth of the 50 M x 5 M, 4 B-entry matrix (f=256, conjugate-gradient solver)
    "c4_shard_syn": dict(n_users=6250000, n_items=5000000, nnz=500000000, f=256, lamb=0.01, alpha=40.0, iters=15,
                         init="device", solver="cg"),
}
