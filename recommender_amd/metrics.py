"""Evaluation metrics of the reference.

  compute_ap / compute_auc        reclibwh/utils/eval_utils.py:5-37   (host, numpy: the reference's own formulas)
  rmse                            reclibwh/utils/utils.py:58-59
  users_to_test                   reclibwh/core/ALS.py:434-437
  mean_average_precision          AUCEval.evaluate, reclibwh/core/EvalProto.py:28-62  (the "AUC" the reference
                                  logs to eval.csv is this MAP over the first 200 items) - host loop over users
  mean_average_precision_device   the same number from torch operations on the device that holds the
                                  recommendations (what RecommenderALS.evaluate uses): no per-user loop, no copy of
                                  the ranked lists to the host
  pair_keys / relevant_test_keys  the sorted (user, item) keys the device form looks memberships up in

The reference filters training items with np.setdiff1d(rec, train, assume_unique=True), which keeps the ranked
order of what remains; a relevant item's rank is its 1-based position in that filtered list.
"""
import numpy as np


def rmse(x, y):
    return float(np.sqrt(np.mean(np.power(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64), 2))))


def compute_auc(rec, test_user, train_user):
    rec_filt = np.setdiff1d(rec, train_user, assume_unique=True)
    pos = np.nonzero(np.isin(rec_filt, test_user))[0]
    p, n = len(pos), len(rec_filt)
    if p == 0:
        return -1
    if n - p == 0:
        return 1
    fp = pos.astype(float) - np.arange(p)
    return 1 - (np.cumsum(fp)[-1] / (n - p)) / p


def compute_ap(rec, test_user, train_user):
    if len(test_user) == 0:
        return -1
    rec_filt = np.setdiff1d(rec, train_user, assume_unique=True)
    pos = np.nonzero(np.isin(rec_filt, test_user))[0] + 1
    p = len(pos)
    if p == 0:
        return 0.
    return float(np.sum(np.arange(1, p + 1) / pos.astype(float)) / p)


def users_to_test(train_csr, max_num_ratings=50, max_users=300):
    """Users with fewer than max_num_ratings training ratings, thinned to about max_users
    (the selection of the reference's training script)."""
    num = np.diff(train_csr.indptr)
    u = np.nonzero(num < max_num_ratings)[0]
    return u[np.arange(0, len(u), max(len(u) // max_users, 1))]


def mean_average_precision(recs, users, train_csr, test_csr, med_score=3.0, lim=200):
    """recs[k] = ranked item ids for users[k] (at least `lim` of them).  Relevant = test items
    rated above med_score; training items are filtered from the list; users without a relevant
    test item are skipped."""
    vals = []
    for k, u in enumerate(users):
        ti = test_csr.indices[test_csr.indptr[u]:test_csr.indptr[u + 1]]
        tr = test_csr.data[test_csr.indptr[u]:test_csr.indptr[u + 1]]
        rel = ti[tr > med_score]
        trn = train_csr.indices[train_csr.indptr[u]:train_csr.indptr[u + 1]]
        ap = compute_ap(np.asarray(recs[k][:lim]), rel, trn)
        if ap < 0:
            continue
        vals.append(ap)
    return float(np.mean(vals)) if vals else float("nan")


# ----------------------------------------------------------------------------- device form
def pair_keys(csr, n_items, device, mask=None):
    """Sorted int64 keys user * n_items + item of the stored entries of a canonical scipy CSR (optionally only those
    where mask is set), as a tensor on `device`."""
    import torch
    rows = np.repeat(np.arange(csr.shape[0], dtype=np.int64), np.diff(csr.indptr))
    keys = rows * int(n_items) + csr.indices.astype(np.int64)
    if mask is not None:
        keys = keys[mask]
    return torch.from_numpy(np.ascontiguousarray(keys)).to(device)


def relevant_test_keys(test_csr, n_items, device, med_score=3.0):
    """(sorted keys of the test entries rated above med_score, number of such entries per user) on `device`."""
    import torch
    mask = test_csr.data > med_score
    rows = np.repeat(np.arange(test_csr.shape[0], dtype=np.int64), np.diff(test_csr.indptr))
    counts = np.bincount(rows[mask], minlength=test_csr.shape[0]).astype(np.int64)
    return pair_keys(test_csr, n_items, device, mask), torch.from_numpy(counts).to(device)


def _member(sorted_keys, q):
    import torch
    if sorted_keys.numel() == 0:
        return torch.zeros_like(q, dtype=torch.bool)
    pos = torch.searchsorted(sorted_keys, q).clamp_(max=sorted_keys.numel() - 1)
    return sorted_keys[pos] == q


def mean_average_precision_device(recs, users, train_keys, rel_keys, rel_counts, n_items, lim=200):
    """mean_average_precision on tensors: recs [U, >= lim] ranked item ids and users [U] (int64, one device),
    train_keys / rel_keys / rel_counts from pair_keys / relevant_test_keys.  Per user (EvalProto.py:41-57,
    eval_utils.py:23-37): the first `lim` recommendations, training items dropped, AP = mean over the relevant items
    found of (their count so far / their position in the filtered list); users without a relevant test item are
    skipped, users whose list holds none of theirs count 0."""
    import torch
    r = recs[:, :lim].to(torch.int64)
    keys = users.to(torch.int64)[:, None] * int(n_items) + r
    kept = ~_member(train_keys, keys)
    rel = kept & _member(rel_keys, keys)
    rank = torch.cumsum(kept, 1).to(torch.float64)              # position in the filtered list (where kept)
    found = torch.cumsum(rel, 1).to(torch.float64)
    ap_sum = torch.where(rel, found / rank.clamp(min=1.0), torch.zeros_like(rank)).sum(1)
    p = found[:, -1] if r.shape[1] else torch.zeros(r.shape[0], dtype=torch.float64, device=r.device)
    ap = torch.where(p > 0, ap_sum / p.clamp(min=1.0), torch.zeros_like(p))
    counted = rel_counts[users.to(torch.int64)] > 0
    if not bool(counted.any()):
        return float("nan")
    return float(ap[counted].mean().item())


def rmse_device(pred, target):
    """rmse (utils.py:58-59) of two tensors on one device, accumulated in float64."""
    import torch
    return float(torch.sqrt(torch.mean((pred.double() - target.double()) ** 2)).item())
