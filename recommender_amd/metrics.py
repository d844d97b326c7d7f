"""Evaluation metrics of the reference, host side (they run on a few hundred users per epoch).

  compute_ap / compute_auc   reclibwh/utils/eval_utils.py:5-37
  rmse                       reclibwh/utils/utils.py:58-59
  users_to_test              reclibwh/core/ALS.py:434-437
  mean_average_precision     AUCEval.evaluate, reclibwh/core/EvalProto.py:28-62  (the "AUC" the
                             reference logs to eval.csv is this MAP over the first 200 items)

The reference filters training items with np.setdiff1d, which also SORTS the remaining
recommendations by item id; ranks are therefore positions in the id-sorted list.  That is
reproduced here, because eval.csv values are only comparable if it is.
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
