"""CPU oracle for the implicit-ALS hot path.  TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference's algorithm, used by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg as the *checker*.
Nothing in the product package (recommender_amd/) imports this module.

Pinned against golden vectors minted from the reference's own ALSRef class
(tests/golden/make_golden.py -> tests/golden/*.npz; see tests/test_oracle.py).

Each function cites the reference lines it follows (paths relative to the
reference repository root):

  gram                      reclibwh/core/ALS.py:70 and :243
  half_sweep_exact          reclibwh/core/ALS.py:67-88   (ALSRef._run_single_step)
  train_exact               reclibwh/core/ALS.py:122-141 (ALSRef.train)
  init_factors              reclibwh/core/ALS.py:56-65 ("ref") and :248-260 ("tf")
  padded_batch              reclibwh/utils/utils.py:142-173 + data/PresetIterators.py:11-17
  half_sweep_batched_f32    reclibwh/core/ALS.py:160-199 + :201-233 (TF path, float32)
  half_sweep_cg             the same normal equations as ALS.py:82-86, solved by a fixed number of
                            conjugate-gradient steps (Hestenes-Stiefel; warm start from the current
                            row) instead of inv().  The reference has no CG: this restates the
                            published algorithm and is pinned to the golden vectors through its
                            limit - with enough steps it must reproduce half_sweep_exact.
  recommend_full            reclibwh/core/RecAlgos.py:20-32
  cosine_similar            reclibwh/core/RecAlgos.py:38-73
  compute_ap / compute_auc  reclibwh/utils/eval_utils.py:5-37
  rmse                      reclibwh/utils/utils.py:58-59
  map_at_k                  reclibwh/core/EvalProto.py:28-62
"""
import numpy as np


# --------------------------------------------------------------------------- init
def init_factors(N, M, K, convention="ref", seed=42, dtype=np.float64):
    """Initial factors ~ N(0, 1/sqrt(K)) from numpy's legacy global MT19937 stream.

    "ref": ALSRef.__init__ re-seeds before X and again before Y (ALS.py:59-65), so
           X and Y start from the same stream.
    "tf" : als_initizalize seeds once, draws X then Y (ALS.py:255-257).
    """
    if convention == "ref":
        np.random.seed(seed)
        X = np.random.normal(0, 1 / np.sqrt(K), size=(N, K))
        np.random.seed(seed)
        Y = np.random.normal(0, 1 / np.sqrt(K), size=(M, K))
    elif convention == "tf":
        np.random.seed(seed)
        X = np.random.normal(0, 1 / np.sqrt(K), size=(N, K))
        Y = np.random.normal(0, 1 / np.sqrt(K), size=(M, K))
    else:
        raise ValueError(convention)
    return X.astype(dtype), Y.astype(dtype)


# --------------------------------------------------------------------------- exact path
def gram(Y):
    """Y^T Y over all rows of Y (ALS.py:70, ALS.py:243)."""
    return np.tensordot(Y.T, Y, axes=1)


def half_sweep_exact(indptr, indices, data, Yfix, Xold, lamb, alpha, rows=None):
    """ALSRef._run_single_step (ALS.py:67-88) on raw CSR arrays.

    For each row u with stored columns p_idx and ratings rp (stored zeros are dropped,
    as sps.find does at ALS.py:79):
        A = Y2 + Yp^T (alpha * rp[:,None] * Yp) + lamb*I        (ALS.py:82)
        x = inv(A) @ (Yp^T (cup * pp)),  cup = 1 + alpha*rp, pp = 1   (ALS.py:83-86)
    Rows that are not visited keep Xold (Xp = X.copy(), ALS.py:73); `rows` restricts the
    visit to a subset (the fold-in path, ALS.py:375-395).  Arithmetic is float64.
    """
    Yfix = np.asarray(Yfix, dtype=np.float64)
    Xp = np.array(Xold, dtype=np.float64, copy=True)
    K = Yfix.shape[1]
    Y2 = gram(Yfix)
    Lamb = lamb * np.eye(K)
    it = range(Xp.shape[0]) if rows is None else rows
    for u in it:
        s, e = int(indptr[u]), int(indptr[u + 1])
        cols = np.asarray(indices[s:e])
        rp = np.asarray(data[s:e], dtype=np.float64)
        keep = rp != 0
        cols, rp = cols[keep], rp[keep]
        Yp = Yfix[cols, :]
        L = np.linalg.inv(Y2 + np.matmul(Yp.T, alpha * np.multiply(np.expand_dims(rp, axis=1), Yp)) + Lamb)
        cup = 1 + alpha * rp
        Xp[u, :] = np.matmul(L, np.tensordot(Yp.T, cup, axes=1))
    return Xp


def half_sweep_cg(indptr, indices, data, Yfix, Xold, lamb, alpha, steps, rows=None, dtype=np.float64, Y2=None):
    """Per row u, `steps` conjugate-gradient steps on A x = b, started at x = Xold[u], with
        A = Y2 + Yp^T diag(alpha rp) Yp + lamb I,  b = Yp^T (1 + alpha rp)      (ALS.py:82-86)
    A is never formed: A v = Y2 v + Yp^T (alpha rp * (Yp v)) + lamb v.

        r = b - A x;  p = r;  rs = r.r
        repeat steps times:  Ap = A p;  a = rs / p.Ap;  x += a p;  r -= a Ap
                             rs' = r.r;  p = r + (rs'/rs) p;  rs = rs'
    A step with rs == 0 or p.Ap <= 0 changes nothing.  A row without stored entries has the
    exact solution 0 and gets it.  Stored zeros are dropped (ALS.py:79).
    Y2: Yfix^T Yfix computed by the caller (tests at full size hand in only the rows of Yfix that the sampled
    rows touch, and the Gram of the whole table)."""
    Yfix = np.asarray(Yfix, dtype=dtype)
    Xp = np.array(Xold, dtype=dtype, copy=True)
    Y2 = (gram(Yfix) if Y2 is None else np.asarray(Y2)).astype(dtype)
    lamb = dtype(lamb); alpha = dtype(alpha)
    it = range(Xp.shape[0]) if rows is None else rows
    for u in it:
        s, e = int(indptr[u]), int(indptr[u + 1])
        cols = np.asarray(indices[s:e]); rp = np.asarray(data[s:e], dtype=dtype)
        keep = rp != 0
        cols, rp = cols[keep], rp[keep]
        if len(cols) == 0:
            Xp[u] = 0
            continue
        Yp = Yfix[cols]; w = alpha * rp

        def A(v):
            return Y2 @ v + Yp.T @ (w * (Yp @ v)) + lamb * v

        x = Xp[u].copy()
        r = Yp.T @ (1 + w) - A(x)
        p = r.copy(); rs = r @ r
        for _ in range(steps):
            Ap = A(p); pAp = p @ Ap
            if not (rs > 0 and pAp > 0):
                break
            a = rs / pAp
            x = x + a * p; r = r - a * Ap
            rs_new = r @ r
            p = r + (rs_new / rs) * p; rs = rs_new
        Xp[u] = x
    return Xp


def csr_transpose(indptr, indices, data, n_cols):
    """CSR of the transpose with sorted indices (what sps.csr_matrix(U.T) yields, ALS.py:432)."""
    indptr = np.asarray(indptr); indices = np.asarray(indices); data = np.asarray(data)
    n_rows = len(indptr) - 1
    counts = np.bincount(indices, minlength=n_cols)
    tptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(counts, out=tptr[1:])
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    order = np.argsort(indices, kind="stable")
    return tptr, rows[order].astype(np.int32), data[order]


def train_exact(indptr, indices, data, shape, K, lamb, alpha, steps, Xinit, Yinit):
    """ALSRef.train (ALS.py:122-141): `steps` x (X half-sweep, Y half-sweep);
    trace[i] = mean|X_old - X_new| + mean|Y_old - Y_new|."""
    N, M = shape
    X = np.array(Xinit, dtype=np.float64); Y = np.array(Yinit, dtype=np.float64)
    tptr, tind, tdat = csr_transpose(indptr, indices, data, M)
    trace = np.zeros(steps)
    hist = []
    for i in range(steps):
        Xp = half_sweep_exact(indptr, indices, data, Y, X, lamb, alpha)
        trace[i] = np.mean(np.abs(X - Xp)); X = Xp
        Yp = half_sweep_exact(tptr, tind, tdat, X, Y, lamb, alpha)
        trace[i] += np.mean(np.abs(Y - Yp)); Y = Yp
        hist.append((X.copy(), Y.copy()))
    return X, Y, trace, hist


# --------------------------------------------------------------------------- TF-path restatement
def padded_batch(indptr, indices, data, users, pad_val):
    """get_pos_ratings_padded (utils.py:142-173) as ALS_data_iter_preset drives it
    (PresetIterators.py:11-17): rp [B, L] zero padded, yp [B, L] padded with pad_val,
    L = the longest row in the batch, column order = CSR storage order."""
    users = np.asarray(users)
    lens = np.array([int(indptr[u + 1] - indptr[u]) for u in users])
    L = int(lens.max()) if len(lens) else 0
    rp = np.zeros((len(users), L), dtype=np.float64)
    yp = int(pad_val) * np.ones((len(users), L), dtype=np.int64)
    for b, u in enumerate(users):
        s, e = int(indptr[u]), int(indptr[u + 1])
        rp[b, : e - s] = data[s:e]
        yp[b, : e - s] = indices[s:e]
    return rp, yp


def half_sweep_batched_f32(indptr, indices, data, n_cols, Ylayer, Xlayer, lamb, alpha, batchsize=100, rows=None):
    """The TensorFlow path in float32 numpy: run_als_epoch over padded batches
    (ALS.py:201-233) calling run_als_step / als_run_regression (ALS.py:160-199).

    n_cols is the CSR's column count = the pad index (PresetIterators.py:13).  Ylayer has
    the layer's row count (n_cols plus the pad/phantom rows, ALS.py:456); the
    Gram is taken over ALL its rows (ALS.py:243); padded slots gather row `n_cols`
    (PresetIterators.py:13) with rating 0, so they add nothing.  Returns (Xnew, diff)
    with diff = sum|X_old[us] - X_new| / (n*k) (ALS.py:195, :233).  Xlayer is not modified.
    """
    Y = np.asarray(Ylayer, dtype=np.float32)
    X = np.array(Xlayer, dtype=np.float32, copy=True)
    k = Y.shape[1]
    assert Y.shape[0] > n_cols, "the layer needs a pad row at index n_cols (ALS.py:456)"
    Y2 = np.tensordot(Y.T, Y, axes=1).astype(np.float32)
    Lamb = (np.float32(lamb) * np.eye(k, dtype=np.float32))
    a = np.float32(alpha)
    all_rows = np.arange(len(indptr) - 1) if rows is None else np.asarray(rows)
    diff = np.float32(0); n = 0
    for s in range(0, len(all_rows), batchsize):
        us = all_rows[s:s + batchsize]
        rp, yp = padded_batch(indptr, indices, data, us, n_cols)
        r = rp.astype(np.float32)[:, :, None]                       # [B, L, 1]
        Ypp = Y[yp.astype(np.int32)]                                 # [B, L, k]   (ALS.py:185)
        c = np.float32(1.) + a * r                                   # (ALS.py:168)
        p = (r > 0).astype(np.float32)                               # (ALS.py:169)
        Lm = Y2 + np.matmul(Ypp.transpose(0, 2, 1), a * (r * Ypp))   # (ALS.py:171)
        Linv = np.linalg.inv(Lm + Lamb)                              # (ALS.py:172)
        x = np.matmul(Linv, np.matmul(Ypp.transpose(0, 2, 1), c * p))  # (ALS.py:173)
        Xnew = x.reshape(len(us), k).astype(np.float32)
        diff += np.sum(np.abs(X[us] - Xnew), dtype=np.float32)       # (ALS.py:195)
        X[us] = Xnew                                                 # (ALS.py:197)
        n += len(us)
    return X, float(diff) / float(max(n, 1) * k)


# --------------------------------------------------------------------------- scoring / similarity
def recommend_full(X, Y, users, M=None):
    """SimpleMFRecAlgo.recommend (RecAlgos.py:20-32): scores X[u].Y[i] for i < M, full
    descending argsort and sort.  Ties keep numpy's argsort-then-reverse order."""
    M = Y.shape[0] if M is None else M
    rhats = np.asarray(X)[np.asarray(users)] @ np.asarray(Y)[:M].T
    return np.argsort(rhats)[:, ::-1], np.sort(rhats)[:, ::-1]


def cosine_similar(E, items, M=None):
    """SimpleMFSimAlgo.similar (RecAlgos.py:38-73): cosine of rows `items` against all
    rows, NaN (zero-norm rows) replaced by 0, truncated to the first M columns."""
    E = np.asarray(E)
    M = E.shape[0] if M is None else M
    with np.errstate(invalid="ignore", divide="ignore"):
        En = E / np.linalg.norm(E, axis=-1, keepdims=True)
    En = np.where(np.isnan(En), 0, En)
    sims = (En[np.asarray(items)] @ En.T)[:, :M]
    return np.argsort(sims)[:, ::-1], np.sort(sims)[:, ::-1]


# --------------------------------------------------------------------------- metrics
def rmse(x, y):
    """utils.py:58-59."""
    return np.sqrt(np.mean(np.power(x - y, 2)))


def compute_auc(rec, test_user, train_user):
    """eval_utils.py:5-21.  np.setdiff1d(..., assume_unique=True) keeps the order of what remains, so positions
    are positions in the ranked list with the training items taken out, exactly as in the reference."""
    rec_filt = np.setdiff1d(rec, train_user, assume_unique=True)
    n = len(rec_filt)
    pos = np.nonzero(np.isin(rec_filt, test_user))[0]
    fp = pos.astype(float) - np.arange(len(pos))
    p = len(pos)
    if p == 0: return -1
    f = n - p
    if f == 0: return 1
    return 1 - (np.cumsum(fp)[-1] / f) / p


def compute_ap(rec, test_user, train_user):
    """eval_utils.py:23-37 (positions as in compute_auc)."""
    if len(test_user) == 0: return -1
    rec_filt = np.setdiff1d(rec, train_user, assume_unique=True)
    pos = np.nonzero(np.isin(rec_filt, test_user))[0] + 1
    p = len(pos)
    if p == 0: return 0.
    return np.sum(np.arange(1, p + 1) / pos.astype(float)) / p


def map_at_k(recs, test_csr, train_csr, users, med_score=3.0, lim=200):
    """AUCEval.evaluate (EvalProto.py:28-62): mean over users of compute_ap on the first
    `lim` recommendations, relevant = test items rated above med_score, train items filtered;
    users whose compute_ap is negative (no relevant test item) are skipped."""
    tptr, tind, tdat = test_csr
    rptr, rind, _ = train_csr
    vals = []
    for k, u in enumerate(users):
        ti = np.asarray(tind[tptr[u]:tptr[u + 1]]); tr = np.asarray(tdat[tptr[u]:tptr[u + 1]])
        rel = ti[tr > med_score]
        trn = np.asarray(rind[rptr[u]:rptr[u + 1]])
        ap = compute_ap(np.asarray(recs[k][:lim]), rel, trn)
        if ap < 0: continue
        vals.append(ap)
    return float(np.mean(vals)) if vals else float("nan")
