#!/usr/bin/env python3
"""Mint the golden vectors under tests/golden/ from the reference's own numpy code.

Runs ONLY in the development container, where /root/reference is mounted.  It
reads the reference's source as text and executes the pure-numpy pieces of it
(the TensorFlow imports at the top of those modules are not available, so the
modules cannot be imported whole):

  * class ALSRef                     reclibwh/core/ALS.py:35-158
  * get_pos_ratings_padded, rmse     reclibwh/utils/utils.py:58-59,142-173
  * compute_auc, compute_ap          reclibwh/utils/eval_utils.py:5-37

Nothing from the reference is written into the repository: the outputs are
input/expected-output arrays only (.npz, loadable with allow_pickle=False).

Usage:  python tests/golden/make_golden.py  [--ref /root/reference]
"""
import argparse
import os
import re
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))


def _restore_numpy_aliases():
    # aliases removed in numpy >= 1.24 that the reference still uses (ALS.py:126, utils.py:163-165)
    for name, typ in (("bool", bool), ("float", float), ("int", int)):
        if not hasattr(np, name):
            setattr(np, name, typ)


def _slice_source(path, start_pat, end_pat):
    """Return the text of `path` from the first line matching start_pat up to (not
    including) the first later line matching end_pat."""
    lines = open(path).read().split("\n")
    s = next(i for i, l in enumerate(lines) if re.match(start_pat, l))
    e = next((i for i in range(s + 1, len(lines)) if re.match(end_pat, lines[i])), len(lines))
    return "\n".join(lines[s:e])


def load_reference(ref_root):
    _restore_numpy_aliases()

    class _NoTqdm:
        def __call__(self, it, *a, **k):
            return it

    ns = dict(np=np, sps=sps, pd=pd, os=os, tqdm=_NoTqdm())
    exec(_slice_source(os.path.join(ref_root, "reclibwh/core/ALS.py"),
                       r"^class ALSRef", r"^@tf\.function"), ns)
    uns = dict(np=np, sps=sps, pd=pd)
    upath = os.path.join(ref_root, "reclibwh/utils/utils.py")
    exec(_slice_source(upath, r"^def rmse", r"^def mean_nnz"), uns)
    exec(_slice_source(upath, r"^def get_pos_ratings_padded", r"(?!)"), uns)
    ens = dict(np=np, pd=pd)
    exec(_slice_source(os.path.join(ref_root, "reclibwh/utils/eval_utils.py"),
                       r"^def compute_auc", r"^def filter_train_rec"), ens)
    return ns["ALSRef"], uns["get_pos_ratings_padded"], uns["rmse"], ens["compute_auc"], ens["compute_ap"]


# ----------------------------------------------------------------------------- inputs
def dummy_block_matrix(m, n, nblocks, rng, split=0.8):
    """Same construction as ExplicitDataDummy (ItemMetadata.py:419-459): block-diagonal
    ratings, block b filled with b%5+1, random permutation, 80/20 split - but seeded."""
    rows, cols, data = [], [], []
    for b in range(nblocks):
        bc = np.tile(np.arange(m * b, m * (b + 1)), (n, 1)).transpose().flatten()
        br = np.tile(np.arange(n * b, n * (b + 1)), (m, 1)).flatten()
        rows.append(br); cols.append(bc); data.append(np.ones(m * n) * (b % 5 + 1))
    rows = np.concatenate(rows).astype(int); cols = np.concatenate(cols).astype(int)
    data = np.concatenate(data)
    perm = rng.permutation(len(data))
    rows, cols, data = rows[perm], cols[perm], data[perm]
    k = int(split * len(data))
    N, M = n * nblocks, m * nblocks
    tr = sps.csr_matrix((data[:k], (rows[:k], cols[:k])), shape=(N, M))
    te = sps.csr_matrix((data[k:], (rows[k:], cols[k:])), shape=(N, M))
    return tr, te


def powerlaw_matrix(N, M, nnz_target, rng):
    """Small MovieLens-like implicit matrix: Zipf item popularity, skewed user activity,
    ratings uniform over {0.5, 1.0, ..., 5.0}; canonical CSR (sorted, duplicate free)."""
    item_p = 1.0 / np.arange(1, M + 1) ** 0.9
    item_p /= item_p.sum()
    user_p = 1.0 / np.arange(1, N + 1) ** 0.6
    user_p /= user_p.sum()
    u = rng.choice(N, size=nnz_target, p=user_p)
    i = rng.choice(M, size=nnz_target, p=item_p)
    key = np.unique(u.astype(np.int64) * M + i)
    u, i = key // M, key % M
    r = rng.integers(1, 11, size=len(key)) * 0.5
    U = sps.csr_matrix((r, (u, i)), shape=(N, M))
    U.sort_indices()
    return U


def csr_parts(U, prefix):
    U = sps.csr_matrix(U)
    return {prefix + "_indptr": U.indptr.astype(np.int64), prefix + "_indices": U.indices.astype(np.int32),
            prefix + "_data": U.data.astype(np.float64), prefix + "_shape": np.array(U.shape, dtype=np.int64)}


def half_sweep(ALSRef, K, lamb, alpha, Yfix, Xold, U):
    """One ALSRef._run_single_step on CSR inputs (C, p built as ALSRef.train does, ALS.py:126-128)."""
    U = sps.csr_matrix(U)
    p = U.astype(bool).astype(float)
    C = U.copy(); C.data = 1 + alpha * C.data
    ref = ALSRef(N=Xold.shape[0], M=Yfix.shape[0], K=K, lamb=lamb, alpha=alpha, Xinit=Xold, Yinit=Yfix)
    return ref._run_single_step(Yfix, Xold, C, U, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    ALSRef, get_pos_ratings_padded, rmse, compute_auc, compute_ap = load_reference(args.ref)

    # sanity: the reference's fast and naive steps agree (ALS.py:67-110)
    rng = np.random.default_rng(7)
    U = sps.random(30, 20, density=0.3, random_state=np.random.RandomState(3), format="csr")
    U.data = np.ceil(U.data * 5)
    a = ALSRef(N=30, M=20, K=5, lamb=1e-2, alpha=1.0)
    p = U.astype(bool).astype(float); C = U.copy(); C.data = 1 + 1.0 * C.data
    d = np.abs(a._run_single_step(a.Y, a.X, C, U, p) - a._run_single_step_naive(a.Y, a.X, U, p)).max()
    assert d < 1e-12, d
    print("ALSRef fast vs naive max-abs:", d)

    # ---- G1: replica of tests/core/test_ALS.py:18-53 (ExplicitDataDummy(3,4,1), K=5, alpha=1, lamb=1e-2)
    rng = np.random.default_rng(101)
    Utr, _ = dummy_block_matrix(3, 4, 1, rng)
    N, M = Utr.shape
    K, alpha, lamb = 5, 1.0, 1e-2
    Xinit = rng.normal(0, 1 / np.sqrt(K), size=(N, K))
    Yinit = rng.normal(0, 1 / np.sqrt(K), size=(M, K))
    Xnew = half_sweep(ALSRef, K, lamb, alpha, Yinit, Xinit, Utr)
    out = dict(K=np.int64(K), alpha=np.float64(alpha), lamb=np.float64(lamb), Xinit=Xinit, Yinit=Yinit, Xnew=Xnew)
    out.update(csr_parts(Utr, "U"))
    np.savez_compressed(os.path.join(HERE, "g1_single_step.npz"), **out)

    # ---- G2: power-law 200 x 150, K in {8, 20, 32, 64}, alpha=40, lamb=0.01, 3 full steps via ALSRef.train
    rng = np.random.default_rng(202)
    U = powerlaw_matrix(200, 150, 3600, rng)
    N, M = U.shape
    out = dict(alpha=np.float64(40.0), lamb=np.float64(0.01), steps=np.int64(3))
    out.update(csr_parts(U, "U"))
    for K in (8, 20, 32, 64):
        for conv in ("ref", "tf"):
            if conv == "ref":   # ALSRef.__init__: re-seed 42 before X and before Y (ALS.py:59-65)
                als = ALSRef(N=N, M=M, K=K, lamb=0.01, alpha=40.0, steps=1)
            else:               # als_initizalize: seed 42 once, X then Y from one stream (ALS.py:255-257)
                np.random.seed(42)
                Xi = np.random.normal(0, 1 / np.sqrt(K), size=(N, K))
                Yi = np.random.normal(0, 1 / np.sqrt(K), size=(M, K))
                als = ALSRef(N=N, M=M, K=K, lamb=0.01, alpha=40.0, steps=1, Xinit=Xi, Yinit=Yi)
            tag = "K%d_%s" % (K, conv)
            out[tag + "_X0"] = als.X.copy(); out[tag + "_Y0"] = als.Y.copy()
            trace = []
            for s in range(3):
                t = als.train(U)          # steps=1: one X half-sweep then one Y half-sweep
                trace.append(t[0])
                out["%s_X%d" % (tag, s + 1)] = als.X.copy()
                out["%s_Y%d" % (tag, s + 1)] = als.Y.copy()
            out[tag + "_trace"] = np.array(trace)
        # 3 single-step trains == one 3-step train (same arithmetic, checked once)
    chk = ALSRef(N=N, M=M, K=8, lamb=0.01, alpha=40.0, steps=3)
    tr3 = chk.train(U)
    assert np.array_equal(chk.X, out["K8_ref_X3"]) and np.allclose(tr3, out["K8_ref_trace"], rtol=0, atol=0)
    np.savez_compressed(os.path.join(HERE, "g2_powerlaw_train.npz"), **out)

    # ---- G3: edge cases - empty rows, single-nnz rows, one long row, an empty column
    rng = np.random.default_rng(303)
    N, M, K = 12, 400, 16
    rows, cols, vals = [], [], []
    # row 0 empty, row 1 one entry, row 2 one entry (last column), row 3 long (all but column 7), rows 4.. random
    rows += [1]; cols += [5]; vals += [3.5]
    rows += [2]; cols += [M - 1]; vals += [0.5]
    lc = np.setdiff1d(np.arange(M), [7])
    rows += [3] * len(lc); cols += list(lc); vals += list(rng.integers(1, 11, size=len(lc)) * 0.5)
    for r in range(4, N - 1):
        c = rng.choice(np.setdiff1d(np.arange(M), [7]), size=int(rng.integers(2, 70)), replace=False)
        rows += [r] * len(c); cols += list(np.sort(c)); vals += list(rng.integers(1, 11, size=len(c)) * 0.5)
    # row N-1 empty as well
    U = sps.csr_matrix((np.array(vals, float), (np.array(rows), np.array(cols))), shape=(N, M))
    U.sort_indices()
    X0 = rng.normal(0, 1 / np.sqrt(K), size=(N, K)); Y0 = rng.normal(0, 1 / np.sqrt(K), size=(M, K))
    Xn = half_sweep(ALSRef, K, 0.01, 40.0, Y0, X0, U)
    Yn = half_sweep(ALSRef, K, 0.01, 40.0, Xn, Y0, sps.csr_matrix(U.T))
    out = dict(K=np.int64(K), alpha=np.float64(40.0), lamb=np.float64(0.01), X0=X0, Y0=Y0, Xnew=Xn, Ynew=Yn)
    out.update(csr_parts(U, "U"))
    np.savez_compressed(os.path.join(HERE, "g3_edge_cases.npz"), **out)

    # ---- G4: metrics - compute_ap / compute_auc / rmse input -> output pairs
    rng = np.random.default_rng(404)
    recs, tests, trains, aps, aucs = [], [], [], [], []
    M = 60
    for case in range(40):
        rec = rng.permutation(M)[: int(rng.integers(5, 40))]
        test = rng.choice(M, size=int(rng.integers(0, 8)), replace=False)
        train = rng.choice(np.setdiff1d(np.arange(M), test), size=int(rng.integers(0, 10)), replace=False)
        if case == 0: test = np.array([], dtype=int)
        if case == 1: test = np.setdiff1d(np.arange(M), rec)[:3]      # no hit
        if case == 2: train = np.array([], dtype=int)
        if case == 3: test = rec.copy(); train = np.array([], dtype=int)  # everything relevant
        recs.append(rec); tests.append(test); trains.append(train)
        aps.append(compute_ap(rec, test, train)); aucs.append(compute_auc(rec, test, train))

    def pad(lst):
        L = max(len(x) for x in lst)
        outp = -np.ones((len(lst), L), dtype=np.int64)
        for k, x in enumerate(lst): outp[k, : len(x)] = x
        return outp
    x = rng.normal(size=50); y = rng.normal(size=50)
    np.savez_compressed(os.path.join(HERE, "g4_metrics.npz"), rec=pad(recs), test=pad(tests), train=pad(trains),
                        ap=np.array(aps, float), auc=np.array(aucs, float), rmse_x=x, rmse_y=y,
                        rmse=np.float64(rmse(x, y)))

    # ---- G5: fold-in - two users with identical ratings get identical factor rows (test_ALS.py:88-101)
    rng = np.random.default_rng(505)
    M, K = 40, 20
    c = np.sort(rng.choice(M, size=9, replace=False)); v = rng.integers(1, 11, size=9) * 0.5
    U = sps.csr_matrix((np.concatenate([v, v]), (np.repeat([0, 1], 9), np.concatenate([c, c]))), shape=(2, M))
    Y0 = rng.normal(0, 1 / np.sqrt(K), size=(M, K)); X0 = rng.normal(0, 1 / np.sqrt(K), size=(2, K))
    Xn = half_sweep(ALSRef, K, 1e-6, 40.0, Y0, X0, U)
    assert np.array_equal(Xn[0], Xn[1])
    out = dict(K=np.int64(K), alpha=np.float64(40.0), lamb=np.float64(1e-6), X0=X0, Y0=Y0, Xnew=Xn)
    out.update(csr_parts(U, "U"))
    np.savez_compressed(os.path.join(HERE, "g5_foldin.npz"), **out)

    # ---- G6: padded batch builder semantics (utils.py:142-173 as used by PresetIterators.py:11-17)
    rng = np.random.default_rng(606)
    U = powerlaw_matrix(17, 23, 120, rng)
    out = csr_parts(U, "U")
    users = np.array([0, 3, 4, 9, 16])
    rp, yp = get_pos_ratings_padded(U, users, U.shape[1])
    out.update(users=users, rp=rp.astype(np.float64), yp=yp.astype(np.int64), pad_val=np.int64(U.shape[1]))
    np.savez_compressed(os.path.join(HERE, "g6_padded_batch.npz"), **out)
    # ---- G7: the saved-run folder of a finished reference ALS run as the reference's repository holds it
    # (deploy/ALS_2020-12-16.15-48-46-serve: algorithm_config.json written by ALS.py:283, eval.csv by
    # EvalProto.py:74-94): two small data files, copied as they are
    import shutil
    src = os.path.join(args.ref, "deploy", "ALS_2020-12-16.15-48-46-serve")
    dst = os.path.join(HERE, "ref_deploy_als")
    os.makedirs(dst, exist_ok=True)
    for name in ("algorithm_config.json", "eval.csv"):
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"): print("  %-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    sys.exit(main())
