"""Randomised shake-out of the whole training loop (ALSEngine: Gram, X sweep, Gram, Y sweep, device-side
transpose, convergence trace) against the numpy restatement of ALSRef.train, 3 iterations per case.
usage (on a GPU box): python tools/fuzz_engine.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as sps
import recommender_amd as ra
from recommender_amd.als import ALSEngine
from oracle import als_ref
rng = np.random.default_rng(7)
worst = 0.0
for case in range(25):
    f = int(rng.choice([4, 8, 20, 32, 50, 64, 96, 128]))
    N = int(rng.integers(3 * f, 3 * f + 1200)); M = int(rng.integers(3 * f, 3 * f + 900))
    dens = float(rng.choice([0.01, 0.03, 0.1]))
    U = sps.random(N, M, density=dens, random_state=int(rng.integers(1 << 30)), format="csr", dtype=np.float64)
    U.data = rng.integers(1, 11, size=U.nnz) * 0.5
    if rng.random() < 0.5:   # one very popular item and one very active user
        U = U.tolil(); U[:, 0] = 3.0; U[0, :] = 2.5; U = U.tocsr()
    U.sort_indices()
    lamb = float(rng.choice([0.01, 0.1])); alpha = float(rng.choice([1.0, 40.0]))
    chunk = int(rng.choice([64, 256, 2048]))
    X0 = rng.normal(0, 1 / np.sqrt(f), size=(N, f)).astype(np.float32)
    Y0 = rng.normal(0, 1 / np.sqrt(f), size=(M, f)).astype(np.float32)
    cu = ra.DeviceCSR.from_scipy(U, "cuda", chunk_nnz=chunk)
    ci = cu.transpose()
    X = torch.from_numpy(X0).cuda(); Y = torch.from_numpy(Y0).cuda()
    eng = ALSEngine(cu, ci, X, Y, lamb, alpha)
    eng.fit(3)
    Xr, Yr, trace, _ = als_ref.train_exact(U.indptr, U.indices, U.data, (N, M), f, lamb, alpha, 3,
                                           X0.astype(np.float64), Y0.astype(np.float64))
    ex = np.abs(X.cpu().numpy() - Xr).max() / np.abs(Xr).max(); ey = np.abs(Y.cpu().numpy() - Yr).max() / np.abs(Yr).max()
    et = np.abs(np.array(eng.trace) - trace).max() / trace.max()
    worst = max(worst, ex, ey)
    flag = "" if max(ex, ey) < 1e-4 and et < 1e-4 else "   <-- FAIL"
    print(case, "f", f, "N", N, "M", M, "nnz", U.nnz, "chunk", chunk, "lamb", lamb, "alpha", alpha,
          "X %.1e Y %.1e trace %.1e" % (ex, ey, et), flag, flush=True)
print("worst factor error %.2e" % worst)
