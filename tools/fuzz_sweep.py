"""Randomised shake-out of both half-sweeps (exact and 3 CG steps) against the CPU oracles: random factor counts
(aligned and not), leading dimensions, chunk sizes, row lengths (empty rows, rows of several chunks), stored zeros.
usage (on a GPU box): python tools/fuzz_sweep.py   - prints one line per case and the worst ratios."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import recommender_amd as ra
from oracle import c_oracle, als_ref
rng = np.random.default_rng(2026)
worst = 0.0; worst_cg = 0.0
for case in range(60):
    f = int(rng.choice([4, 5, 8, 12, 16, 20, 31, 32, 33, 48, 64, 65, 80, 96, 100, 127, 128]))
    N = int(rng.integers(1, 400)); M = int(rng.integers(max(f, 2), 3000))
    chunk = int(rng.choice([64, 128, 256, 2048]))
    lens = np.minimum(rng.zipf(1.4, size=N) + rng.integers(0, 40, size=N), M).astype(np.int64)
    lens[rng.integers(0, N, size=max(N // 10, 1))] = 0
    if rng.random() < 0.5: lens[rng.integers(0, N)] = min(M, int(rng.integers(chunk, 6 * chunk + 1)))
    ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    ind = np.concatenate([np.sort(rng.choice(M, size=l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    dat = (rng.integers(1, 11, size=ptr[-1]) * 0.5).astype(np.float32)
    if dat.size and rng.random() < 0.3: dat[rng.integers(0, dat.size, size=max(dat.size // 20, 1))] = 0.0   # stored zeros
    lamb = float(rng.choice([0.01, 0.1, 1.0])); alpha = float(rng.choice([1.0, 10.0, 40.0]))
    ld = f + int(rng.choice([0, 0, 4, 3]))
    Yh = np.zeros((M, ld), np.float32); Yh[:, :f] = rng.normal(0, 1 / np.sqrt(f), size=(M, f))
    Xh = np.zeros((N, ld), np.float32); Xh[:, :f] = rng.normal(0, 1 / np.sqrt(f), size=(N, f))
    Y = torch.from_numpy(Yh).cuda()[:, :f]; X = torch.from_numpy(Xh).cuda()[:, :f]
    csr = ra.DeviceCSR(ptr, ind, dat, (N, M), "cuda", chunk_nnz=chunk)
    gc = ra.gram(Y, lamb)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    Xc = X.clone() if ld == f else torch.from_numpy(Xh).cuda()[:, :f]
    ra.half_sweep(csr, Y, gc, Xc, alpha, info=info)
    ref = c_oracle.half_sweep(ptr, ind, dat, Yh[:, :f], Xh[:, :f], lamb, alpha, dtype=np.float64, nthreads=4)
    # ill-conditioned systems (M < f-ish, tiny lamb) are out of the float32 bar: compare against the float32 oracle's own error
    ref32 = c_oracle.half_sweep(ptr, ind, dat, Yh[:, :f], Xh[:, :f], lamb, alpha, dtype=np.float32, nthreads=4).astype(np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(Xc.cpu().numpy().astype(np.float64) - ref).max() / scale
    err32 = np.abs(ref32 - ref).max() / scale
    bar = max(1e-4, 3 * err32)
    flag = "" if (err < bar and int(info.item()) == 0) else "  <-- FAIL"
    worst = max(worst, err / bar)
    if 64 < f <= 128 and f % 4 == 0 and ld % 4 == 0:       # the persistent stream kernel: the one-wave kernel's bits
        for cg in (0, 3):
            got = []
            for rk in ("wave", "stream"):
                Xs = torch.from_numpy(Xh).cuda()[:, :f]
                o = ra.ops.sweep_opts(row_kernel=rk)
                if cg: ra.ops.half_sweep_cg(csr, Y, gc, Xs, alpha, cg, form="matrix", opts=o)
                else: ra.half_sweep(csr, Y, gc, Xs, alpha, opts=o)
                got.append(Xs.cpu().numpy())
            if not np.array_equal(got[0], got[1]): flag += "  <-- STREAM != WAVE (cg=%d)" % cg
    # every accumulate mode, and the short rows in their entry-space form (n_opp >= 16 f: the Gram that is factored is
    # far from rank deficiency, the condition under which the engine turns these on)
    for mode in ("bf16x3", "fp16x2", "fp32"):
        for short in (False, True):
            if short and not (ra.short_rows_max_entries(f) > 0 and ld % 4 == 0 and M >= 16 * f):
                continue
            if mode == "fp16x2" and M < 16 * f:
                continue
            Xm = torch.from_numpy(Xh).cuda()[:, :f]
            info.zero_()
            if short: ra.whiten(Y, gc, info=info)
            ra.half_sweep(csr, Y, gc, Xm, alpha, info=info, opts=ra.ops.sweep_opts(accumulate=mode), short_rows=short)
            e = np.abs(Xm.cpu().numpy().astype(np.float64) - ref).max() / scale
            worst = max(worst, e / bar)
            if not (e < bar and int(info.item()) == 0): flag += "  <-- FAIL (%s%s: %.2e)" % (mode, ", short rows" if short else "", e)
    Xg = torch.from_numpy(Xh).cuda()[:, :f]
    ra.half_sweep_cg(csr, Y, gc, Xg, alpha, 3)
    refcg = als_ref.half_sweep_cg(ptr, ind, dat.astype(np.float64), Yh[:, :f].astype(np.float64), Xh[:, :f].astype(np.float64), lamb, alpha, 3)
    errcg = np.abs(Xg.cpu().numpy().astype(np.float64) - refcg).max() / max(np.abs(refcg).max(), 1e-30)
    flagcg = "" if errcg < 5e-4 else "  <-- CG FAIL"
    worst_cg = max(worst_cg, errcg)
    print(case, "f", f, "N", N, "M", M, "chunk", chunk, "ld", ld, "nnz", int(ptr[-1]), "err %.2e (bar %.1e) cg %.2e" % (err, bar, errcg), flag, flagcg, flush=True)
print("worst exact err/bar %.3f  worst cg %.2e" % (worst, worst_cg))
