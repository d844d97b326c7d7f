"""numpy emulation of the wave-level blocked solve (MFMA C layout), to validate index logic."""
import numpy as np

LANES = np.arange(64)
G = LANES >> 4
C = LANES & 15


def fidx(j, t, VW):
    if VW <= 4:
        return VW * j + t
    return np.where(t < 4, 4 * j + t, 64 + 4 * j + (t - 4))


def tile_index(ta, tb, VW):
    return ta * VW - ta * (ta - 1) // 2 + (tb - ta)


def mfma(a, b, c):
    """v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] B[k][j]; A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
    C/D: lane l reg r = (row 4*(l>>4)+r, col l&15).  a, b: [64]; c: [64, 4]."""
    A = np.zeros((16, 4), np.float32); B = np.zeros((4, 16), np.float32)
    for l in range(64):
        A[l & 15, l >> 4] = a[l]; B[l >> 4, l & 15] = b[l]
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def shfl(x, src):
    return x[src]


def row_bcast(x, k):          # DPP row_newbcast:k
    return x[(LANES & 48) | k]


def row_sum16(x):             # butterfly within 16-lane rows
    x = x + x[LANES ^ 1]
    x = x + x[LANES ^ 2]
    x = x + x[(LANES & ~7) | (7 - (LANES & 7))]
    x = x + x[(LANES & 48) | (15 - (LANES & 15))]
    return x


def grp_sum(x):               # butterfly across the 4 lane groups
    x = x + x[LANES ^ 16]
    x = x + x[LANES ^ 32]
    return x


def tile_from_dense(Mx):      # 16x16 -> [64,4] C layout
    t = np.zeros((64, 4), np.float32)
    for l in range(64):
        for r in range(4):
            t[l, r] = Mx[4 * (l >> 4) + r, l & 15]
    return t


def dense_from_tile(t):
    Mx = np.zeros((16, 16), np.float32)
    for l in range(64):
        for r in range(4):
            Mx[4 * (l >> 4) + r, l & 15] = t[l, r]
    return Mx


def xtz(X, Z, out=None, neg=False):   # out (+)= X^T Z, 4 MFMAs
    if out is None:
        out = np.zeros((64, 4), np.float32)
    for r in range(4):
        out = mfma(-X[:, r] if neg else X[:, r], Z[:, r], out)
    return out


def col_to_row(vcol):         # vrow[:, r] = v[4g + r]
    return np.stack([shfl(vcol, 16 * G + 4 * G + r) for r in range(4)], axis=1)


def row_to_col(vrow):         # vcol(c) = vrow[c & 3] at group c >> 2
    tmp = vrow[LANES, C & 3]
    return shfl(tmp, 16 * (C >> 2) + C)


def xt_v(X, vrow):            # out[c] = sum_k X[k][c] v[k]   (col-indexed, replicated over groups)
    p = np.zeros(64, np.float32)
    for r in range(4):
        p = p + X[:, r] * vrow[:, r]
    return grp_sum(p)


def x_v(X, vcol):             # out[k=4g+r] = sum_c X[k][c] v[c]   (row-indexed [64,4], replicated over c)
    return np.stack([row_sum16(X[:, r] * vcol) for r in range(4)], axis=1)


def diag_block(D):
    """Forward elimination of the 16x16 tile D (C layout).  Returns X = U_d^{-1} tile and pivots ok."""
    D = D.copy()
    N = tile_from_dense(np.eye(16, dtype=np.float32))
    for k in range(16):
        reg, src = k & 3, 16 * (k >> 2)
        d = D[src + k, reg]                           # readlane
        rd = np.float32(1.0) / d
        rowk = shfl(D[:, reg], src + C)               # row k of D, broadcast over groups
        v = np.where(C > k, rowk * rd, np.float32(0)).astype(np.float32)
        for r in range(4):
            D[:, r] = D[:, r] - row_bcast(D[:, r], k) * v
            N[:, r] = N[:, r] - row_bcast(N[:, r], k) * v
    # diagonal d_c still sits at lane (c>>2, c) reg c&3
    tmp = D[LANES, C & 3]
    dvec = shfl(tmp, 16 * (C >> 2) + C)
    ok = np.all(dvec > 0)
    s = (1.0 / np.sqrt(dvec)).astype(np.float32)
    X = N * s[:, None]
    return X, ok


def blocked_solve(T, bcol, VW):
    """T: dict t -> [64,4] tiles (ta<=tb) of the SPD matrix in permuted block order; bcol: [VW][64]
    col-indexed rhs per block.  Returns xcol [VW][64]."""
    T = {k: v.copy() for k, v in T.items()}
    b = [x.copy() for x in bcol]
    Xd, zrow = [None] * VW, [None] * VW
    for kb in range(VW):
        X, ok = diag_block(T[tile_index(kb, kb, VW)])
        assert ok
        Xd[kb] = X
        for jb in range(kb + 1, VW):
            T[tile_index(kb, jb, VW)] = xtz(X, T[tile_index(kb, jb, VW)])
        z = xt_v(X, col_to_row(b[kb]))                # z_kb = U_d^{-T} b'_kb   (col-indexed)
        zrow[kb] = col_to_row(z)
        for ib in range(kb + 1, VW):
            Ui = T[tile_index(kb, ib, VW)]
            for jb in range(ib, VW):
                t = tile_index(ib, jb, VW)
                T[t] = xtz(Ui, T[tile_index(kb, jb, VW)], out=T[t], neg=True)
            b[ib] = b[ib] - xt_v(Ui, zrow[kb])
    x = [None] * VW
    for kb in range(VW - 1, -1, -1):
        acc = zrow[kb].copy()
        for jb in range(kb + 1, VW):
            acc = acc - x_v(T[tile_index(kb, jb, VW)], x[jb])
        x[kb] = row_to_col(x_v(Xd[kb], row_to_col(acc)))
    return x


def run(VW, f, seed=0):
    rng = np.random.default_rng(seed)
    FP = 16 * VW
    Yf = rng.normal(size=(3 * f, f))
    A = Yf.T @ Yf + 0.5 * np.eye(f)
    bb = rng.normal(size=f)
    Ap = np.eye(FP); Ap[:f, :f] = A
    bp = np.zeros(FP); bp[:f] = bb
    # build tiles: block (ta,tb) element (i,j) = Ap[fidx(i,ta), fidx(j,tb)]
    idx = [np.array([int(fidx(i, t, VW)) for i in range(16)]) for t in range(VW)]
    T = {}
    for ta in range(VW):
        for tb in range(ta, VW):
            T[tile_index(ta, tb, VW)] = tile_from_dense(Ap[np.ix_(idx[ta], idx[tb])].astype(np.float32))
    bcol = [bp[idx[t]][C].astype(np.float32) for t in range(VW)]
    x = blocked_solve(T, bcol, VW)
    xs = np.zeros(FP)
    for t in range(VW):
        xs[idx[t]] = x[t][:16]
        assert np.allclose(x[t][:16], x[t][16:32]) and np.allclose(x[t][:16], x[t][48:])
    ref = np.linalg.solve(Ap, bp)
    err = np.abs(xs - ref).max() / np.abs(ref).max()
    print("VW", VW, "f", f, "rel err", err)
    return err


if __name__ == "__main__":
    for VW, f in [(1, 5), (1, 16), (2, 20), (2, 32), (4, 50), (4, 64), (8, 100), (8, 128)]:
        assert run(VW, f) < 2e-4
