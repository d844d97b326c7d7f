// Short rows of the exact half-sweep in their "entry space" form, and the whitened factor table it reads.
//
// A row u with L stored entries (items i_1..i_L, ratings r_1..r_L) solves, in the reference,
//     (G + sum_e w_e y_e y_e^T) x = sum_e c_e y_e,   G = Y^T Y + lamb I,  w_e = alpha r_e,  c_e = 1 + w_e
// (reclibwh/core/ALS.py:82-86): an f x f system whose cost does not depend on L.  With the Cholesky factor G = R^T R
// (R upper triangular) and the whitened rows  v_e = sqrt(w_e) R^{-T} y_e  the same x is
//     x = G^{-1} sum_e sqrt(w_e) t_e y_e,     (I + V V^T) t = c / sqrt(w),     V = [v_1 .. v_L]^T
// (push-through identity: (I + V^T V)^{-1} V^T = V^T (I + V V^T)^{-1}).  That is an L x L system - symmetric, positive
// definite, every eigenvalue >= 1 - and for L < f it is the smaller one: the 16 x 16 tile count of the matrix and the
// length of the blocked Cholesky fall from f/16 to ceil(L/16) block rows.  Per half-sweep the fixed factor is whitened
// once: WZ[i] = [ y_i R^{-1} | y_i G^{-1} ]  (one GEMM against a 2f x f table), and a row then
//   pass 1: gathers the W halves of its entries, scales them by sqrt(w_e), cuts them exactly into three bf16 pieces
//           and forms V V^T on the bf16 matrix cores (contraction over the factor index, 32 per MFMA) - the rows of
//           one 16-entry group are at once the A and the B operand fragment, nothing moves between lanes;
//   solve : adds I and runs the blocked Cholesky of als_tiles.inc on the ceil(L/16)^2 / 2 accumulator tiles in place;
//   pass 2: gathers the Z halves and sums  sqrt(w_e) t_e z_e  over the entries: that is x.
// The result depends on the row and on the table only, so it is the same bits for any launch shape or sharding.
// Numerically the form is better conditioned than the f x f one (its matrix is I plus a positive semi-definite part).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <utility>

#include "als_hip.h"

int als_fail_(int code, const char *fmt, ...);
int32_t als_short_rows_launch_(const int32_t *col, const float *val, const int64_t *row_ids, const als_item_t *items,
                               const int64_t *bounds, int32_t nb_max, const float *WZ, int64_t ldw, int32_t f, float alpha,
                               float rmax_f16, float *Xown, int64_t ldx, float *rowdiff, int32_t *info, hipStream_t st);

#define ALS_HIP_CHECK3(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return als_fail_(ALS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

#include "als_tiles.inc"

struct ShortParams {
    const int32_t *col;
    const float *val;
    const int64_t *row_ids;
    const als_item_t *items;
    const float *WZ;      // [n_opp, ldw]: W = Y R^-1 in columns [0, f), Z = Y G^-1 in columns [f, 2 f)
    float *X;
    float *rowdiff;
    int32_t *info;
    int64_t ldw, ldx;
    int32_t f;
    float alpha;
    float rmax;           // fp16 x 2 form: an upper bound of the ratings
};

// 8 consecutive factor indices of one entry's row, scaled by s, as the lane's operand fragment of each bf16 piece
// (register p holds k-slots 2 p and 2 p + 1)
__device__ __forceinline__ void split8_scaled(const f32x4 &a, const f32x4 &b, const float s, unsigned (&H)[4],
                                              unsigned (&M)[4], unsigned (&L)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = s * (p < 2 ? a[2 * p] : b[2 * p - 4]), x1 = s * (p < 2 ? a[2 * p + 1] : b[2 * p - 3]);
        const float r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
        const float l0 = r0 - trunc_bf16(r0), l1 = r1 - trunc_bf16(r1);
        H[p] = pack_trunc(x0, x1);
        M[p] = pack_trunc(r0, r1);
        L[p] = pack_trunc(l0, l1);
    }
}

// The same in the fp16 x 2 form (als_kernels.hip): two fp16 pieces of s x.
__device__ __forceinline__ void split8_scaled_f16(const f32x4 &a, const f32x4 &b, const float s, unsigned (&H)[4],
                                                  unsigned (&L)[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = s * (p < 2 ? a[2 * p] : b[2 * p - 4]), x1 = s * (p < 2 ? a[2 * p + 1] : b[2 * p - 3]);
        const unsigned h = pack_f16_pinned(x0, x1);
        H[p] = h;
        L[p] = pack_f16(x0 - f16_lo(h), x1 - f16_hi(h));
    }
}

// block rows from which the solve's tile products run on the bf16 pipe (below: float32 MFMAs); 5 and 7 measured the
// same as 8 on the 1.25 M x 1 M block (A/B on one box, within +-0.5 %)
#ifndef SHORT_BF16_SOLVE_FROM
#define SHORT_BF16_SOLVE_FROM 8
#endif
// waves per SIMD the kernel's register budget is sized for
__host__ __device__ constexpr int short_waves(const int NB) { return NB <= 2 ? 4 : NB <= 4 ? 3 : NB <= 6 ? 2 : 1; }

// One wave per row.  Items [first_item, first_item + gridDim.x) of p.items are whole rows with at most 16 NB entries.
// KS = 32-wide slices of the factor index (f <= 32 KS); FULL: f == 32 KS.
// Lane (g, i) = (lane >> 4, lane & 15) holds, of entry 16 a + i (a < NB), the factor indices 32 ks + 8 g .. + 7.
// F16: the products V V^T in the fp16 x 2 form - the whitened rows have entries of magnitude at most 1
// (W^T W = I - lamb G^-1), so the operand bound is sqrt(alpha rmax) alone.
template <int NB, int KS, bool FULL, bool F16>
__global__ __launch_bounds__(64, short_waves(NB)) void short_rows_kernel(const ShortParams p, const int64_t first_item) {
    using G = Geo<NB>;
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const als_item_t it = p.items[first_item + blockIdx.x];
    const int row = it.row, n = it.count;
    const int64_t xr = p.row_ids ? p.row_ids[row] : (int64_t)row;
    float *xrow = p.X + xr * p.ldx;
    if (n <= 0) {      // no entries: b = 0, so x = 0 exactly
        float d = 0.f;
        for (int k = lane; k < p.f; k += 64) {
            d += fabsf(xrow[k]);
            xrow[k] = 0.f;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) p.rowdiff[row] = d;
        return;
    }
    const uint32_t ldu = (uint32_t)(p.ldw / 4);          // row pitch in 16-byte units
    const f32x4 *__restrict__ Wv = reinterpret_cast<const f32x4 *>(p.WZ);
#ifdef ALS_ABLATE_PASS2_FROM_W      // timing-only build (wrong values): pass 2 re-reads the rows pass 1 fetched
    const f32x4 *__restrict__ Zv = Wv;
#else
    const f32x4 *__restrict__ Zv = Wv + p.f / 4;
#endif
    // the lane's entries.  Stored zeros are dropped (ALS.py:79) and slots past the row's end are empty: both get
    // weight 0, an identity row in the matrix and t = 0; their loads go to the row's first entry's table row, so
    // nothing but rows the user touched is ever read.
    uint32_t off[NB];
    float sq[NB], rhs[NB];
    const OpScale opsc = F16 ? op_scale(__builtin_sqrtf(p.alpha * p.rmax)) : OpScale{1.f, 1.f};
    const uint32_t off_first = (uint32_t)p.col[it.begin] * ldu;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const int e = 16 * a + i;
        const bool in = e < n;
        const int64_t idx = it.begin + (in ? e : 0);
        const int cc = p.col[idx];
        const float rr = p.val[idx];
        const float w = p.alpha * rr;
        const bool keep = in && rr != 0.f;
        const float s = __builtin_sqrtf(w);
        sq[a] = keep ? s : 0.f;
        rhs[a] = keep ? (1.f + w) / s : 0.f;
        off[a] = in ? (uint32_t)cc * ldu : off_first;
    }
    // this lane's 32 bytes of a table row lie within f (not FULL: f < 32 KS, f % 4 == 0)
    auto ok4 = [&](const int ks, const int h) { return FULL || 32 * ks + 8 * g + 4 * h < p.f; };
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};

    // the row's current value, for the reported change: loaded now, used at the very end
    float xold[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 32 * ks + 8 * g + (i & 7);
        xold[ks] = (FULL || k < p.f) ? xrow[k] : 0.f;
    }

    // ---- pass 1: T = V V^T (tiles ta <= tb).  The loads of slice ks + 1 are issued in front of the arithmetic of
    // slice ks (two register sets), so that only the first slice's latency is exposed.
    f32x4 T[G::NT];
#pragma unroll
    for (int t = 0; t < G::NT; ++t) T[t] = zero4;
    f32x4 va[2][NB], vb[2][NB];
    auto load_w = [&](const int ks, f32x4 (&a4)[NB], f32x4 (&b4)[NB]) {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            const f32x4 *src = Wv + off[a] + 8 * ks + 2 * g;
            a4[a] = ok4(ks, 0) ? src[0] : zero4;
            b4[a] = ok4(ks, 1) ? src[1] : zero4;
        }
    };
    load_w(0, va[0], vb[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load_w(ks + 1, va[(ks + 1) & 1], vb[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        unsigned H[NB][4], M[NB][4], L[NB][4];
        if constexpr (F16) {
#pragma unroll
            for (int a = 0; a < NB; ++a) split8_scaled_f16(va[ks & 1][a], vb[ks & 1][a], sq[a] * opsc.s, H[a], L[a]);
            constexpr int FA[3] = {2, 0, 0}, FB[3] = {0, 2, 0};      // lo.hi, hi.lo, hi.hi
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int ta = 0; ta < NB; ++ta)
#pragma unroll
                    for (int tb = ta; tb < NB; ++tb) {
                        const int t = tile_index(ta, tb, NB);
                        T[t] = mfma_f16(FA[pr] == 0 ? H[ta] : L[ta], FB[pr] == 0 ? H[tb] : L[tb], T[t]);
                    }
        } else {
#pragma unroll
            for (int a = 0; a < NB; ++a) split8_scaled(va[ks & 1][a], vb[ks & 1][a], sq[a], H[a], M[a], L[a]);
            // per tile the six significant piece products, smallest first; the tiles inner, so that consecutive
            // MFMAs go to different accumulators
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // 0: H, 1: M, 2: L
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int ta = 0; ta < NB; ++ta)
#pragma unroll
                    for (int tb = ta; tb < NB; ++tb) {
                        const int t = tile_index(ta, tb, NB);
                        const unsigned (&A)[4] = PA[pr] == 0 ? H[ta] : PA[pr] == 1 ? M[ta] : L[ta];
                        const unsigned (&B)[4] = PB[pr] == 0 ? H[tb] : PB[pr] == 1 ? M[tb] : L[tb];
                        T[t] = mfma_bf16(A, B, T[t]);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);      // one slice's operands at a time
    }
    if constexpr (F16) {
#pragma unroll
        for (int t = 0; t < G::NT; ++t) T[t] *= opsc.inv2;
    }
    // ---- I + V V^T, solved in place
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[tile_index(a, a, NB)][r] += (4 * g + r == i) ? 1.f : 0.f;
    float t[NB];
    bool ok;
    if constexpr (NB >= SHORT_BF16_SOLVE_FROM) ok = blocked_solve_bf16<NB>(T, rhs, t, lane);
    else ok = blocked_solve<NB>(T, rhs, t, lane);

    // ---- pass 2: x = sum_e sqrt(w_e) t_e z_e.  The matrix is dead: its registers take the Z rows of ZB slices at a
    // time, all issued before the first is used.
    float ts[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) ts[a] = t[a] * sq[a];
    float d = 0.f;
    constexpr int ZB = NB <= 6 ? (KS < 2 ? KS : 2) : KS;
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += ZB) {
        f32x4 za[ZB][NB], zb[ZB][NB];
#pragma unroll
        for (int kk = 0; kk < ZB; ++kk)
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const f32x4 *src = Zv + off[a] + 8 * (k0 + kk) + 2 * g;
                za[kk][a] = ok4(k0 + kk, 0) ? src[0] : zero4;
                zb[kk][a] = ok4(k0 + kk, 1) ? src[1] : zero4;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < ZB; ++kk) {
            const int ks = k0 + kk;
            f32x4 s0 = zero4, s1 = zero4;
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s0[q] = fmaf(ts[a], za[kk][a][q], s0[q]);
                    s1[q] = fmaf(ts[a], zb[kk][a][q], s1[q]);
                }
            // sum over the 16 entries of the lane row (fixed butterfly); lane i < 8 keeps factor index 32 ks + 8 g + i
            float xv = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float r = row_sum16(q < 4 ? s0[q] : s1[q - 4]);
                xv = (i == q) ? r : xv;
            }
            const int k = 32 * ks + 8 * g + i;
            if (i < 8 && (FULL || k < p.f)) {
                d += fabsf(xold[ks] - xv);
                xrow[k] = xv;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
    if (!ok && lane == 0 && p.info) atomicAdd(p.info, 1);
    if (lane == 0) p.rowdiff[row] = d;
}

// ----------------------------------------------------------------------------- whitening table
// One workgroup of 1024 threads: G = gram + lamb I in float64, its Cholesky factor G = L L^T, X = L^-1 and
// G^-1 = X^T X; writes the float32 table B [2 f, f] that the whitening GEMM takes as its "item" rows:
//     B[j]     = row j of L^-1 = column j of R^-1 (R = L^T)   (so <y, B[j]>     = (y R^-1)[j]),
//     B[f + j] = row j of G^-1                                  (so <y, B[f + j]> = (y G^-1)[j]).
// Thread (ty, tx) of 32 x 32 keeps elements (ty + 32 a, tx + 32 b), a, b < 4, of the Schur complement and of X in registers.
// Step k of the right-looking factorisation needs column k of the complement and row k of X: their owners put them
// into LDS (two buffers, by the parity of k: one barrier per step) and every thread updates its 2 x 16 elements:
//     S[r][c] -= s_rk s_ck / s_kk,   X[k][:] = x_k / sqrt(s_kk),   X[r][:] -= s_rk x_k / s_kk   (r > k)
// - the forward substitution L X = I carried along with the factorisation.
// info (optional) is incremented when G is not positive definite in float64 (the table is then NaN).
constexpr int WH_MAX_F = 128;
constexpr int WH_LD = WH_MAX_F + 1;      // odd row pitch: column walks hit different LDS banks
constexpr int WH_T = 32;                 // threads per side (1024 threads; 16 x 16 threads with 8 x 8 elements each measured slower)
constexpr int WH_E = WH_MAX_F / WH_T;    // elements per thread and side
__global__ __launch_bounds__(WH_T * WH_T) void whiten_factor_kernel(const float *__restrict__ gram, const float lamb,
                                                                    const int f, float *__restrict__ B, int32_t *info) {
    __shared__ double Xs[WH_MAX_F * WH_LD];
    __shared__ double colk[2][WH_MAX_F], rowx[2][WH_MAX_F];
    const int tid = threadIdx.x, tx = tid % WH_T, ty = tid / WH_T;
    double S[WH_E][WH_E], X[WH_E][WH_E];
#pragma unroll
    for (int a = 0; a < WH_E; ++a)
#pragma unroll
        for (int b = 0; b < WH_E; ++b) {
            const int r = ty + WH_T * a, c = tx + WH_T * b;
            const bool in = r < f && c < f;
            S[a][b] = in ? (double)gram[r * f + c] + (r == c ? (double)lamb : 0.0) : (r == c ? 1.0 : 0.0);
            X[a][b] = r == c ? 1.0 : 0.0;
        }
    bool bad = false;
    for (int k = 0; k < f; ++k) {
        const int kq = k / WH_T, kr = k % WH_T, par = k & 1;
        // column k of the complement and row k of X, from their owners (kq is uniform: the branch picks the register)
#pragma unroll
        for (int q = 0; q < WH_E; ++q) {
            if (kq != q) continue;
            if (tx == kr) {
#pragma unroll
                for (int a = 0; a < WH_E; ++a) colk[par][ty + WH_T * a] = S[a][q];
            }
            if (ty == kr) {
#pragma unroll
                for (int b = 0; b < WH_E; ++b) rowx[par][tx + WH_T * b] = X[q][b];
            }
        }
        __syncthreads();
        const double d = colk[par][k];
        bad = bad || !(d > 0.0);
        // 1 / d and 1 / sqrt(d) from the hardware estimates and two Newton steps each (to float64 rounding): the
        // library division and square root are about a hundred instructions, per wave and step
        double inv = __builtin_amdgcn_rcp(d), isq = __builtin_amdgcn_rsq(d);
        inv = inv * fma(-d, inv, 2.0);
        inv = inv * fma(-d, inv, 2.0);
        isq = isq * fma(-0.5 * d * isq, isq, 1.5);
        isq = isq * fma(-0.5 * d * isq, isq, 1.5);
        // Only the live part is touched: rows > k, columns > k of the complement (a block column wholly at or left of
        // k is skipped) and columns <= k of X (row k of X is zero to the right of k).  The branches are uniform.
        double cc[WH_E], xc[WH_E];
#pragma unroll
        for (int b = 0; b < WH_E; ++b) {
            cc[b] = colk[par][tx + WH_T * b];
            xc[b] = rowx[par][tx + WH_T * b];
        }
#pragma unroll
        for (int a = 0; a < WH_E; ++a) {
            if (WH_T * a + WH_T - 1 < k) continue;               // every row of this block row is <= k - 1: retired
            const int r = ty + WH_T * a;
            const double cr = colk[par][r] * inv;
            if (WH_T * a > k) {                                   // every row of the block row is > k
#pragma unroll
                for (int b = 0; b < WH_E; ++b) {
                    if (WH_T * b + WH_T - 1 > k) S[a][b] -= cr * cc[b];
                    if (WH_T * b <= k) X[a][b] -= cr * xc[b];
                }
            } else {                                              // the block row that holds row k
#pragma unroll
                for (int b = 0; b < WH_E; ++b) {
                    if (r > k) {
                        S[a][b] -= cr * cc[b];
                        X[a][b] -= cr * xc[b];
                    } else if (r == k) {
                        X[a][b] = xc[b] * isq;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < WH_E; ++a)
#pragma unroll
        for (int b = 0; b < WH_E; ++b) Xs[(ty + WH_T * a) * WH_LD + tx + WH_T * b] = X[a][b];
    __syncthreads();
    // (G^-1)_rc = sum_m X_mr X_mc  (X lower triangular: the terms with m < max(r, c) are exact zeros); S is dead
#pragma unroll
    for (int a = 0; a < WH_E; ++a)
#pragma unroll
        for (int b = 0; b < WH_E; ++b) S[a][b] = 0.0;
    for (int m = 0; m < f; ++m) {
        double xr[WH_E], xc[WH_E];
#pragma unroll
        for (int a = 0; a < WH_E; ++a) xr[a] = Xs[m * WH_LD + ty + WH_T * a];
#pragma unroll
        for (int b = 0; b < WH_E; ++b) xc[b] = Xs[m * WH_LD + tx + WH_T * b];
#pragma unroll
        for (int a = 0; a < WH_E; ++a) {
            if (WH_T * a > m) continue;                           // X_mr = 0 for r > m
#pragma unroll
            for (int b = 0; b < WH_E; ++b) S[a][b] += xr[a] * xc[b];
        }
    }
#pragma unroll
    for (int a = 0; a < WH_E; ++a)
#pragma unroll
        for (int b = 0; b < WH_E; ++b) {
            const int r = ty + WH_T * a, c = tx + WH_T * b;
            if (r < f && c < f) {
                B[r * f + c] = (float)X[a][b];
                B[(f + r) * f + c] = (float)S[a][b];
            }
        }
    if (__syncthreads_or(bad) && tid == 0 && info) atomicAdd(info, 1);
}

template <int NB, int KS>
void launch_short(const ShortParams &p, const int64_t first, const int64_t count, hipStream_t st) {
    if (count <= 0) return;
    const dim3 grid((unsigned)count), block(64);
    if (p.rmax > 0.f) {
        if (p.f == 32 * KS) hipLaunchKernelGGL((short_rows_kernel<NB, KS, true, true>), grid, block, 0, st, p, first);
        else hipLaunchKernelGGL((short_rows_kernel<NB, KS, false, true>), grid, block, 0, st, p, first);
    } else {
        if (p.f == 32 * KS) hipLaunchKernelGGL((short_rows_kernel<NB, KS, true, false>), grid, block, 0, st, p, first);
        else hipLaunchKernelGGL((short_rows_kernel<NB, KS, false, false>), grid, block, 0, st, p, first);
    }
}

}  // namespace

// bounds[j], j = 0..ALS_SHORT_MAX_GROUPS: index in `items` of the first whole-row item with at most 16 j entries (items
// are sorted by entry count, descending); bounds[ALS_SHORT_MAX_GROUPS + 1] = the end of the item list.  Rows with (16 (j - 1), 16 j] entries are
// items [bounds[j], bounds[j - 1]); the empty rows [bounds[0], end) go with the first group.
int32_t als_short_rows_launch_(const int32_t *col, const float *val, const int64_t *row_ids, const als_item_t *items,
                               const int64_t *bounds, int32_t nb_max, const float *WZ, int64_t ldw, int32_t f, float alpha,
                               float rmax_f16, float *Xown, int64_t ldx, float *rowdiff, int32_t *info, hipStream_t st) {
    ShortParams p;
    p.rmax = rmax_f16;            // > 0: the fp16 x 2 form of the products with this bound of the ratings; 0: bf16 x 3
    p.col = col; p.val = val; p.row_ids = row_ids; p.items = items; p.WZ = WZ; p.X = Xown; p.rowdiff = rowdiff;
    p.info = info; p.ldw = ldw; p.ldx = ldx; p.f = f; p.alpha = alpha;
    const int64_t end = bounds[ALS_SHORT_MAX_GROUPS + 1];
    for (int j = nb_max; j >= 1; --j) {
        const int64_t first = bounds[j], last = j == 1 ? end : bounds[j - 1];
        if (last - first > 0x7fffffffLL) return als_fail_(ALS_ERR_UNSUPPORTED, "too many short rows for one launch");
        if (f <= 64) {
            switch (j) {
                case 1: launch_short<1, 2>(p, first, last - first, st); break;
                case 2: launch_short<2, 2>(p, first, last - first, st); break;
                case 3: launch_short<3, 2>(p, first, last - first, st); break;
                default: launch_short<4, 2>(p, first, last - first, st); break;
            }
        } else {
            switch (j) {
                case 1: launch_short<1, 4>(p, first, last - first, st); break;
                case 2: launch_short<2, 4>(p, first, last - first, st); break;
                case 3: launch_short<3, 4>(p, first, last - first, st); break;
                case 4: launch_short<4, 4>(p, first, last - first, st); break;
                case 5: launch_short<5, 4>(p, first, last - first, st); break;
                case 6: launch_short<6, 4>(p, first, last - first, st); break;
                case 7: launch_short<7, 4>(p, first, last - first, st); break;
                default: launch_short<8, 4>(p, first, last - first, st); break;
            }
        }
        ALS_HIP_CHECK3(hipGetLastError());
    }
    return ALS_OK;
}

extern "C" {

int32_t als_short_rows_max_groups(int32_t f) {
    if (f <= 32 || f > ALS_MAX_FACTORS_EXACT || f % 4) return 0;
    return f <= 64 ? 4 : 8;
}

int32_t als_plan_short_bounds(const int64_t *rowptr, int64_t n_rows, int32_t chunk_nnz, int64_t *bounds_out) {
    if (!rowptr || n_rows < 0 || !bounds_out) return als_fail_(ALS_ERR_INVALID_ARG, "als_plan_short_bounds: null pointer or negative n_rows");
    if (chunk_nnz < 64 || chunk_nnz % 64) return als_fail_(ALS_ERR_INVALID_ARG, "chunk_nnz must be a positive multiple of 64");
    // the plan lists the chunk items of long rows first, then the whole rows by entry count, descending
    int64_t chunk_items = 0, whole = 0, le[ALS_SHORT_MAX_GROUPS + 1] = {};
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t nnz = rowptr[r + 1] - rowptr[r];
        if (nnz < 0) return als_fail_(ALS_ERR_INVALID_ARG, "rowptr is not non-decreasing at row %lld", (long long)r);
        if (nnz > chunk_nnz) {
            chunk_items += (nnz + chunk_nnz - 1) / chunk_nnz;
            continue;
        }
        whole += 1;
        for (int j = ALS_SHORT_MAX_GROUPS; j >= 0 && nnz <= 16 * j; --j) le[j] += 1;
    }
    for (int j = 0; j <= ALS_SHORT_MAX_GROUPS; ++j) bounds_out[j] = chunk_items + whole - le[j];
    bounds_out[ALS_SHORT_MAX_GROUPS + 1] = chunk_items + whole;
    return ALS_OK;
}

size_t als_whiten_workspace_bytes(int32_t f) {
    if (als_short_rows_max_groups(f) == 0) return 0;
    return (size_t)2 * f * f * sizeof(float) + 256;
}

int32_t als_whiten_f32(const float *gram, float lamb, int32_t f, const float *Y, int64_t n, int64_t ldy, float *WZ,
                       int64_t ldw, int32_t *info_out, void *workspace, size_t ws_bytes, void *stream) {
    if (!gram || !Y || !WZ || !workspace) return als_fail_(ALS_ERR_INVALID_ARG, "als_whiten_f32: null pointer");
    if (als_short_rows_max_groups(f) == 0)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_whiten_f32: f=%d outside (32, %d] or not a multiple of 4", f, ALS_MAX_FACTORS_EXACT);
    if (n < 1 || ldy < f || ldw < 2 * f) return als_fail_(ALS_ERR_INVALID_ARG, "als_whiten_f32: bad sizes (n=%lld ldy=%lld ldw=%lld)",
                                                          (long long)n, (long long)ldy, (long long)ldw);
    if (ldy % 4 || ldw % 4 || (uintptr_t)Y % 16 || (uintptr_t)WZ % 16)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_whiten_f32: factor rows must be 16-byte aligned (ldy, ldw multiples of 4)");
    if (n * (ldw / 4) >= 0xffffffffLL) return als_fail_(ALS_ERR_UNSUPPORTED, "als_whiten_f32: table too large for 32-bit row offsets");
    if (ws_bytes < als_whiten_workspace_bytes(f)) return als_fail_(ALS_ERR_WORKSPACE, "als_whiten_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *B = (float *)(((uintptr_t)workspace + 255) / 256 * 256);
    hipLaunchKernelGGL(whiten_factor_kernel, dim3(1), dim3(WH_T * WH_T), 0, st, gram, lamb, f, B, info_out);
    ALS_HIP_CHECK3(hipGetLastError());
    return als_scores_f32(Y, ldy, nullptr, n, B, f, 2 * (int64_t)f, f, WZ, ldw, stream);
}

}  // extern "C"
