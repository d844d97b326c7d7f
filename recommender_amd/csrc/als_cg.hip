// Conjugate-gradient half-sweep (any f <= 256) and the wide Gram (128 < f <= 256) of libals_hip.so.
//
// The exact path keeps the f x f normal matrix of a row in one wave's accumulator registers,
// which ends at f = 128.  This path never forms the matrix:
//
//     A_u v = (Y^T Y) v + sum_j alpha r_uj (y_j . v) y_j + lamb v
//
// and runs a FIXED number of conjugate-gradient steps per row, warm-started at the row's current
// value.  All rows of the CSR advance in lock step, so the dense part (Y^T Y) v of every row is
// one GEMM per step on the matrix cores (V [n_rows, f] times the f x f Gram), and the sparse part
// is one pass of the gather kernel per step:
//
//   lane group g = lane>>4 takes one gathered row per k-step, lane j = lane&15 holds its
//   elements 64h + 4j .. 64h + 4j + 3 (one 16-byte load per 64 factors).  The dot product
//   y_j . v is VW fmas and a 4-step DPP butterfly inside the 16 lanes; the update
//   acc += coef * y_j is VW fmas.  The four lane groups' sums are combined once per row in a
//   fixed order, long rows are cut at multiples of chunk_nnz and combined in chunk order, so the
//   result depends only on the inputs and chunk_nnz - never on the launch geometry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "als_hip.h"

int als_fail_(int code, const char *fmt, ...);   // als_kernels.hip
int32_t als_sum_rowdiff_(const float *rowdiff, int64_t n_rows, double *partials, double *out, hipStream_t st);  // als_kernels.hip
int32_t als_half_sweep_tiles_(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                              const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                              const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                              const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, int32_t f,
                              float alpha, int32_t cg_steps, float *Xown, int64_t ldx, double *absdiff_out,
                              int32_t *info_out, const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes,
                              void *stream);                                                                // als_kernels.hip
size_t als_half_sweep_wide_workspace_bytes_(int64_t n_rows, int64_t n_slots);                              // als_kernels.hip
int32_t als_half_sweep_wide_(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                             const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                             const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                             const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, float lamb, int32_t f,
                             float alpha, int32_t cg_steps, float *Xown, int64_t ldx, double *absdiff_out,
                             int32_t *info_out, const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes,
                             void *stream);                                                                 // als_kernels.hip

#define ALS_HIP_CHECK3(expr)                                                                      \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return als_fail_(ALS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <int CTRL>
__device__ __forceinline__ float dpp_f(const float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, the same bits in every lane of the row (every step adds two
// values that both lanes of a pair hold, and addition commutes)
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);  // row_half_mirror
    x += dpp_f<0x140>(x);  // row_mirror
    return x;
}

// ----------------------------------------------------------------------------- row vectors
// Lane j (0..15) of a lane group holds, in register t, element 64*(t>>2) + 4*j + (t&3).
// Elements >= f read as 0 and are never stored.  VEC: 16-byte accesses (f % 4 == 0, aligned rows).
template <int VW, bool VEC>
__device__ __forceinline__ void load_row(const float *__restrict__ row, const int f, const int j, float (&v)[VW]) {
#pragma unroll
    for (int h = 0; h < VW / 4; ++h) {
        const int i0 = 64 * h + 4 * j;
        if constexpr (VEC) {
            const bool ok = i0 < f;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(row + (ok ? i0 : 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * h + q] = ok ? x[q] : 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = i0 + q < f;
                const float x = row[ok ? i0 + q : 0];
                v[4 * h + q] = ok ? x : 0.f;
            }
        }
    }
}
template <int VW, bool VEC>
__device__ __forceinline__ void store_row(float *__restrict__ row, const int f, const int j, const float (&v)[VW]) {
#pragma unroll
    for (int h = 0; h < VW / 4; ++h) {
        const int i0 = 64 * h + 4 * j;
        if constexpr (VEC) {
            if (i0 < f) *reinterpret_cast<f32x4 *>(row + i0) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + q < f) row[i0 + q] = v[4 * h + q];
        }
    }
}
template <int VW>
__device__ __forceinline__ float row_dot(const float (&a)[VW], const float (&b)[VW]) {
    float d = a[0] * b[0];
#pragma unroll
    for (int t = 1; t < VW; ++t) d = fmaf(a[t], b[t], d);
    return row_sum16(d);
}

// ----------------------------------------------------------------------------- gather pass
template <int VW>
struct CgStageGeo {
    static constexpr int GS = VW > 8 ? 2 : 4;   // k-steps (of 4 gathered rows) per pipeline stage
    static constexpr int NG = 16 / GS;          // stages per 64-entry chunk
};
template <int VW>
struct CgStage {
    float v[CgStageGeo<VW>::GS][VW];
    float w[CgStageGeo<VW>::GS];
    float cw[CgStageGeo<VW>::GS];
};

template <int VW, bool VEC, bool INIT>
__device__ __forceinline__ void cg_issue(CgStage<VW> &st, const int g, const int cur_col, const float cur_w,
                                         const float cur_cw, const float *__restrict__ Y, const int64_t ld,
                                         const int f, const int lane) {
    constexpr int GS = CgStageGeo<VW>::GS;
    const int grp = lane >> 4, j = lane & 15;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        const int src = g * (GS * 4) + s * 4 + grp;
        const int c = __shfl(cur_col, src);
        st.w[s] = __shfl(cur_w, src);
        if constexpr (INIT) st.cw[s] = __shfl(cur_cw, src);
        load_row<VW, VEC>(Y + (int64_t)c * ld, f, j, st.v[s]);
    }
}

// INIT:  acc += (c_e - w_e (y_e . dv)) y_e     (right-hand side minus the sparse part of A x)
// else:  acc += w_e (y_e . dv) y_e             (the sparse part of A p)
template <int VW, bool INIT>
__device__ __forceinline__ void cg_compute(const CgStage<VW> &st, const float (&dv)[VW], float (&acc)[VW]) {
#pragma unroll
    for (int s = 0; s < CgStageGeo<VW>::GS; ++s) {
        const float d = row_dot<VW>(st.v[s], dv);
        const float coef = INIT ? fmaf(-st.w[s], d, st.cw[s]) : st.w[s] * d;
#pragma unroll
        for (int t = 0; t < VW; ++t) acc[t] = fmaf(coef, st.v[s][t], acc[t]);
    }
}

// Entries [s, e) of the CSR (wave uniform, e > s), w = alpha r, c = 1 + alpha r (0 for a stored 0).
// Returns whether any entry of the range has a non-zero rating (stored zeros are dropped, ALS.py:79: a
// row that holds nothing else has the exact solution 0).
template <int VW, bool VEC, bool INIT>
__device__ __forceinline__ bool cg_accumulate(float (&acc)[VW], const float (&dv)[VW], const int32_t *__restrict__ col,
                                              const float *__restrict__ val, const int64_t s, const int64_t e,
                                              const float *__restrict__ Y, const int64_t ld, const int f,
                                              const float alpha, const int lane) {
    constexpr int GS = CgStageGeo<VW>::GS, NG = CgStageGeo<VW>::NG;
    CgStage<VW> A, B;
    int cur_col, nxt_col = 0;
    float cur_w, cur_cw, nxt_w = 0.f, nxt_cw = 0.f;
    bool kept = false;
    auto load_chunk = [&](const int64_t base, int &c, float &w, float &cw) {
        const int64_t idx = base + lane;
        const bool ok = idx < e;
        const int64_t ii = ok ? idx : s;
        const int cc = col[ii];
        const float rr = val[ii];
        const float ww = alpha * rr;
        c = cc;                                   // a lane past the end re-reads entry s with weight 0
        w = ok ? ww : 0.f;
        cw = (ok && rr != 0.f) ? 1.f + ww : 0.f;
        kept = kept || (ok && rr != 0.f);
    };
    load_chunk(s, cur_col, cur_w, cur_cw);
    cg_issue<VW, VEC, INIT>(A, 0, cur_col, cur_w, cur_cw, Y, ld, f, lane);
    for (int64_t base = s; base < e; base += 64) {
        const int64_t nb = base + 64;
        const bool more = nb < e;
        if (more) load_chunk(nb, nxt_col, nxt_w, nxt_cw);
        const int rem = more ? 64 : (int)(e - base);
        const int ng = (rem + 4 * GS - 1) / (4 * GS);   // stages of this chunk that hold entries
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            CgStage<VW> &cur = (g & 1) ? B : A;
            CgStage<VW> &nxt = (g & 1) ? A : B;
            if (g + 1 < NG) {
                if (ng > g + 1) cg_issue<VW, VEC, INIT>(nxt, g + 1, cur_col, cur_w, cur_cw, Y, ld, f, lane);
            } else if (more) {
                cur_col = nxt_col;
                cur_w = nxt_w;
                cur_cw = nxt_cw;
                cg_issue<VW, VEC, INIT>(nxt, 0, cur_col, cur_w, cur_cw, Y, ld, f, lane);
            }
            if (ng > g) cg_compute<VW, INIT>(cur, dv, acc);
        }
    }
    return __any(kept);
}

// the 4 lane groups hold partial sums of the same elements: (g0 + g1) + (g2 + g3) in every group
template <int VW>
__device__ __forceinline__ void reduce_groups(float (&acc)[VW]) {
#pragma unroll
    for (int t = 0; t < VW; ++t) {
        acc[t] += __shfl_xor(acc[t], 16);
        acc[t] += __shfl_xor(acc[t], 32);
    }
}

// ----------------------------------------------------------------------------- kernels
struct CgParams {
    const int64_t *rowptr;
    const int32_t *col;
    const float *val;
    const int64_t *row_ids;
    const als_item_t *items;
    const als_long_row_t *long_rows;
    const float *Y;
    float *X;           // own factor: x_old, and the result of the last step
    const float *xin;   // iterate before this step, per CSR row (ld = ldw); null: take it from X
    float *xout;        // iterate after this step (not written by the last step)
    float *P, *R, *Z, *RS;
    float *slots;
    float *rowdiff;
    int32_t *info;
    int64_t ldy, ldx, ldw;
    int32_t f, chunk_nnz;
    float alpha, lamb;
};

enum { CG_INIT = 0, CG_ITER = 1, CG_LAST = 2 };

// What a row does once the sparse sum `acc` over all its entries is known.  `dv` is the vector the
// pass multiplied by: x for CG_INIT, p otherwise.  Vectors are replicated over the 4 lane groups;
// lane group 0 stores.
template <int VW, bool VEC, int MODE>
__device__ __forceinline__ void cg_finish(const CgParams &p, const int row, const bool nonempty, float (&acc)[VW],
                                          const float (&dv)[VW], const int lane) {
    const int j = lane & 15;
    const bool writer = lane < 16;
    const int64_t w_off = (int64_t)row * p.ldw;
    if constexpr (MODE == CG_INIT) {
        float r[VW];
        float rs = 0.f;
        if (nonempty) {
            float z[VW];
            load_row<VW, VEC>(p.Z + w_off, p.f, j, z);
#pragma unroll
            for (int t = 0; t < VW; ++t) r[t] = acc[t] - z[t] - p.lamb * dv[t];
            rs = row_dot<VW>(r, r);
        } else {
#pragma unroll
            for (int t = 0; t < VW; ++t) r[t] = 0.f;
        }
        if (writer) {
            store_row<VW, VEC>(p.R + w_off, p.f, j, r);
            store_row<VW, VEC>(p.P + w_off, p.f, j, r);
            if (lane == 0) p.RS[row] = rs;
        }
    } else {
        const int64_t xr = p.row_ids ? p.row_ids[row] : (int64_t)row;
        float x[VW];
        if (p.xin) load_row<VW, VEC>(p.xin + w_off, p.f, j, x);
        else load_row<VW, VEC>(p.X + xr * p.ldx, p.f, j, x);
        float rn[VW], pn[VW];
        float rs_new = 0.f;
        if (nonempty) {
            float z[VW], r[VW];
            load_row<VW, VEC>(p.Z + w_off, p.f, j, z);
            load_row<VW, VEC>(p.R + w_off, p.f, j, r);
            const float rs = p.RS[row];
#pragma unroll
            for (int t = 0; t < VW; ++t) acc[t] = acc[t] + z[t] + p.lamb * dv[t];     // A p
            const float pAp = row_dot<VW>(dv, acc);
            const bool ok = rs > 0.f && pAp > 0.f;
            if (rs > 0.f && !(pAp > 0.f) && lane == 0 && p.info) atomicAdd(p.info, 1);   // not positive definite
            const float a = ok ? rs / pAp : 0.f;
#pragma unroll
            for (int t = 0; t < VW; ++t) {
                x[t] = fmaf(a, dv[t], x[t]);
                rn[t] = fmaf(-a, acc[t], r[t]);
            }
            const float rr = row_dot<VW>(rn, rn);
            const float beta = ok ? rr / rs : 0.f;
#pragma unroll
            for (int t = 0; t < VW; ++t) pn[t] = fmaf(beta, dv[t], rn[t]);
            rs_new = ok ? rr : 0.f;          // a step that could not be taken ends the row's iteration
        } else {
#pragma unroll
            for (int t = 0; t < VW; ++t) x[t] = rn[t] = pn[t] = 0.f;               // no entries: x = 0 exactly
        }
        if constexpr (MODE == CG_ITER) {
            if (writer) {
                store_row<VW, VEC>(p.xout + w_off, p.f, j, x);
                store_row<VW, VEC>(p.R + w_off, p.f, j, rn);
                store_row<VW, VEC>(p.P + w_off, p.f, j, pn);
                if (lane == 0) p.RS[row] = rs_new;
            }
        } else {
            float xo[VW];
            load_row<VW, VEC>(p.X + xr * p.ldx, p.f, j, xo);
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < VW; ++t) d += fabsf(xo[t] - x[t]);
            d = row_sum16(d);
            if (writer) {
                store_row<VW, VEC>(p.X + xr * p.ldx, p.f, j, x);
                if (lane == 0) p.rowdiff[row] = d;
            }
        }
    }
}

template <int VW, bool VEC, int MODE>
__device__ __forceinline__ void cg_load_direction(const CgParams &p, const int row, const int j, float (&dv)[VW]) {
    if constexpr (MODE == CG_INIT) {
        const int64_t xr = p.row_ids ? p.row_ids[row] : (int64_t)row;
        load_row<VW, VEC>(p.X + xr * p.ldx, p.f, j, dv);
    } else {
        load_row<VW, VEC>(p.P + (int64_t)row * p.ldw, p.f, j, dv);
    }
}

template <int VW, bool VEC, int MODE>
__global__ __launch_bounds__(64) void cg_items_kernel(const CgParams p) {
    const int lane = threadIdx.x;
    const als_item_t it = p.items[blockIdx.x];
    const int row = it.row;
    const int64_t s = it.begin, e = it.begin + it.count;
    float dv[VW], acc[VW];
    cg_load_direction<VW, VEC, MODE>(p, row, lane & 15, dv);
#pragma unroll
    for (int t = 0; t < VW; ++t) acc[t] = 0.f;
    bool kept = false;
    if (e > s) {
        kept = cg_accumulate<VW, VEC, MODE == CG_INIT>(acc, dv, p.col, p.val, s, e, p.Y, p.ldy, p.f, p.alpha, lane);
        reduce_groups<VW>(acc);
    }
    if (it.chunk >= 0) {
        if (lane < 16) store_row<VW, true>(p.slots + (int64_t)it.slot * (16 * VW), 16 * VW, lane, acc);
        return;
    }
    cg_finish<VW, VEC, MODE>(p, row, kept, acc, dv, lane);      // (a row cut into chunks counts as holding ratings)
}

// A long row: add its chunk sums in chunk order, then finish as above.
template <int VW, bool VEC, int MODE>
__global__ __launch_bounds__(64) void cg_long_rows_kernel(const CgParams p) {
    const int lane = threadIdx.x;
    const als_long_row_t lr = p.long_rows[blockIdx.x];
    float dv[VW], acc[VW];
    cg_load_direction<VW, VEC, MODE>(p, lr.row, lane & 15, dv);
    load_row<VW, true>(p.slots + (int64_t)lr.first_slot * (16 * VW), 16 * VW, lane & 15, acc);
    for (int c = 1; c < lr.n_chunks; ++c) {
        float part[VW];
        load_row<VW, true>(p.slots + (int64_t)(lr.first_slot + c) * (16 * VW), 16 * VW, lane & 15, part);
#pragma unroll
        for (int t = 0; t < VW; ++t) acc[t] += part[t];
    }
    cg_finish<VW, VEC, MODE>(p, lr.row, true, acc, dv, lane);
}

inline int cg_vw_for(int f) { return f <= 64 ? 4 : f <= 128 ? 8 : f <= 192 ? 12 : 16; }

template <int VW, bool VEC, int MODE>
int32_t cg_pass_launch(const CgParams &p, int64_t n_items, int64_t n_long, hipStream_t st) {
    if (n_items > 0) {
        hipLaunchKernelGGL((cg_items_kernel<VW, VEC, MODE>), dim3((unsigned)n_items), dim3(64), 0, st, p);
        ALS_HIP_CHECK3(hipGetLastError());
    }
    if (n_long > 0) {
        hipLaunchKernelGGL((cg_long_rows_kernel<VW, VEC, MODE>), dim3((unsigned)n_long), dim3(64), 0, st, p);
        ALS_HIP_CHECK3(hipGetLastError());
    }
    return ALS_OK;
}
template <int VW, bool VEC>
int32_t cg_pass_mode(const CgParams &p, int mode, int64_t n_items, int64_t n_long, hipStream_t st) {
    switch (mode) {
        case CG_INIT: return cg_pass_launch<VW, VEC, CG_INIT>(p, n_items, n_long, st);
        case CG_ITER: return cg_pass_launch<VW, VEC, CG_ITER>(p, n_items, n_long, st);
        default: return cg_pass_launch<VW, VEC, CG_LAST>(p, n_items, n_long, st);
    }
}
int32_t cg_pass(const CgParams &p, int mode, bool vec, int64_t n_items, int64_t n_long, hipStream_t st) {
    switch (cg_vw_for(p.f)) {
        case 4: return vec ? cg_pass_mode<4, true>(p, mode, n_items, n_long, st) : cg_pass_mode<4, false>(p, mode, n_items, n_long, st);
        case 8: return vec ? cg_pass_mode<8, true>(p, mode, n_items, n_long, st) : cg_pass_mode<8, false>(p, mode, n_items, n_long, st);
        case 12: return vec ? cg_pass_mode<12, true>(p, mode, n_items, n_long, st) : cg_pass_mode<12, false>(p, mode, n_items, n_long, st);
        default: return vec ? cg_pass_mode<16, true>(p, mode, n_items, n_long, st) : cg_pass_mode<16, false>(p, mode, n_items, n_long, st);
    }
}

// ----------------------------------------------------------------------------- wide Gram
// 128 < f <= 256: the Gram is cut into 64 x 64 blocks (NB = ceil(f/64) per side); one wave sums
// one block pair (bi <= bj) over ALS_GRAM_WIDE_CHUNK_ROWS rows as 4 x 4 tiles of
// v_mfma_f32_16x16x4_f32.  Lane (g, j) loads elements 64b + 4j .. 4j+3 of row 4k+g for both blocks:
// register ta of the first load is the A operand of tile row ta, register tb of the second the B
// operand of tile column tb.  Tile (ta, tb), lane (g, j), register r holds
// G[64bi + 4(4g + r) + ta][64bj + 4j + tb].
constexpr int WIDE_TILE_FLOATS = 16 * 256;   // floats per block pair in a partial

__device__ __forceinline__ void wide_pair(const int pair, const int nb, int &bi, int &bj) {
    int q = pair;
    bi = 0;
    while (q >= nb - bi) {
        q -= nb - bi;
        ++bi;
    }
    bj = bi + q;
}

template <bool VEC>
__device__ __forceinline__ f32x4 wide_load(const float *__restrict__ row, const int i0, const int f) {
    if constexpr (VEC) {
        const bool ok = i0 < f;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(row + (ok ? i0 : 0));
        return ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        f32x4 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = i0 + q < f;
            const float v = row[ok ? i0 + q : 0];
            x[q] = ok ? v : 0.f;
        }
        return x;
    }
}

template <bool VEC>
__global__ __launch_bounds__(64) void gram_wide_chunk_kernel(const float *__restrict__ Y, const int64_t row0,
                                                            const int64_t n_end, const int64_t ld, const int f,
                                                            const int nb, float *__restrict__ out) {
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    int bi, bj;
    wide_pair(blockIdx.y, nb, bi, bj);
    const int64_t s = row0 + (int64_t)blockIdx.x * ALS_GRAM_WIDE_CHUNK_ROWS;
    const int64_t e = min(n_end, s + (int64_t)ALS_GRAM_WIDE_CHUNK_ROWS);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 4;   // k-steps in flight
    for (int64_t base = s; base < e; base += 4 * UN) {
        f32x4 xa[UN], xb[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t r = base + 4 * u + g;
            const bool ok = r < e;
            const float *row = Y + (ok ? r : s) * ld;
            xa[u] = wide_load<VEC>(row, 64 * bi + 4 * j, ok ? f : 0);
            xb[u] = wide_load<VEC>(row, 64 * bj + 4 * j, ok ? f : 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][a], xb[u][b], acc[a][b], 0, 0, 0);
    }
    const int np = nb * (nb + 1) / 2;
    float *P = out + ((int64_t)blockIdx.x * np + blockIdx.y) * WIDE_TILE_FLOATS;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[((a * 4 + b) * 4 + r) * 64 + lane] = acc[a][b][r];
}

// out[g][e] = in[g*group][e] + in[g*group+1][e] + ... (in order)
__global__ void wide_sum_groups_kernel(const float *__restrict__ in, const int64_t n_in, const int group,
                                       const int elems, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    const int64_t g = blockIdx.y;
    const int64_t first = g * group, last = min(n_in, first + group);
    float acc = in[first * elems + e];
    int64_t i = first + 1;
    for (; i + 7 < last; i += 8) {          // 8 loads in flight, added in order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = in[(i + q) * elems + e];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; i < last; ++i) acc += in[i * elems + e];
    out[g * elems + e] = acc;
}

// Sums the 65 536-row partials in order and writes the row-major Gram (both triangles from the
// upper one, so it is exactly symmetric) and, in tiles_out, the same partial layout summed.
__global__ void gram_wide_finalize_kernel(const float *__restrict__ super, const int64_t n_super, const int f,
                                          const int nb, float *__restrict__ gram_out, float *__restrict__ tiles_out) {
    const int np = nb * (nb + 1) / 2;
    const int elems = np * WIDE_TILE_FLOATS;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    float v = super[e];
    for (int64_t i = 1; i < n_super; ++i) v += super[i * elems + e];
    tiles_out[e] = v;
    const int pair = e / WIDE_TILE_FLOATS, w = e % WIDE_TILE_FLOATS;
    const int lane = w & 63, q = w >> 6, r = q & 3, tb = (q >> 2) & 3, ta = q >> 4;
    int bi, bj;
    wide_pair(pair, nb, bi, bj);
    const int a = 64 * bi + 4 * (4 * (lane >> 4) + r) + ta, b = 64 * bj + 4 * (lane & 15) + tb;
    if (a < f && b < f && a <= b) {
        gram_out[a * f + b] = v;
        gram_out[b * f + a] = v;
    }
}

}  // namespace

// ============================================================================= internal (als_kernels.hip)
int als_wide_blocks_(int f) { return (f + 63) / 64; }
size_t als_gram_wide_tiles_bytes_(int f) {
    const int nb = als_wide_blocks_(f);
    return (size_t)(nb * (nb + 1) / 2) * WIDE_TILE_FLOATS * sizeof(float);
}
size_t als_gram_wide_workspace_bytes_(int64_t n, int f) {
    const int64_t n_chunks = std::max<int64_t>(1, ceil_div(n, ALS_GRAM_WIDE_CHUNK_ROWS));
    const int64_t n_super = std::max<int64_t>(1, ceil_div(n, ALS_GRAM_SUPER_ROWS));
    return (size_t)(n_chunks + n_super) * als_gram_wide_tiles_bytes_(f) + 256;
}
int32_t als_gram_wide_partials_(const float *Y, int64_t row_begin, int64_t row_end, int32_t f, int64_t ldy,
                                float *super_out, void *workspace, hipStream_t st) {
    const int nb = als_wide_blocks_(f), np = nb * (nb + 1) / 2;
    const int64_t n = row_end - row_begin;
    const int64_t n_chunks = ceil_div(n, ALS_GRAM_WIDE_CHUNK_ROWS), n_super = ceil_div(n, ALS_GRAM_SUPER_ROWS);
    float *chunks = (float *)workspace;
    const bool vec = (f % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)Y % 16 == 0);
    const dim3 grid((unsigned)n_chunks, (unsigned)np), block(64);
    if (vec) hipLaunchKernelGGL(gram_wide_chunk_kernel<true>, grid, block, 0, st, Y, row_begin, row_end, ldy, f, nb, chunks);
    else hipLaunchKernelGGL(gram_wide_chunk_kernel<false>, grid, block, 0, st, Y, row_begin, row_end, ldy, f, nb, chunks);
    ALS_HIP_CHECK3(hipGetLastError());
    const int elems = np * WIDE_TILE_FLOATS;
    hipLaunchKernelGGL(wide_sum_groups_kernel, dim3((elems + 255) / 256, (unsigned)n_super), dim3(256), 0, st, chunks,
                       n_chunks, ALS_GRAM_SUPER_ROWS / ALS_GRAM_WIDE_CHUNK_ROWS, elems, super_out);
    ALS_HIP_CHECK3(hipGetLastError());
    return ALS_OK;
}
int32_t als_gram_wide_finalize_(const float *super_in, int64_t n_super, int32_t f, float *gram_out, float *tiles_out,
                                hipStream_t st) {
    const int nb = als_wide_blocks_(f);
    const int elems = nb * (nb + 1) / 2 * WIDE_TILE_FLOATS;
    hipLaunchKernelGGL(gram_wide_finalize_kernel, dim3((elems + 255) / 256), dim3(256), 0, st, super_in, n_super, f, nb,
                       gram_out, tiles_out);
    ALS_HIP_CHECK3(hipGetLastError());
    return ALS_OK;
}

// ============================================================================= C ABI
extern "C" {

// workspace: P, R, Z, XW [n_rows, ldw] each; RS [n_rows]; slots [n_slots, 16 VW]; rowdiff [n_rows];
// the double partials of the |delta| sum
size_t als_half_sweep_cg_workspace_bytes(int64_t n_rows, int64_t n_slots, int32_t f) {
    if (f < 1 || f > ALS_MAX_FACTORS || n_rows < 0 || n_slots < 0) return 0;
    const size_t ldw = align_up((size_t)f, 4);
    size_t b = 4 * align_up((size_t)n_rows * ldw * sizeof(float), 256);
    b += align_up((size_t)n_rows * sizeof(float), 256) * 2;
    b += align_up((size_t)n_slots * 16 * cg_vw_for(f) * sizeof(float), 256);
    b += align_up((size_t)(ceil_div(n_rows, 8192) + 1) * sizeof(double), 256);
    b += 256;
    // f <= ALS_MAX_FACTORS_EXACT with gram_tiles given runs on the accumulator tiles: that path's workspace
    if (f <= ALS_MAX_FACTORS_EXACT) b = std::max(b, als_half_sweep_workspace_bytes(n_rows, n_slots, f));
    else b = std::max(b, als_half_sweep_wide_workspace_bytes_(n_rows, n_slots));
    return b;
}

int32_t als_half_sweep_cg_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                              const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                              const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                              const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram, const float *gram_tiles,
                              float lamb, int32_t f, float alpha, int32_t cg_steps, int32_t form, float *Xown,
                              int64_t ldx, double *absdiff_out, int32_t *info_out, const als_sweep_opts_t *opts,
                              void *workspace, size_t ws_bytes, void *stream) {
    if (cg_steps < 1 || cg_steps > 1024) return als_fail_(ALS_ERR_INVALID_ARG, "cg_steps=%d outside [1, 1024]", cg_steps);
    if (form != ALS_CG_FORM_AUTO && form != ALS_CG_FORM_LOCK_STEP && form != ALS_CG_FORM_MATRIX)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: unknown form %d", form);
    if (form == ALS_CG_FORM_AUTO)
        form = (gram_tiles && f <= ALS_MAX_FACTORS_EXACT) ? ALS_CG_FORM_MATRIX : ALS_CG_FORM_LOCK_STEP;
    if (form == ALS_CG_FORM_MATRIX && !gram_tiles)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: the matrix form needs gram_tiles");
    if (form == ALS_CG_FORM_LOCK_STEP) gram_tiles = nullptr;
    if (gram_tiles && f <= ALS_MAX_FACTORS_EXACT)      // one gather pass: form the row's matrix, iterate in registers
        return als_half_sweep_tiles_(rowptr, col, val, n_rows, row_ids, items, n_items, long_rows, n_long, n_slots,
                                     chunk_nnz, Yopp, n_opp, ldy, gram_tiles, f, alpha, cg_steps, Xown, ldx, absdiff_out,
                                     info_out, opts, workspace, ws_bytes, stream);
    if (gram_tiles && f <= ALS_MAX_FACTORS)            // the same with the row's tiles spread over 4 waves
        return als_half_sweep_wide_(rowptr, col, val, n_rows, row_ids, items, n_items, long_rows, n_long, n_slots, chunk_nnz,
                                    Yopp, n_opp, ldy, gram_tiles, lamb, f, alpha, cg_steps, Xown, ldx, absdiff_out, info_out,
                                    opts, workspace, ws_bytes, stream);
    if (!rowptr || !col || !val || !items || !Yopp || !gram || !Xown || !workspace)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: null pointer");
    if (n_rows < 1 || n_rows > INT32_MAX || n_items < n_rows || n_long < 0 || n_slots < 0 || n_opp < 1)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: bad sizes (n_rows=%lld n_items=%lld n_opp=%lld)",
                         (long long)n_rows, (long long)n_items, (long long)n_opp);
    if ((n_long > 0 && !long_rows) || (n_long > 0) != (n_slots > 0) || n_items != n_rows - n_long + n_slots)
        return als_fail_(ALS_ERR_INVALID_ARG, "plan sizes are inconsistent (n_items=%lld n_long=%lld n_slots=%lld)",
                         (long long)n_items, (long long)n_long, (long long)n_slots);
    if (chunk_nnz < 64 || chunk_nnz % 64) return als_fail_(ALS_ERR_INVALID_ARG, "chunk_nnz must be a positive multiple of 64");
    if (f < 1 || f > ALS_MAX_FACTORS) return als_fail_(ALS_ERR_UNSUPPORTED, "f=%d outside [1, %d]", f, ALS_MAX_FACTORS);
    if (ldy < f || ldx < f) return als_fail_(ALS_ERR_INVALID_ARG, "leading dimension smaller than f");
    if (n_items > 0x7fffffffLL) return als_fail_(ALS_ERR_UNSUPPORTED, "too many work items for one launch");
    if (ws_bytes < als_half_sweep_cg_workspace_bytes(n_rows, n_slots, f))
        return als_fail_(ALS_ERR_WORKSPACE, "cg workspace too small: %zu < %zu", ws_bytes,
                         als_half_sweep_cg_workspace_bytes(n_rows, n_slots, f));
    hipStream_t st = (hipStream_t)stream;
    if (opts && ((opts->tap_begin == nullptr) != (opts->tap_end == nullptr)))
        return als_fail_(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: pass both tap events or neither");
    if (opts && opts->tap_begin) ALS_HIP_CHECK3(hipEventRecord((hipEvent_t)opts->tap_begin, st));
    const int64_t ldw = (int64_t)align_up((size_t)f, 4);
    char *ws = (char *)workspace;
    const size_t vec_bytes = align_up((size_t)n_rows * ldw * sizeof(float), 256);
    CgParams p;
    p.rowptr = rowptr; p.col = col; p.val = val; p.row_ids = row_ids; p.items = items; p.long_rows = long_rows;
    p.Y = Yopp; p.X = Xown;
    p.P = (float *)ws; ws += vec_bytes;
    p.R = (float *)ws; ws += vec_bytes;
    p.Z = (float *)ws; ws += vec_bytes;
    float *XW = (float *)ws; ws += vec_bytes;
    p.RS = (float *)ws; ws += align_up((size_t)n_rows * sizeof(float), 256);
    p.rowdiff = (float *)ws; ws += align_up((size_t)n_rows * sizeof(float), 256);
    p.slots = (float *)ws; ws += align_up((size_t)n_slots * 16 * cg_vw_for(f) * sizeof(float), 256);
    double *diff_partials = (double *)ws;
    p.info = info_out; p.ldy = ldy; p.ldx = ldx; p.ldw = ldw; p.f = f; p.chunk_nnz = chunk_nnz; p.alpha = alpha;
    p.lamb = lamb;
    const bool vec = (f % 4 == 0) && (ldy % 4 == 0) && (ldx % 4 == 0) && ((uintptr_t)Yopp % 16 == 0) &&
                     ((uintptr_t)Xown % 16 == 0) && ((uintptr_t)workspace % 16 == 0);

    // Z = V G for all rows (G symmetric, so row i of G is its column i): V = the rows' x (pass 0) or p
    auto dense_part = [&](const float *V, int64_t ldv, const int64_t *ids) -> int32_t {
        const int64_t step = 65535LL * 64;
        for (int64_t r0 = 0; r0 < n_rows; r0 += step) {
            const int64_t nr = std::min(step, n_rows - r0);
            const int32_t rc = als_scores_f32(ids ? V : V + r0 * ldv, ldv, ids ? ids + r0 : nullptr, nr, gram, f, f, f,
                                              p.Z + r0 * ldw, ldw, stream);
            if (rc) return rc;
        }
        return ALS_OK;
    };

    p.xin = nullptr; p.xout = nullptr;
    int32_t rc = dense_part(Xown, ldx, row_ids);
    if (rc) return rc;
    rc = cg_pass(p, CG_INIT, vec, n_items, n_long, st);
    if (rc) return rc;
    for (int step = 0; step < cg_steps; ++step) {
        rc = dense_part(p.P, ldw, nullptr);
        if (rc) return rc;
        const bool last = step == cg_steps - 1;
        p.xin = step == 0 ? nullptr : XW;
        p.xout = XW;
        rc = cg_pass(p, last ? CG_LAST : CG_ITER, vec, n_items, n_long, st);
        if (rc) return rc;
    }
    if (opts && opts->tap_end) ALS_HIP_CHECK3(hipEventRecord((hipEvent_t)opts->tap_end, st));
    if (absdiff_out) return als_sum_rowdiff_(p.rowdiff, n_rows, diff_partials, absdiff_out, st);
    return ALS_OK;
}

}  // extern "C"
