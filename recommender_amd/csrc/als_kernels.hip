// libals_hip.so - hand-written gfx950 (MI355X, CDNA4) kernels for implicit ALS and the
// C ABI declared in include/als_hip.h.
//
// Layout of one wavefront's work (all kernels use 64-thread workgroups = one wave):
//
//   A 16-lane group g = lane>>4 handles one gathered factor row per k-step, lane
//   j = lane&15 of the group holds VW = FPAD/16 of its elements (one 16-byte load for
//   FPAD = 64).  In the fp32 mode those registers are directly the A and B operands of
//   v_mfma_f32_16x16x4_f32 (A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15]); in the default
//   bf16 x 3 mode eight k-steps are cut into bf16 pieces and fed, again without any data
//   movement, to v_mfma_f32_16x16x32_bf16 (see "bf16 x 3 accumulate" below).  Either way the
//   rank-k update  sum_k w_k y_k y_k^T  of the f x f normal matrix needs no LDS staging and no
//   cross-lane traffic.  Tile (ta, tb) element (i, j) is matrix element
//   (fidx(i, ta), fidx(j, tb)); only tiles ta <= tb are computed (symmetry).
//
//   Rows are visited in CSR order, long rows are cut at fixed multiples of chunk_nnz and
//   combined in chunk order: results do not depend on the launch geometry.
//
//   The f x f solve is a right-looking blocked Cholesky that runs on the accumulator tiles in
//   place (see "blocked solve" below): no LDS, no layout change.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "als_hip.h"

int32_t als_sum_rowdiff_(const float *rowdiff, int64_t n_rows, double *partials, double *out, hipStream_t st);
int32_t als_half_sweep_tiles_(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                              const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                              const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                              const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, int32_t f,
                              float alpha, int32_t cg_steps, float *Xown, int64_t ldx, double *absdiff_out,
                              int32_t *info_out, const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes,
                              void *stream);
// short rows in their entry-space form, als_short_rows.hip
int32_t als_short_rows_launch_(const int32_t *col, const float *val, const int64_t *row_ids, const als_item_t *items,
                               const int64_t *bounds, int32_t nb_max, const float *WZ, int64_t ldw, int32_t f, float alpha,
                               float rmax_f16, float *Xown, int64_t ldx, float *rowdiff, int32_t *info, hipStream_t st);
// wide Gram (128 < f <= 256), als_cg.hip
size_t als_gram_wide_tiles_bytes_(int f);
size_t als_gram_wide_workspace_bytes_(int64_t n, int f);
int32_t als_gram_wide_partials_(const float *Y, int64_t row_begin, int64_t row_end, int32_t f, int64_t ldy,
                                float *super_out, void *workspace, hipStream_t st);
int32_t als_gram_wide_finalize_(const float *super_in, int64_t n_super, int32_t f, float *gram_out, float *tiles_out,
                                hipStream_t st);

// Scratch of als_half_sweep_csr_static_f32, the entry point for callers that cannot allocate: chunk partials of
// long rows, the per-row |delta| and its partial sums.  Every GPU the library runs on holds its own copy at its own
// address; in host code the name is only a handle for hipGetSymbolAddress (static_scratch_ptr below), never an address.
static __device__ __align__(256) unsigned char scratch[ALS_STATIC_SCRATCH_BYTES];

namespace {

// ----------------------------------------------------------------------------- errors
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define ALS_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(ALS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Device address of `scratch` on the current device, resolved once per device (hipGetSymbolAddress) and kept.
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[MAX_DEVICES] = {};
int32_t static_scratch_ptr(void **out) {
    int dev = -1;
    ALS_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(ALS_ERR_UNSUPPORTED, "device ordinal %d outside [0, %d)", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        ALS_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(ALS_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch buffer");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return ALS_OK;
}

#include "als_tiles.inc"

#ifndef ALS_SWEEP_WAVES_F128
#define ALS_SWEEP_WAVES_F128 1  // 64 < f <= 128: 144 accumulator registers + the stages need > 256 registers
#endif
#ifndef ALS_STAGE_F128
#define ALS_STAGE_F128 8  // with one wave per SIMD, in-wave prefetch (32 rows per stage) is the latency hiding
#endif
#ifndef ALS_SWEEP_WAVES
#define ALS_SWEEP_WAVES 4  // waves per SIMD the per-row kernel's register budget is sized for (f <= 64)
#endif
// k-steps (of 4 gathered rows) per pipeline stage (16 rows in flight per stage)
template <int VW>
struct StageGeo {
    static constexpr int GS = VW >= 8 ? ALS_STAGE_F128 : 4;
    static constexpr int NG = 16 / GS;  // stages per 64-entry chunk
};

template <int VW>
struct Stage {
    float v[StageGeo<VW>::GS][VW];
    float w[StageGeo<VW>::GS];
    float cw[StageGeo<VW>::GS];
};

// ----------------------------------------------------------------------------- loads
// Vector unit of the factor loads: VW floats for VW <= 4, 4 floats (two loads) for VW = 8.
template <int VW>
struct VecUnit {
    static constexpr int W = VW >= 4 ? 4 : VW;
};

// Elements fidx(j, 0..VW-1) of one factor row; elements >= f read as 0 (not needed when FULL,
// i.e. f == 16*VW).  VEC: `off` is the row's offset in vector units (row * ld / W) and the row is
// fetched with 8- or 16-byte loads; otherwise `off` is the row index and elements are loaded one
// by one (any f, any ld, any alignment).
template <int VW, bool VEC, bool FULL>
__device__ __forceinline__ void load_vec(const float *__restrict__ Y, const uint32_t off, const int64_t ld,
                                         const int f, const int j, float (&v)[VW]) {
    if constexpr (!VEC || VW == 1) {
        const float *rowp = Y + (int64_t)off * ld;
#pragma unroll
        for (int t = 0; t < VW; ++t) {
            const int idx = fidx<VW>(j, t);
            const bool ok = FULL || idx < f;
            const float x = rowp[ok ? idx : 0];
            v[t] = ok ? x : 0.f;
        }
    } else if constexpr (VW == 2) {
        const bool ok = FULL || 2 * j < f;
        const f32x2 x = reinterpret_cast<const f32x2 *>(Y)[off + (ok ? j : 0)];
        v[0] = ok ? x.x : 0.f;
        v[1] = ok ? x.y : 0.f;
    } else {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const bool ok = FULL || 64 * h + 4 * j < f;
            const f32x4 x = reinterpret_cast<const f32x4 *>(Y)[off + (ok ? 16 * h + j : 0)];
            v[4 * h + 0] = ok ? x.x : 0.f;
            v[4 * h + 1] = ok ? x.y : 0.f;
            v[4 * h + 2] = ok ? x.z : 0.f;
            v[4 * h + 3] = ok ? x.w : 0.f;
        }
    }
}

// ----------------------------------------------------------------------------- accumulate
// Issue the gathers of group g (16 entries) of the 64-entry chunk whose row offsets and weights
// sit one per lane in (cur_off, cur_w, cur_cw).
template <int VW, bool VEC, bool FULL>
__device__ __forceinline__ void issue(Stage<VW> &st, const int g, const uint32_t cur_off, const float cur_w,
                                      const float cur_cw, const float *__restrict__ Y, const int64_t ld, const int f,
                                      const int lane) {
    constexpr int GS = StageGeo<VW>::GS;
    const int grp = lane >> 4, j = lane & 15;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        const int src = g * (GS * 4) + s * 4 + grp;
        const uint32_t o = (uint32_t)__shfl((int)cur_off, src);
        st.w[s] = __shfl(cur_w, src);
        st.cw[s] = __shfl(cur_cw, src);
        load_vec<VW, VEC, FULL>(Y, o, ld, f, j, st.v[s]);
    }
}

template <int VW, bool WITH_B>
__device__ __forceinline__ void compute(const Stage<VW> &st, f32x4 (&acc)[Geo<VW>::NT], float (&bacc)[VW]) {
#pragma unroll
    for (int s = 0; s < StageGeo<VW>::GS; ++s) {
#pragma unroll
        for (int ta = 0; ta < VW; ++ta) {
            const float a = st.w[s] * st.v[s][ta];
#pragma unroll
            for (int tb = ta; tb < VW; ++tb) {
                const int t = tile_index(ta, tb, VW);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st.v[s][tb], acc[t], 0, 0, 0);
            }
            if constexpr (WITH_B) bacc[ta] = fmaf(st.cw[s], st.v[s][ta], bacc[ta]);
        }
    }
}

// ----------------------------------------------------------------------------- bf16 x 3 accumulate
// The rank-32 update  sum_k w_k y_k y_k^T  on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 16
// cycles per 16 x 16 x 32 block against 8 x 32 cycles of the fp32 form, and that pipe runs beside
// the VALU).  Every value x = sqrt(w_k) y_k[i] is cut into three bf16 pieces by truncation,
//     hi = x & 0xffff0000,  mid = (x - hi) & 0xffff0000,  lo = (x - hi) - mid,
// which is exact: x = hi + mid + lo (3 x 8 mantissa bits).  Of the 9 piece products the 6 of
// relative size >= 2^-16 are kept (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid); the dropped ones
// are below 2^-23 of the product, i.e. at the rounding of the fp32 result.
// No data movement is needed: lane (g, j) holds, for its 4 (or 8) factor indices, the entries
// 4q + g (q = 0..7) of a 32-entry block - exactly the 8 k-slots a lane owns in the A and in the B
// fragment (slot 8g + q <-> entry 4q + g, the same permutation for both operands).

template <int VW>
struct SplitOps {
    unsigned H[VW][4], M[VW][4], L[VW][4];   // fragment register r holds k-slots 2r (low half) and 2r + 1
};

// ----------------------------------------------------------------------------- fp16 x 2 accumulate
// The same rank-32 update with every operand value cut into TWO fp16 pieces, x s = hi + lo (hi = the value rounded to
// fp16, lo = the rounded rest: 2 x 11 mantissa bits, the representation error is below 2^-22 of the value), and the
// three piece products of relative size >= 2^-11 (hi.hi, hi.lo, lo.hi; lo.lo < 2^-22) on v_mfma_f32_16x16x32_f16 -
// half the matrix-pipe time of the six bf16 products and half the conversion work, at an error of 2^-21 per
// product instead of 2^-23.  fp16 has a narrow exponent range, so the operands are scaled by a power of two s chosen
// per call from bounds the caller has anyway: |sqrt(alpha r) y| s <= 2^15 with r <= the CSR's largest rating and
// |y| <= sqrt(max_k (Y^T Y)_kk) (als_gram_f32 leaves that bound behind its tile-order Gram).  Values far below the
// bound lose RELATIVE precision in their lo piece (fp16 subnormals: absolute error 2^-25 in scaled units) but
// stay exact to 2^-40 of the bound, which is what the sums see.  The accumulators hold s^2 times the sums and are
// multiplied by the exact 1 / s^2 when the row's matrix is complete.

// Pieces of the GS k-steps of one stage into fragment registers [QB/2, QB/2 + GS/2); also bacc.
// st.w holds sqrt(alpha r) here.
// The elementwise work runs on pairs of adjacent factor indices (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two float32 operations per instruction, bit-identical to the scalar forms).
__device__ __forceinline__ f32x2 trunc_bf16(const f32x2 x) { return f32x2{trunc_bf16(x.x), trunc_bf16(x.y)}; }
template <int VW, int MODE = ALS_MODE_BF16X3>
__device__ __forceinline__ void split_stage(const Stage<VW> &st, const int QB, SplitOps<VW> &o, float (&bacc)[VW]) {
    constexpr int GS = StageGeo<VW>::GS;
    static_assert(VW % 2 == 0, "pairs of factor indices");
    if constexpr (MODE == ALS_MODE_FP16X2) {      // st.w holds s sqrt(alpha r)
        // (no contraction: x - hi must be the float32 product minus hi in every kernel that cuts operands, or the
        // one-wave and the stream kernel would differ in the last bit)
#pragma clang fp contract(off)
#pragma unroll
        for (int p = 0; p < GS / 2; ++p) {
#pragma unroll
            for (int t = 0; t < VW; ++t) {
                const float x0 = st.w[2 * p] * st.v[2 * p][t], x1 = st.w[2 * p + 1] * st.v[2 * p + 1][t];
                bacc[t] = fmaf(st.cw[2 * p], st.v[2 * p][t], bacc[t]);
                bacc[t] = fmaf(st.cw[2 * p + 1], st.v[2 * p + 1][t], bacc[t]);
                const unsigned h = pack_f16_pinned(x0, x1);
                o.H[t][QB / 2 + p] = h;
                o.L[t][QB / 2 + p] = pack_f16(x0 - f16_lo(h), x1 - f16_hi(h));
            }
        }
        return;
    }
#ifndef ALS_SPLIT_PACKED_MIN_VW
#define ALS_SPLIT_PACKED_MIN_VW 8
#endif
    if constexpr (VW < ALS_SPLIT_PACKED_MIN_VW) {      // 3 waves per SIMD: the scalar form needs fewer registers there and is as fast
#pragma unroll
        for (int p = 0; p < GS / 2; ++p) {
#pragma unroll
            for (int t = 0; t < VW; ++t) {
                const float x0 = st.w[2 * p] * st.v[2 * p][t], x1 = st.w[2 * p + 1] * st.v[2 * p + 1][t];
                bacc[t] = fmaf(st.cw[2 * p], st.v[2 * p][t], bacc[t]);
                bacc[t] = fmaf(st.cw[2 * p + 1], st.v[2 * p + 1][t], bacc[t]);
                const float r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
                const float l0 = r0 - trunc_bf16(r0), l1 = r1 - trunc_bf16(r1);
                o.H[t][QB / 2 + p] = pack_trunc(x0, x1);
                o.M[t][QB / 2 + p] = pack_trunc(r0, r1);
                o.L[t][QB / 2 + p] = pack_trunc(l0, l1);
            }
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < GS / 2; ++p) {
#pragma unroll
        for (int t = 0; t < VW; t += 2) {
            const f32x2 v0{st.v[2 * p][t], st.v[2 * p][t + 1]}, v1{st.v[2 * p + 1][t], st.v[2 * p + 1][t + 1]};
            const f32x2 x0 = v0 * st.w[2 * p], x1 = v1 * st.w[2 * p + 1];
            f32x2 b{bacc[t], bacc[t + 1]};
            b = __builtin_elementwise_fma(f32x2{st.cw[2 * p], st.cw[2 * p]}, v0, b);
            b = __builtin_elementwise_fma(f32x2{st.cw[2 * p + 1], st.cw[2 * p + 1]}, v1, b);
            bacc[t] = b.x; bacc[t + 1] = b.y;
            const f32x2 r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
            const f32x2 l0 = r0 - trunc_bf16(r0), l1 = r1 - trunc_bf16(r1);
            o.H[t][QB / 2 + p] = pack_trunc(x0.x, x1.x); o.H[t + 1][QB / 2 + p] = pack_trunc(x0.y, x1.y);
            o.M[t][QB / 2 + p] = pack_trunc(r0.x, r1.x); o.M[t + 1][QB / 2 + p] = pack_trunc(r0.y, r1.y);
            o.L[t][QB / 2 + p] = pack_trunc(l0.x, l1.x); o.L[t + 1][QB / 2 + p] = pack_trunc(l0.y, l1.y);
        }
    }
}
template <int VW>
__device__ __forceinline__ void split_zero(const int QB, SplitOps<VW> &o) {
    constexpr int GS = StageGeo<VW>::GS;
#pragma unroll
    for (int p = 0; p < GS / 2; ++p)
#pragma unroll
        for (int t = 0; t < VW; ++t) o.H[t][QB / 2 + p] = o.M[t][QB / 2 + p] = o.L[t][QB / 2 + p] = 0u;
}
template <int VW>
__device__ __forceinline__ void split_mfma_pass(const unsigned (&A)[VW][4], const unsigned (&B)[VW][4],
                                                f32x4 (&acc)[Geo<VW>::NT]) {
#pragma unroll
    for (int ta = 0; ta < VW; ++ta)
#pragma unroll
        for (int tb = ta; tb < VW; ++tb) {
            const int t = tile_index(ta, tb, VW);
            acc[t] = mfma_bf16(A[ta], B[tb], acc[t]);
        }
}
// one 32-entry block: smallest products first; consecutive MFMAs go to different accumulators
template <int VW>
__device__ __forceinline__ void split_mfma_pass_f16(const unsigned (&A)[VW][4], const unsigned (&B)[VW][4],
                                                    f32x4 (&acc)[Geo<VW>::NT]) {
#pragma unroll
    for (int ta = 0; ta < VW; ++ta)
#pragma unroll
        for (int tb = ta; tb < VW; ++tb) {
            const int t = tile_index(ta, tb, VW);
            acc[t] = mfma_f16(A[ta], B[tb], acc[t]);
        }
}
template <int VW, int MODE = ALS_MODE_BF16X3>
__device__ __forceinline__ void split_block(const SplitOps<VW> &o, f32x4 (&acc)[Geo<VW>::NT]) {
    if constexpr (MODE == ALS_MODE_FP16X2) {
        split_mfma_pass_f16<VW>(o.L, o.H, acc);
        split_mfma_pass_f16<VW>(o.H, o.L, acc);
        split_mfma_pass_f16<VW>(o.H, o.H, acc);
        return;
    }
    split_mfma_pass<VW>(o.L, o.H, acc);
    split_mfma_pass<VW>(o.H, o.L, acc);
    split_mfma_pass<VW>(o.M, o.M, acc);
    split_mfma_pass<VW>(o.M, o.H, acc);
    split_mfma_pass<VW>(o.H, o.M, acc);
    split_mfma_pass<VW>(o.H, o.H, acc);
}

// acc += sum_{e in [s, e)} w_e y_e y_e^T, bacc += sum c_e y_e over CSR entries [s, e) (GATHER)
// with w = alpha r, c = 1 + alpha r (0 for a stored zero), or over the dense rows [s, e) of Y with
// w = 1 (!GATHER, the Gram).  s, e are wave uniform, e > s.
// MODE: ALS_MODE_FP32 (float32 fma chain), _BF16X3, _FP16X2 (operands scaled by op_s, see above).
template <int VW, bool VEC, bool FULL, bool GATHER, int MODE = ALS_MODE_FP32>
__device__ __forceinline__ void accumulate_range(f32x4 (&acc)[Geo<VW>::NT], float (&bacc)[VW],
                                                 const int32_t *__restrict__ col, const float *__restrict__ val,
                                                 const int64_t s, const int64_t e, const float *__restrict__ Y,
                                                 const int64_t ld, const int f, const float alpha, const int lane,
                                                 const float op_s = 1.f) {
    constexpr bool SPLIT = MODE != ALS_MODE_FP32;
    constexpr int GS = StageGeo<VW>::GS, NG = StageGeo<VW>::NG;
    Stage<VW> A, B;
    SplitOps<VW> ops;   // only touched when SPLIT
    uint32_t cur_off, nxt_off = 0;
    float cur_w, cur_cw, nxt_w = 0.f, nxt_cw = 0.f;
    const uint32_t ldu = VEC ? (uint32_t)(ld / VecUnit<VW>::W) : 1u;   // row pitch in vector units
    auto load_chunk = [&](const int64_t base, uint32_t &off, float &w, float &cw) {
        const int64_t idx = base + lane;
        const bool ok = idx < e;
        if constexpr (GATHER) {
            const int64_t ii = ok ? idx : s;
            const int cc = col[ii];
            const float rr = val[ii];
            const float ww = alpha * rr;
            off = (uint32_t)cc * ldu;      // a lane past the end gathers the range's first entry's row (ii = s) with weight 0:
                                           // only rows this CSR row references are ever read
            w = ok ? (SPLIT ? __builtin_sqrtf(ww) * op_s : ww) : 0.f;      // SPLIT: both operands carry sqrt(w) (op_s = 1 but for fp16 x 2)
            cw = (ok && rr != 0.f) ? 1.f + ww : 0.f;
        } else {
            off = ok ? (uint32_t)idx * ldu : 0u;
            w = ok ? 1.f : 0.f;
            cw = 0.f;
        }
    };
    // The loads have a FIXED shape: every chunk loads the next chunk's ids and every stage issues the next stage's
    // factor rows, whether or not they hold entries (past the end every lane reads entry s / row 0 with weight 0, and
    // stages without entries are not multiplied).  With the issue behind a condition the two ways to the next
    // stage's arithmetic have different loads pending, hipcc can only wait for all of them (s_waitcnt vmcnt(0)), and
    // the prefetch of the stage ahead is lost; with one shape it counts (vmcnt(GS)) and the loads overlap the arithmetic.
    load_chunk(s, cur_off, cur_w, cur_cw);
    issue<VW, VEC, FULL>(A, 0, cur_off, cur_w, cur_cw, Y, ld, f, lane);
    for (int64_t base = s; base < e; base += 64) {
        const int64_t nb = base + 64;
        const bool more = nb < e;
#ifdef ALS_CONDITIONAL_ISSUE
        if (more)
#endif
        load_chunk(nb, nxt_off, nxt_w, nxt_cw);
        const int rem = more ? 64 : (int)(e - base);
        const int ng = (rem + 4 * GS - 1) / (4 * GS);   // stages of this chunk that hold entries
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            Stage<VW> &cur = (g & 1) ? B : A;
            Stage<VW> &nxt = (g & 1) ? A : B;
            if (g + 1 < NG) {
#ifdef ALS_CONDITIONAL_ISSUE
                if (ng > g + 1)
#endif
                issue<VW, VEC, FULL>(nxt, g + 1, cur_off, cur_w, cur_cw, Y, ld, f, lane);
            } else
#ifdef ALS_CONDITIONAL_ISSUE
            if (more)
#endif
            {   // last stage of a chunk: rotate to the prefetched chunk
                cur_off = nxt_off;
                cur_w = nxt_w;
                cur_cw = nxt_cw;
                issue<VW, VEC, FULL>(nxt, 0, cur_off, cur_w, cur_cw, Y, ld, f, lane);
            }
            if constexpr (SPLIT) {
                // a 32-entry block is 8 / GS stages; its MFMAs run when its last stage has been split
                constexpr int SPB = 8 / GS;                    // stages per block (2 for GS = 4, 1 for GS = 8)
                const int QB = (g % SPB) * GS;                 // first k-slot of this stage within its block
                if (ng > g) split_stage<VW, MODE>(cur, QB, ops, bacc);
                else if (ng > g - (g % SPB)) split_zero<VW>(QB, ops);   // block started, this stage is empty
                if ((g % SPB) == SPB - 1 && ng > g - (g % SPB)) split_block<VW, MODE>(ops, acc);
            } else {
                if (ng > g) compute<VW, GATHER>(cur, acc, bacc);
            }
        }
    }
}

template <int VW>
__device__ __forceinline__ void reduce_b(float (&bacc)[VW]) {
    // the 4 lane groups hold partial sums of the same elements: fixed butterfly order
#pragma unroll
    for (int t = 0; t < VW; ++t) {
        bacc[t] += __shfl_xor(bacc[t], 16);
        bacc[t] += __shfl_xor(bacc[t], 32);
    }
}

// ----------------------------------------------------------------------------- CG on the formed matrix
// y = T v for the symmetric block matrix in tile storage (tiles ta <= tb; block (tb, ta) is the
// transpose of tile (ta, tb)).  v and y are column indexed per block.  The per-lane products of
// every block row are summed first and reduced across lanes once per block.
template <int VW>
__device__ __forceinline__ void sym_tiles_matvec(const f32x4 (&T)[Geo<VW>::NT], const float (&v)[VW], float (&y)[VW],
                                                 const int lane) {
    // register pairs: two multiply-adds per v_pk_fma_f32
    f32x4 vrow[VW];
    f32x2 olo[VW], ohi[VW], ocol[VW];
#pragma unroll
    for (int t = 0; t < VW; ++t) {
        vrow[t] = vec_col_to_row(v[t], lane);
        olo[t] = ohi[t] = ocol[t] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int ib = 0; ib < VW; ++ib)
#pragma unroll
        for (int jb = ib; jb < VW; ++jb) {
            const f32x4 &t = T[tile_index(ib, jb, VW)];
            const f32x2 vj{v[jb], v[jb]};
            olo[ib] = __builtin_elementwise_fma(t.lo, vj, olo[ib]);                  // (T v)[4g + r], before the lane sum
            ohi[ib] = __builtin_elementwise_fma(t.hi, vj, ohi[ib]);
            if (jb > ib) {                                                           // (T^T v)[c], before the group sum
                ocol[jb] = __builtin_elementwise_fma(t.lo, vrow[ib].lo, ocol[jb]);
                ocol[jb] = __builtin_elementwise_fma(t.hi, vrow[ib].hi, ocol[jb]);
            }
        }
#pragma unroll
    for (int ib = 0; ib < VW; ++ib) {
        const f32x4 srow{row_sum16(olo[ib].x), row_sum16(olo[ib].y), row_sum16(ohi[ib].x), row_sum16(ohi[ib].y)};
        y[ib] = vec_row_to_col(srow, lane) + grp_sum4(ocol[ib].x + ocol[ib].y);
    }
}
template <int VW>
__device__ __forceinline__ float col_dot(const float (&a)[VW], const float (&b)[VW]) {
    float d = a[0] * b[0];
#pragma unroll
    for (int t = 1; t < VW; ++t) d = fmaf(a[t], b[t], d);
    return row_sum16(d);      // the same bits in every lane (vectors are replicated over the lane groups)
}
// `steps` conjugate-gradient steps on T x = b from the start value x (column indexed, updated in
// place).  T is not modified.  Returns false if a step met p^T T p <= 0 with a non-zero residual.
// A step that cannot be taken (zero residual, or not positive definite) ends the iteration.
template <int VW>
__device__ __forceinline__ bool tiles_cg(const f32x4 (&T)[Geo<VW>::NT], const float (&b)[VW], float (&x)[VW],
                                         const int steps, const int lane) {
    float r[VW], pv[VW], q[VW];
    sym_tiles_matvec<VW>(T, x, q, lane);
#pragma unroll
    for (int t = 0; t < VW; ++t) {
        r[t] = b[t] - q[t];
        pv[t] = r[t];
    }
    float rs = col_dot<VW>(r, r);
    bool spd = true;
    for (int s = 0; s < steps; ++s) {
        sym_tiles_matvec<VW>(T, pv, q, lane);
        const float pAp = col_dot<VW>(pv, q);
        const bool ok = rs > 0.f && pAp > 0.f;
        spd = spd && !(rs > 0.f && !(pAp > 0.f));
        const float a = ok ? rs / pAp : 0.f;
#pragma unroll
        for (int t = 0; t < VW; ++t) {
            x[t] = fmaf(a, pv[t], x[t]);
            r[t] = fmaf(-a, q[t], r[t]);
        }
        const float rr = col_dot<VW>(r, r);
        const float beta = ok ? rr / rs : 0.f;
#pragma unroll
        for (int t = 0; t < VW; ++t) pv[t] = fmaf(beta, pv[t], r[t]);
        rs = ok ? rr : 0.f;
    }
    return spd;
}

// tiles [T0, T1) of the tile-order Gram (a lane's 16 bytes of tile t at gtiles[(t * 64 + lane) * 4]) into g
template <int T0, int T1>
__device__ __forceinline__ void load_gram_tiles(f32x4 (&g)[T1 - T0], const float *__restrict__ gtiles, const int lane) {
#pragma unroll
    for (int t = T0; t < T1; ++t) g[t - T0] = reinterpret_cast<const f32x4 *>(gtiles)[t * 64 + lane];
}
// acc += gram_tiles, GB tiles per batch of loads (all NT at once would need as many registers as acc)
template <int VW, int GB>
__device__ __forceinline__ void add_gram(f32x4 (&acc)[Geo<VW>::NT], const float *__restrict__ gtiles, const int lane) {
    constexpr int NT = Geo<VW>::NT;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += GB) {
#pragma unroll
        for (int t = t0; t < (t0 + GB < NT ? t0 + GB : NT); ++t)
            acc[t] += reinterpret_cast<const f32x4 *>(gtiles)[t * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);      // keeps the scheduler from hoisting every load to the top
    }
}
template <int VW>
__device__ __forceinline__ void load_row_values(float (&xold)[VW], const float *__restrict__ xrow, const int f, const int lane) {
    const int j = lane & 15;
#pragma unroll
    for (int t = 0; t < VW; ++t) {
        const int idx = fidx<VW>(j, t);
        xold[t] = idx < f ? xrow[idx] : 0.f;
    }
}
// T x = b for one row whose matrix T = acc + Gram is complete; xold: the row's current value.  Writes x to xrow
// and returns sum |x_old - x_new| (all lanes).  cg_steps == 0: blocked Cholesky; > 0: that many CG steps from xold.
template <int VW>
__device__ __forceinline__ float solve_row_formed(f32x4 (&acc)[Geo<VW>::NT], float (&bacc)[VW], const float (&xold)[VW],
                                                  const int f, float *__restrict__ xrow, int32_t *info,
                                                  const int cg_steps, const int lane) {
    const int j = lane & 15;
    float x[VW];
    bool ok;
    if (cg_steps > 0) {
        // b = 0: the row holds nothing but stored zeros, which are dropped (ALS.py:79) - its solution is 0, and
        // it gets it rather than cg_steps steps towards it (the exact solve returns 0 by itself)
        bool has_b = false;
#pragma unroll
        for (int t = 0; t < VW; ++t) has_b = has_b || bacc[t] != 0.f;
        has_b = __any(has_b);
#pragma unroll
        for (int t = 0; t < VW; ++t) x[t] = has_b ? xold[t] : 0.f;
        ok = has_b ? tiles_cg<VW>(acc, bacc, x, cg_steps, lane) : true;
    } else {
#ifndef ALS_SOLVE_BF16_MIN_VW
#define ALS_SOLVE_BF16_MIN_VW 8
#endif
        if constexpr (VW >= ALS_SOLVE_BF16_MIN_VW) ok = blocked_solve_bf16<VW>(acc, bacc, x, lane);
        else ok = blocked_solve<VW>(acc, bacc, x, lane);
    }
    float d = 0.f;
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < VW; ++t) {
            const int idx = fidx<VW>(j, t);
            if (idx < f) {
                d += fabsf(xold[t] - x[t]);
                xrow[idx] = x[t];
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
    if (!ok && lane == 0 && info) atomicAdd(info, 1);
    return d;
}
// (acc + gram_tiles) x = b for one row; writes x to xrow and returns sum |x_old - x_new| (all lanes).
template <int VW>
__device__ __forceinline__ float solve_row_blocked(f32x4 (&acc)[Geo<VW>::NT], float (&bacc)[VW],
                                                   const float *__restrict__ gtiles, const int f,
                                                   float *__restrict__ xrow, int32_t *info, const int cg_steps,
                                                   const int lane) {
    add_gram<VW, 4>(acc, gtiles, lane);
    float xold[VW];
    load_row_values<VW>(xold, xrow, f, lane);
    return solve_row_formed<VW>(acc, bacc, xold, f, xrow, info, cg_steps, lane);
}

// ----------------------------------------------------------------------------- kernels
struct SweepParams {
    const int64_t *rowptr;
    const int32_t *col;
    const float *val;
    const int64_t *row_ids;
    const als_item_t *items;
    const als_long_row_t *long_rows;
    const float *Y;
    const float *gtiles;
    float *X;
    float *partials;
    float *rowdiff;
    int32_t *info;
    int32_t *stream_next;   // two counters of the stream kernel's launches (chunk items, whole rows): next unclaimed item
    int64_t ldy, ldx;
    int32_t f, chunk_nnz;
    float alpha;
    int32_t cg_steps;   // 0: exact solve
    float lamb;         // only the f > 128 kernels read it (for f <= 128 the Gram tiles carry it)
    float rmax;         // fp16 x 2 accumulate: an upper bound of the ratings
};
// the scale of the fp16 x 2 operands of a sweep: |sqrt(alpha r) y| <= sqrt(alpha rmax) * (bound of |y| behind the Gram tiles)
template <int VW>
__device__ __forceinline__ OpScale sweep_op_scale(const SweepParams &p) {
    return op_scale(__builtin_sqrtf(p.alpha * p.rmax) * p.gtiles[Geo<VW>::TQ * 64]);
}

#ifndef ALS_SWEEP_WAVES_SPLIT
#define ALS_SWEEP_WAVES_SPLIT 3  // f <= 64 with the bf16 x 3 accumulate: 48 more registers for the fragments
#endif
template <int VW, bool VEC, bool FULL, int MODE>
__global__ __launch_bounds__(64, (VW >= 8 ? ALS_SWEEP_WAVES_F128 : MODE != ALS_MODE_FP32 ? ALS_SWEEP_WAVES_SPLIT : ALS_SWEEP_WAVES))
void sweep_items_kernel(const SweepParams p) {
    using G = Geo<VW>;
    const int lane = threadIdx.x;
    const als_item_t it = p.items[blockIdx.x];
    const int row = it.row;
    const int64_t s = it.begin, e = it.begin + it.count;
    f32x4 acc[G::NT];
    float bacc[VW];
#pragma unroll
    for (int t = 0; t < G::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < VW; ++t) bacc[t] = 0.f;

#ifndef ALS_ABLATE_NO_ACCUMULATE
    if (e > s) {
        if constexpr (MODE == ALS_MODE_FP16X2) {
            const OpScale sc = sweep_op_scale<VW>(p);
            accumulate_range<VW, VEC, FULL, true, MODE>(acc, bacc, p.col, p.val, s, e, p.Y, p.ldy, p.f, p.alpha, lane, sc.s);
#pragma unroll
            for (int t = 0; t < G::NT; ++t) acc[t] *= sc.inv2;
        } else {
            accumulate_range<VW, VEC, FULL, true, MODE>(acc, bacc, p.col, p.val, s, e, p.Y, p.ldy, p.f, p.alpha, lane);
        }
        reduce_b<VW>(bacc);
    }
#endif
    if (it.chunk >= 0) {
        float *P = p.partials + (int64_t)it.slot * (G::PQ * 64);
#pragma unroll
        for (int t = 0; t < G::NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(t * 4 + r) * 64 + lane] = acc[t][r];
#pragma unroll
        for (int t = 0; t < VW; ++t) P[(G::TQ + t) * 64 + lane] = bacc[t];
        return;
    }
    const int64_t xr = p.row_ids ? p.row_ids[row] : (int64_t)row;
    float *xrow = p.X + xr * p.ldx;
    float d;
#ifdef ALS_ABLATE_NO_SOLVE
    if (e > s) {  // timing-only build: keeps the accumulators live, writes garbage
        d = 0.f;
        float sacc = 0.f;
        for (int t = 0; t < G::NT; ++t) sacc += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        if (lane < p.f) xrow[lane] = sacc + bacc[0];
    } else {
#else
    if (e > s) {
        d = solve_row_blocked<VW>(acc, bacc, p.gtiles, p.f, xrow, p.info, p.cg_steps, lane);
    } else {
#endif  // no entries: b = 0, so x = 0 exactly
        d = 0.f;
        for (int k = lane; k < p.f; k += 64) {
            d += fabsf(xrow[k]);
            xrow[k] = 0.f;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
    }
    if (lane == 0) p.rowdiff[row] = d;
}

// Adds the chunk partials of every long row into the row's first slot, element by element in
// chunk order (so the sum is the same left-to-right chain for any launch geometry).
// grid = (n_long, ceil(elems / 256)).
__global__ __launch_bounds__(256) void sum_slots_kernel(float *__restrict__ partials,
                                                       const als_long_row_t *__restrict__ long_rows,
                                                       const int elems) {
    const als_long_row_t lr = long_rows[blockIdx.x];
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= elems) return;
    float *P = partials + (int64_t)lr.first_slot * elems + e;
    float acc = P[0];
    int c = 1;
    for (; c + 7 < lr.n_chunks; c += 8) {      // 8 loads in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = P[(int64_t)(c + q) * elems];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; c < lr.n_chunks; ++c) acc += P[(int64_t)c * elems];
    P[0] = acc;
}

template <int VW>
__global__ __launch_bounds__(64) void sweep_long_rows_kernel(const SweepParams p) {
    using G = Geo<VW>;
    const int lane = threadIdx.x;
    const als_long_row_t lr = p.long_rows[blockIdx.x];
    f32x4 acc[G::NT];
    float bacc[VW];
    const float *P = p.partials + (int64_t)lr.first_slot * (G::PQ * 64);
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = P[(t * 4 + r) * 64 + lane];
#pragma unroll
    for (int t = 0; t < VW; ++t) bacc[t] = P[(G::TQ + t) * 64 + lane];
    const int row = lr.row;
    const int64_t xr = p.row_ids ? p.row_ids[row] : (int64_t)row;
    const float d = solve_row_blocked<VW>(acc, bacc, p.gtiles, p.f, p.X + xr * p.ldx, p.info, p.cg_steps, lane);
    if (lane == 0) p.rowdiff[row] = d;
}

// Gram, level 1: one wave per ALS_GRAM_CHUNK_ROWS rows of Y[row0 .. n_end).
template <int VW, bool VEC, bool FULL>
__global__ __launch_bounds__(64) void gram_chunk_kernel(const float *__restrict__ Y, const int64_t row0,
                                                       const int64_t n_end, const int64_t ld, const int f,
                                                       float *__restrict__ out) {
    using G = Geo<VW>;
    const int lane = threadIdx.x;
    const int64_t s = row0 + (int64_t)blockIdx.x * ALS_GRAM_CHUNK_ROWS;
    const int64_t e = min(n_end, s + (int64_t)ALS_GRAM_CHUNK_ROWS);
    f32x4 acc[G::NT];
    float bacc[VW];
#pragma unroll
    for (int t = 0; t < G::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < VW; ++t) bacc[t] = 0.f;
#ifndef ALS_GRAM_SPLIT_MIN_VW
#define ALS_GRAM_SPLIT_MIN_VW 4      // f > 32: the bf16 x 3 form of the accumulate (weights 1), as in the sweeps
#endif
    if (e > s) accumulate_range<VW, VEC, FULL, false, (VW >= ALS_GRAM_SPLIT_MIN_VW ? ALS_MODE_BF16X3 : ALS_MODE_FP32)>(acc, bacc, nullptr, nullptr, s, e, Y, ld, f, 1.0f, lane);
    float *P = out + (int64_t)blockIdx.x * (G::TQ * 64);
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(t * 4 + r) * 64 + lane] = acc[t][r];
}

// out[g][e] = sum of in[g*group .. g*group+group)[e], e < elems, as a fixed two-level tree: the
// group is cut into SUBS runs of group/SUBS consecutive partials, each run is summed in order by one
// thread (threadIdx.y), the run sums are added in run order.  blockDim = (64, SUBS).
constexpr int SUM_SUBS = 8;
__global__ __launch_bounds__(64 * SUM_SUBS) void sum_groups_kernel(const float *__restrict__ in, const int64_t n_in,
                                                                   const int group, const int elems,
                                                                   float *__restrict__ out) {
    __shared__ float part[SUM_SUBS][64];
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int64_t g = blockIdx.y;
    const int run = group / SUM_SUBS;
    const int64_t first = g * group + (int64_t)threadIdx.y * run;
    const int64_t last = min(n_in, first + run);
    float acc = 0.f;
    if (e < elems && first < last) {
        acc = in[first * elems + e];
        int64_t i = first + 1;
        for (; i + 7 < last; i += 8) {          // 8 loads in flight, added in order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = in[(i + q) * elems + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; i < last; ++i) acc += in[i * elems + e];
    }
    part[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && e < elems) {
        float t = part[0][threadIdx.x];
        const int64_t n_here = min(n_in - g * group, (int64_t)group);      // partials in this group (>= 1)
        for (int k = 1; k < SUM_SUBS; ++k)
            if ((int64_t)k * run < n_here) t += part[k][threadIdx.x];
        out[g * elems + e] = t;
    }
}

// Sums the second-level partials in order, then writes the row-major Gram and the tile-order
// copy with lamb on the diagonal.
template <int VW>
__global__ void gram_finalize_kernel(const float *__restrict__ super, const int64_t n_super, const int f,
                                     const float lamb, float *__restrict__ gram_out,
                                     float *__restrict__ tiles_out) {
    using G = Geo<VW>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G::TQ * 64) return;
    float v = super[e];
    for (int64_t i = 1; i < n_super; ++i) v += super[i * (G::TQ * 64) + e];
    const int lane = e & 63, q = e >> 6, r = q & 3, t = q >> 2;
    int ta = 0, tb = 0;
    for (int a = 0; a < VW; ++a)
        for (int b = a; b < VW; ++b)
            if (tile_index(a, b, VW) == t) {
                ta = a;
                tb = b;
            }
    const int i = (lane >> 4) * 4 + r, j = lane & 15;
    const int ai = fidx<VW>(i, ta), bi = fidx<VW>(j, tb);
    // the tile-order copy holds a lane's 4 values of a tile side by side: one 16-byte load per tile in the sweeps
    tiles_out[(t * 64 + lane) * 4 + r] = (ai == bi) ? (ai < f ? v + lamb : 1.0f) : v;   // padded indices: identity rows
    if (ai < f && bi < f && (ta != tb || i >= j)) {
        gram_out[ai * f + bi] = v;
        gram_out[bi * f + ai] = v;
    }
}

// bound_out[0] = sqrt(max_k (Y^T Y)_kk) >= every |y_ik|: the operand bound of the fp16 x 2 accumulate, kept behind the
// tile-order Gram.  One wave.
__global__ __launch_bounds__(64) void gram_bound_kernel(const float *__restrict__ gram, const int f, float *__restrict__ bound_out) {
    float m = 0.f;
    for (int k = threadIdx.x; k < f; k += 64) m = fmaxf(m, gram[(int64_t)k * f + k]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    bound_out[threadIdx.x] = threadIdx.x == 0 ? __builtin_sqrtf(m) : 0.f;      // the whole 256-byte tail is defined
}

// level 1: block b sums rowdiff[b*8192 .. ) into a double; level 2 (single block) sums those.
constexpr int DIFF_BLOCK_ROWS = 8192;
template <typename T>
__global__ __launch_bounds__(256) void sum_to_double_kernel(const T *__restrict__ in, const int64_t n,
                                                           const int per_block, double *__restrict__ out) {
    __shared__ double sh[256];
    const int64_t first = (int64_t)blockIdx.x * per_block;
    const int64_t last = min(n, first + per_block);
    double acc = 0.0;
    int64_t i = first + threadIdx.x;
    for (; i + 7 * 256 < last; i += 8 * 256) {      // 8 loads in flight, added in order
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = in[i + q * 256];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (double)v[q];
    }
    for (; i < last; i += 256) acc += (double)in[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

#include "als_wide_row.inc"
#ifndef ALS_STREAM_CHUNKS_MIN_PER_WAVE
#define ALS_STREAM_CHUNKS_MIN_PER_WAVE 4      // chunk items per resident wave from which they go to the stream kernel
#endif
#include "als_row_stream.inc"

// ----------------------------------------------------------------------------- host helpers
inline int vw_for(int f) { return f <= 16 ? 1 : f <= 32 ? 2 : f <= 64 ? 4 : 8; }
inline int tq_for(int f) {
    const int vw = vw_for(f);
    return vw * (vw + 1) / 2 * 4;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Vector loads need whole vectors inside a row, an aligned base and row offsets (in vector
// units) that fit 32 bits; otherwise the element-wise path is used.
inline bool vec_ok(const float *p, int64_t ld, int f, int64_t n_rows) {
    const int vw = vw_for(f);
    if (vw == 1) return false;
    const int w = vw >= 4 ? 4 : 2;
    return (f % w == 0) && (ld % w == 0) && ((uintptr_t)p % (4 * w) == 0) && (n_rows * (ld / w) < 0xffffffffLL);
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <int VW>
static int32_t gram_partials_impl(const float *Y, int64_t row_begin, int64_t row_end, int32_t f, int64_t ldy,
                                  float *super_out, void *workspace, hipStream_t st) {
    using G = Geo<VW>;
    const int64_t n = row_end - row_begin;
    const int64_t n_chunks = ceil_div(n, ALS_GRAM_CHUNK_ROWS);
    const int64_t n_super = ceil_div(n, ALS_GRAM_SUPER_ROWS);
    float *chunks = (float *)workspace;
    const dim3 grid((unsigned)n_chunks), block(64);
    if (vec_ok(Y, ldy, f, row_end) && f == G::FPAD)
        hipLaunchKernelGGL((gram_chunk_kernel<VW, true, true>), grid, block, 0, st, Y, row_begin, row_end, ldy, f, chunks);
    else if (vec_ok(Y, ldy, f, row_end))
        hipLaunchKernelGGL((gram_chunk_kernel<VW, true, false>), grid, block, 0, st, Y, row_begin, row_end, ldy, f, chunks);
    else
        hipLaunchKernelGGL((gram_chunk_kernel<VW, false, false>), grid, block, 0, st, Y, row_begin, row_end, ldy, f, chunks);
    ALS_HIP_CHECK(hipGetLastError());
    const int elems = G::TQ * 64;
    static_assert((ALS_GRAM_SUPER_ROWS / ALS_GRAM_CHUNK_ROWS) % SUM_SUBS == 0, "runs must tile a group");
    hipLaunchKernelGGL(sum_groups_kernel, dim3((elems + 63) / 64, (unsigned)n_super), dim3(64, SUM_SUBS), 0, st, chunks,
                       n_chunks, ALS_GRAM_SUPER_ROWS / ALS_GRAM_CHUNK_ROWS, elems, super_out);
    ALS_HIP_CHECK(hipGetLastError());
    return ALS_OK;
}

// Per-call options with the defaults filled in (opts may be NULL).
struct SweepOpts {
    bool split;                 // a split accumulate on the bf16 / fp16 matrix pipe (where the kernel has it), else float32
    bool f16;                   // ... the fp16 x 2 form (needs rmax), else bf16 x 3
    bool f16_auto;              // f16 was the library's choice (ALS_ACCUMULATE_DEFAULT), not the caller's
    float rmax;                 // upper bound of the ratings, 0: unknown
    int row_kernel;             // ALS_ROW_KERNEL_*
    hipEvent_t tap0, tap1;
    const float *whitened;      // table of als_whiten_f32 for the fixed factor, or null: no short-row form
    int64_t ldw;
    const int64_t *bounds;      // host: als_plan_short_bounds
    int reserve_cus;            // compute units the stream kernel leaves to other streams
};
inline int32_t read_opts(const als_sweep_opts_t *o, SweepOpts &out) {
    out.split = true; out.f16 = false; out.f16_auto = false; out.rmax = 0.f; out.row_kernel = ALS_ROW_KERNEL_AUTO; out.tap0 = out.tap1 = nullptr;
    out.whitened = nullptr; out.ldw = 0; out.bounds = nullptr; out.reserve_cus = 0;
    if (!o) return ALS_OK;
    if (o->reserve_cus < 0) return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: reserve_cus must be >= 0");
    out.reserve_cus = o->reserve_cus;
    if (o->accumulate != ALS_ACCUMULATE_DEFAULT && o->accumulate != ALS_ACCUMULATE_FP32 &&
        o->accumulate != ALS_ACCUMULATE_BF16X3 && o->accumulate != ALS_ACCUMULATE_FP16X2)
        return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: unknown accumulate mode %d", o->accumulate);
    if (!(o->max_rating >= 0.f)) return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: max_rating must be >= 0 (0: unknown)");
    if (o->accumulate == ALS_ACCUMULATE_FP16X2 && !(o->max_rating > 0.f))
        return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: the fp16x2 accumulate needs max_rating, an upper bound of the ratings");
    out.rmax = o->max_rating;
    // default: fp16 x 2 when the caller gave the bound it needs, else bf16 x 3
    out.f16 = o->accumulate == ALS_ACCUMULATE_FP16X2 || (o->accumulate == ALS_ACCUMULATE_DEFAULT && o->max_rating > 0.f);
    out.f16_auto = o->accumulate == ALS_ACCUMULATE_DEFAULT;
    if (o->row_kernel != ALS_ROW_KERNEL_AUTO && o->row_kernel != ALS_ROW_KERNEL_WAVE && o->row_kernel != ALS_ROW_KERNEL_STREAM)
        return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: unknown row_kernel %d", o->row_kernel);
    if ((o->tap_begin == nullptr) != (o->tap_end == nullptr))
        return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: pass both tap events or neither");
    out.split = o->accumulate != ALS_ACCUMULATE_FP32;
    out.row_kernel = o->row_kernel;
    out.tap0 = (hipEvent_t)o->tap_begin;
    out.tap1 = (hipEvent_t)o->tap_end;
    if (o->whitened) {
        if (!o->short_bounds) return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: whitened without short_bounds");
        if (o->ld_whitened % 4 || (uintptr_t)o->whitened % 16)
            return fail(ALS_ERR_INVALID_ARG, "als_sweep_opts_t: the whitened table needs 16-byte aligned rows");
        out.whitened = o->whitened; out.ldw = o->ld_whitened; out.bounds = o->short_bounds;
    }
    return ALS_OK;
}

template <int VW>
static int32_t half_sweep_impl(const SweepParams &p, int64_t n_rows, int64_t n_items, int64_t n_long,
                               double *absdiff_out, double *diff_partials, bool vec, const SweepOpts &o, hipStream_t st) {
    if (o.tap0) ALS_HIP_CHECK(hipEventRecord(o.tap0, st));
    // Short rows (at most 16 nb_max entries) in their entry-space form (als_short_rows.hip) when the caller handed in
    // the whitened table of the fixed factor: the plan's whole rows are sorted by entry count, so they are the tail
    // [bounds[nb_max], n_items) of the item list, and the kernels below stop in front of it.  Exact solve only.
    const int64_t n_items_all = n_items;
    int nb_max = 0;
    if (o.whitened && p.cg_steps == 0) {
        nb_max = als_short_rows_max_groups(p.f);
        if (nb_max == 0) return fail(ALS_ERR_UNSUPPORTED, "the short-row form exists for 32 < f <= 128, f a multiple of 4 (f=%d)", p.f);
        const int64_t *b = o.bounds;
        const int64_t n_chunk_items = n_items - (n_rows - n_long);
        for (int j = 0; j <= ALS_SHORT_MAX_GROUPS; ++j)
            if (b[j] < n_chunk_items || b[j] > n_items || (j > 0 && b[j] > b[j - 1]))
                return fail(ALS_ERR_INVALID_ARG, "short_bounds do not belong to this plan");
        if (b[ALS_SHORT_MAX_GROUPS + 1] != n_items) return fail(ALS_ERR_INVALID_ARG, "short_bounds do not belong to this plan");
        if (o.ldw < 2 * (int64_t)p.f) return fail(ALS_ERR_INVALID_ARG, "ld_whitened < 2 f");
        n_items = b[nb_max];
    }
    const int64_t n_short = n_items_all - n_items;
    // 64 < f <= 128 with 16-byte aligned rows and the bf16 x 3 accumulate: the whole rows can go to the persistent
    // stream kernel (als_row_stream.inc)
    int rk = ALS_ROW_KERNEL_WAVE;
    if constexpr (VW == 8) {
        const bool can = vec && o.split;
        if (o.row_kernel == ALS_ROW_KERNEL_STREAM && !can)
            return fail(ALS_ERR_UNSUPPORTED, "row_kernel STREAM needs 64 < f <= 128 with 16-byte aligned factor rows and the "
                        "bf16x3 accumulate");
        if (can && o.row_kernel != ALS_ROW_KERNEL_AUTO) rk = o.row_kernel;
        // AUTO: the stream kernel once every resident wave (4 per CU) has a whole row of its own; both kernels
        // return the same bits, so the choice never shows in the results
        if (can && o.row_kernel == ALS_ROW_KERNEL_AUTO) {
            int dev = 0, cus = 0;
            ALS_HIP_CHECK(hipGetDevice(&dev));
            ALS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            if (n_rows - n_long - n_short >= 4 * (int64_t)std::max(cus, 1)) rk = ALS_ROW_KERNEL_STREAM;
        }
    } else if (o.row_kernel == ALS_ROW_KERNEL_STREAM) {
        return fail(ALS_ERR_UNSUPPORTED, "row_kernel STREAM exists for 64 < f <= 128 only (f=%d)", p.f);
    }
    if (n_items > 0 && rk == ALS_ROW_KERNEL_STREAM) {
        if constexpr (VW == 8) {
            // the plan lists the chunk items of long rows first (one per partial slot), then the whole rows
            const int64_t n_whole = n_rows - n_long - n_short, n_chunk_items = n_items - n_whole;
            int dev = 0, cus = 0;
            ALS_HIP_CHECK(hipGetDevice(&dev));
            ALS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            cus = std::max(cus, 1);
            // resident waves: one per SIMD - a workgroup of four per compute unit -, but for the compute units the caller
            // wants left to other streams
            const int64_t waves = STREAM_WG_WAVES * (int64_t)std::max(cus - o.reserve_cus, 1);
            // The chunk items of long rows: as streams too once every resident wave gets several of them (their static
            // shares are then even; with few, the one-wave kernel's workgroups, handed out as SIMDs fall free, balance
            // better).  Same partial sums bit for bit either way.
            // the items of a launch are handed out through a counter that starts behind every wave's two fixed batches
            // (batches of STREAM_BATCH items; smaller ones when a wave would not get two of those)
            auto batch_for = [&](const int64_t n, const int64_t n_waves) {
                return (int)std::max<int64_t>(1, std::min<int64_t>(STREAM_BATCH, n / (2 * n_waves)));
            };
            auto start_counter = [&](int32_t *c, const int64_t n_waves, const int batch) {
                return hipMemsetD32Async((hipDeviceptr_t)c, (int)(2 * n_waves * batch), 1, st);
            };
            if (n_items > 0x7fffffffLL - 8 * waves * STREAM_BATCH)
                return fail(ALS_ERR_UNSUPPORTED, "too many work items for one launch of the stream kernel");
            if (n_chunk_items >= ALS_STREAM_CHUNKS_MIN_PER_WAVE * 4 * (int64_t)cus) {
                const dim3 grid((unsigned)(waves / STREAM_WG_WAVES)), block(64 * STREAM_WG_WAVES);
                const int batch = batch_for(n_chunk_items, waves);
                ALS_HIP_CHECK(start_counter(p.stream_next, waves, batch));
                if (o.f16) {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_stream_kernel<true, true, true>), grid, block, 0, st, p, (int64_t)0, n_chunk_items, batch);
                    else hipLaunchKernelGGL((sweep_stream_kernel<false, true, true>), grid, block, 0, st, p, (int64_t)0, n_chunk_items, batch);
                } else {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_stream_kernel<true, true, false>), grid, block, 0, st, p, (int64_t)0, n_chunk_items, batch);
                    else hipLaunchKernelGGL((sweep_stream_kernel<false, true, false>), grid, block, 0, st, p, (int64_t)0, n_chunk_items, batch);
                }
                ALS_HIP_CHECK(hipGetLastError());
            } else if (n_chunk_items > 0) {
                const dim3 grid((unsigned)n_chunk_items), block(64);
                if (o.f16) {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_items_kernel<VW, true, true, ALS_MODE_FP16X2>), grid, block, 0, st, p);
                    else hipLaunchKernelGGL((sweep_items_kernel<VW, true, false, ALS_MODE_FP16X2>), grid, block, 0, st, p);
                } else {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_items_kernel<VW, true, true, ALS_MODE_BF16X3>), grid, block, 0, st, p);
                    else hipLaunchKernelGGL((sweep_items_kernel<VW, true, false, ALS_MODE_BF16X3>), grid, block, 0, st, p);
                }
                ALS_HIP_CHECK(hipGetLastError());
            }
            if (n_whole > 0) {
                const int64_t groups = ceil_div(std::min<int64_t>(n_whole, waves), STREAM_WG_WAVES);
                const dim3 grid((unsigned)groups), block(64 * STREAM_WG_WAVES);
                const int batch = batch_for(n_whole, groups * STREAM_WG_WAVES);
                ALS_HIP_CHECK(start_counter(p.stream_next + 1, groups * STREAM_WG_WAVES, batch));
                if (o.f16) {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_stream_kernel<true, false, true>), grid, block, 0, st, p, n_chunk_items, n_whole, batch);
                    else hipLaunchKernelGGL((sweep_stream_kernel<false, false, true>), grid, block, 0, st, p, n_chunk_items, n_whole, batch);
                } else {
                    if (p.f == 128) hipLaunchKernelGGL((sweep_stream_kernel<true, false, false>), grid, block, 0, st, p, n_chunk_items, n_whole, batch);
                    else hipLaunchKernelGGL((sweep_stream_kernel<false, false, false>), grid, block, 0, st, p, n_chunk_items, n_whole, batch);
                }
                ALS_HIP_CHECK(hipGetLastError());
            }
        }
    } else if (n_items > 0) {
        const dim3 grid((unsigned)n_items), block(64);
        constexpr bool CAN_SPLIT = VW >= 4;
        constexpr int M3 = CAN_SPLIT ? ALS_MODE_BF16X3 : ALS_MODE_FP32, M2 = CAN_SPLIT ? ALS_MODE_FP16X2 : ALS_MODE_FP32;
        const int mode = !(CAN_SPLIT && o.split) ? ALS_MODE_FP32 : o.f16 ? ALS_MODE_FP16X2 : ALS_MODE_BF16X3;
#define ALS_LAUNCH_ITEMS(VEC_, FULL_)                                                                                      \
        if (mode == ALS_MODE_FP16X2) hipLaunchKernelGGL((sweep_items_kernel<VW, VEC_, FULL_, M2>), grid, block, 0, st, p);       \
        else if (mode == ALS_MODE_BF16X3) hipLaunchKernelGGL((sweep_items_kernel<VW, VEC_, FULL_, M3>), grid, block, 0, st, p);  \
        else hipLaunchKernelGGL((sweep_items_kernel<VW, VEC_, FULL_, ALS_MODE_FP32>), grid, block, 0, st, p);
        if (vec && p.f == Geo<VW>::FPAD) { ALS_LAUNCH_ITEMS(true, true) }
        else if (vec) { ALS_LAUNCH_ITEMS(true, false) }
        else { ALS_LAUNCH_ITEMS(false, false) }
#undef ALS_LAUNCH_ITEMS
        ALS_HIP_CHECK(hipGetLastError());
    }
    if (n_long > 0) {
        const int elems = Geo<VW>::PQ * 64;
        hipLaunchKernelGGL(sum_slots_kernel, dim3((unsigned)n_long, (unsigned)((elems + 255) / 256)), dim3(256), 0, st,
                           p.partials, p.long_rows, elems);
        ALS_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(sweep_long_rows_kernel<VW>, dim3((unsigned)n_long), dim3(64), 0, st, p);
        ALS_HIP_CHECK(hipGetLastError());
    }
    if (n_short > 0) {
        const int32_t rc = als_short_rows_launch_(p.col, p.val, p.row_ids, p.items, o.bounds, nb_max, o.whitened, o.ldw, p.f,
                                                  p.alpha, (o.split && o.f16) ? o.rmax : 0.f, p.X, p.ldx, p.rowdiff, p.info, st);
        if (rc) return rc;
    }
    if (o.tap1) ALS_HIP_CHECK(hipEventRecord(o.tap1, st));      // every per-row kernel of the half-sweep lies between the taps
    if (absdiff_out) return als_sum_rowdiff_(p.rowdiff, n_rows, diff_partials, absdiff_out, st);
    return ALS_OK;
}

}  // namespace

// shared with als_cg.hip: *out = sum of rowdiff[0 .. n_rows) in double, fixed two-level order
int32_t als_sum_rowdiff_(const float *rowdiff, int64_t n_rows, double *partials, double *out, hipStream_t st) {
    const int64_t nb = ceil_div(n_rows, DIFF_BLOCK_ROWS);
    hipLaunchKernelGGL(sum_to_double_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, rowdiff, n_rows,
                       DIFF_BLOCK_ROWS, partials);
    ALS_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(sum_to_double_kernel<double>, dim3(1), dim3(256), 0, st, partials, nb, (int)nb, out);
    ALS_HIP_CHECK(hipGetLastError());
    return ALS_OK;
}

// shared with als_scoring.hip: records the calling thread's last error and returns `code`
int als_fail_(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ============================================================================= C ABI
extern "C" {

const char *als_last_error_string(void) { return g_err; }

int32_t als_abi_version(void) { return 8; }

// ----------------------------------------------------------------------------- plan
int32_t als_plan_count(const int64_t *rowptr, int64_t n_rows, int32_t chunk_nnz, int64_t *n_items, int64_t *n_long,
                       int64_t *n_slots) {
    if (!rowptr || n_rows < 0 || !n_items || !n_long || !n_slots)
        return fail(ALS_ERR_INVALID_ARG, "als_plan_count: null pointer or negative n_rows");
    if (chunk_nnz < 64 || chunk_nnz % 64) return fail(ALS_ERR_INVALID_ARG, "chunk_nnz must be a positive multiple of 64");
    if (n_rows > INT32_MAX) return fail(ALS_ERR_UNSUPPORTED, "more than 2^31-1 rows in one plan");
    int64_t items = 0, lng = 0, slots = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t nnz = rowptr[r + 1] - rowptr[r];
        if (nnz < 0) return fail(ALS_ERR_INVALID_ARG, "rowptr is not non-decreasing at row %lld", (long long)r);
        if (nnz > chunk_nnz) {
            const int64_t c = ceil_div(nnz, chunk_nnz);
            items += c;
            slots += c;
            lng += 1;
        } else {
            items += 1;
        }
    }
    if (slots > INT32_MAX) return fail(ALS_ERR_UNSUPPORTED, "too many partial slots");
    *n_items = items;
    *n_long = lng;
    *n_slots = slots;
    return ALS_OK;
}

int32_t als_plan_fill(const int64_t *rowptr, int64_t n_rows, int32_t chunk_nnz, als_item_t *items,
                      als_long_row_t *long_rows) {
    int64_t n_items, n_long, n_slots;
    const int32_t rc = als_plan_count(rowptr, n_rows, chunk_nnz, &n_items, &n_long, &n_slots);
    if (rc) return rc;
    if (!items || (n_long && !long_rows)) return fail(ALS_ERR_INVALID_ARG, "als_plan_fill: null output");
    // chunk items first (the largest work units), rows in order
    int64_t pos = 0, lpos = 0;
    int32_t slot = 0;
    std::vector<int64_t> bucket(chunk_nnz + 2, 0);  // counting sort of the short rows by nnz, descending
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t nnz = rowptr[r + 1] - rowptr[r];
        if (nnz > chunk_nnz) {
            const int32_t c = (int32_t)ceil_div(nnz, chunk_nnz);
            long_rows[lpos++] = als_long_row_t{(int32_t)r, slot, c, 0};
            for (int32_t k = 0; k < c; ++k) {
                const int64_t b = rowptr[r] + (int64_t)k * chunk_nnz;
                items[pos++] = als_item_t{(int32_t)r, k, slot++, (int32_t)std::min<int64_t>(chunk_nnz, rowptr[r + 1] - b), b, 0};
            }
        } else {
            bucket[chunk_nnz - nnz + 1] += 1;  // key = chunk_nnz - nnz: longest rows first
        }
    }
    for (int64_t k = 1; k < (int64_t)bucket.size(); ++k) bucket[k] += bucket[k - 1];
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t nnz = rowptr[r + 1] - rowptr[r];
        if (nnz <= chunk_nnz) items[pos + bucket[chunk_nnz - nnz]++] = als_item_t{(int32_t)r, -1, -1, (int32_t)nnz, rowptr[r], 0};
    }
    return ALS_OK;
}

// ----------------------------------------------------------------------------- gram
size_t als_gram_tiles_bytes(int32_t f) {
    if (f < 1 || f > ALS_MAX_FACTORS) return 0;
    if (f > ALS_MAX_FACTORS_EXACT) return als_gram_wide_tiles_bytes_(f);
    return (size_t)tq_for(f) * 64 * sizeof(float) + 256;      // the tiles, then the bound of |y| (gram_bound_kernel)
}

size_t als_gram_partial_bytes(int32_t f) {
    if (f < 1 || f > ALS_MAX_FACTORS) return 0;
    if (f > ALS_MAX_FACTORS_EXACT) return als_gram_wide_tiles_bytes_(f);
    return (size_t)tq_for(f) * 64 * sizeof(float);
}

size_t als_gram_workspace_bytes(int64_t n, int32_t f) {
    if (f < 1 || f > ALS_MAX_FACTORS || n < 0) return 0;
    if (f > ALS_MAX_FACTORS_EXACT) return als_gram_wide_workspace_bytes_(n, f);
    const size_t tile = (size_t)tq_for(f) * 64 * sizeof(float);
    const int64_t n_chunks = std::max<int64_t>(1, ceil_div(n, ALS_GRAM_CHUNK_ROWS));
    const int64_t n_super = std::max<int64_t>(1, ceil_div(n, ALS_GRAM_SUPER_ROWS));
    return (size_t)(n_chunks + n_super) * tile + 256;
}

int32_t als_gram_partials_f32(const float *Y, int64_t row_begin, int64_t row_end, int32_t f, int64_t ldy,
                              float *super_out, void *workspace, size_t ws_bytes, void *stream) {
    if (!Y || !super_out || !workspace) return fail(ALS_ERR_INVALID_ARG, "als_gram_partials_f32: null pointer");
    if (f < 1 || f > ALS_MAX_FACTORS) return fail(ALS_ERR_UNSUPPORTED, "f=%d outside [1, %d]", f, ALS_MAX_FACTORS);
    if (row_begin < 0 || row_end <= row_begin || row_begin % ALS_GRAM_SUPER_ROWS)
        return fail(ALS_ERR_INVALID_ARG, "row range [%lld, %lld) empty or row_begin not a multiple of %d",
                    (long long)row_begin, (long long)row_end, ALS_GRAM_SUPER_ROWS);
    if (row_end > INT32_MAX) return fail(ALS_ERR_UNSUPPORTED, "more than 2^31-1 factor rows");
    if (ldy < f) return fail(ALS_ERR_INVALID_ARG, "ldy < f");
    if (ws_bytes < als_gram_workspace_bytes(row_end - row_begin, f))
        return fail(ALS_ERR_WORKSPACE, "gram workspace too small: %zu < %zu", ws_bytes,
                    als_gram_workspace_bytes(row_end - row_begin, f));
    hipStream_t st = (hipStream_t)stream;
    if (f > ALS_MAX_FACTORS_EXACT) return als_gram_wide_partials_(Y, row_begin, row_end, f, ldy, super_out, workspace, st);
    switch (vw_for(f)) {
        case 1: return gram_partials_impl<1>(Y, row_begin, row_end, f, ldy, super_out, workspace, st);
        case 2: return gram_partials_impl<2>(Y, row_begin, row_end, f, ldy, super_out, workspace, st);
        case 4: return gram_partials_impl<4>(Y, row_begin, row_end, f, ldy, super_out, workspace, st);
        default: return gram_partials_impl<8>(Y, row_begin, row_end, f, ldy, super_out, workspace, st);
    }
}

int32_t als_gram_finalize_f32(const float *super_in, int64_t n_super, int32_t f, float lamb, float *gram_out,
                              float *tiles_out, void *stream) {
    if (!super_in || !gram_out || !tiles_out || n_super < 1)
        return fail(ALS_ERR_INVALID_ARG, "als_gram_finalize_f32: null pointer or n_super < 1");
    if (f < 1 || f > ALS_MAX_FACTORS) return fail(ALS_ERR_UNSUPPORTED, "f=%d outside [1, %d]", f, ALS_MAX_FACTORS);
    hipStream_t st = (hipStream_t)stream;
    if (f > ALS_MAX_FACTORS_EXACT) return als_gram_wide_finalize_(super_in, n_super, f, gram_out, tiles_out, st);
    const int elems = tq_for(f) * 64;
    const dim3 grid((elems + 255) / 256), block(256);
    switch (vw_for(f)) {
        case 1: hipLaunchKernelGGL(gram_finalize_kernel<1>, grid, block, 0, st, super_in, n_super, f, lamb, gram_out, tiles_out); break;
        case 2: hipLaunchKernelGGL(gram_finalize_kernel<2>, grid, block, 0, st, super_in, n_super, f, lamb, gram_out, tiles_out); break;
        case 4: hipLaunchKernelGGL(gram_finalize_kernel<4>, grid, block, 0, st, super_in, n_super, f, lamb, gram_out, tiles_out); break;
        default: hipLaunchKernelGGL(gram_finalize_kernel<8>, grid, block, 0, st, super_in, n_super, f, lamb, gram_out, tiles_out); break;
    }
    ALS_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(gram_bound_kernel, dim3(1), dim3(64), 0, st, gram_out, f, tiles_out + elems);
    ALS_HIP_CHECK(hipGetLastError());
    return ALS_OK;
}

int32_t als_gram_f32(const float *Y, int64_t n, int32_t f, int64_t ldy, float lamb, float *gram_out, float *tiles_out,
                     void *workspace, size_t ws_bytes, void *stream) {
    if (n < 1) return fail(ALS_ERR_INVALID_ARG, "als_gram_f32: n < 1");
    if (!workspace || ws_bytes < als_gram_workspace_bytes(n, f) || als_gram_workspace_bytes(n, f) == 0)
        return fail(ALS_ERR_WORKSPACE, "gram workspace too small or f unsupported");
    const size_t tile = f > ALS_MAX_FACTORS_EXACT ? als_gram_tiles_bytes(f) : (size_t)tq_for(f) * 64 * sizeof(float);
    const int64_t n_chunks = ceil_div(n, f > ALS_MAX_FACTORS_EXACT ? ALS_GRAM_WIDE_CHUNK_ROWS : ALS_GRAM_CHUNK_ROWS);
    const int64_t n_super = ceil_div(n, ALS_GRAM_SUPER_ROWS);
    float *super = (float *)((char *)workspace + (size_t)n_chunks * tile);
    int32_t rc = als_gram_partials_f32(Y, 0, n, f, ldy, super, workspace, ws_bytes, stream);
    if (rc) return rc;
    return als_gram_finalize_f32(super, n_super, f, lamb, gram_out, tiles_out, stream);
}

// ----------------------------------------------------------------------------- half sweep
size_t als_half_sweep_workspace_bytes(int64_t n_rows, int64_t n_slots, int32_t f) {
    if (f < 1 || f > ALS_MAX_FACTORS_EXACT || n_rows < 0 || n_slots < 0) return 0;
    const int vw = vw_for(f);
    const size_t pq = (size_t)(vw * (vw + 1) / 2 * 4 + vw) * 64 * sizeof(float);
    size_t b = align_up((size_t)n_slots * pq, 256);
    b += align_up((size_t)n_rows * sizeof(float), 256);
    b += align_up((size_t)(ceil_div(n_rows, DIFF_BLOCK_ROWS) + 1) * sizeof(double), 256);
    return b + 256;
}

int32_t als_half_sweep_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                               const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                               const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                               const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, int32_t f,
                               float alpha, float *Xown, int64_t ldx, double *absdiff_out, int32_t *info_out,
                               const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes, void *stream) {
    return als_half_sweep_tiles_(rowptr, col, val, n_rows, row_ids, items, n_items, long_rows, n_long, n_slots, chunk_nnz,
                                 Yopp, n_opp, ldy, gram_tiles, f, alpha, 0, Xown, ldx, absdiff_out, info_out, opts,
                                 workspace, ws_bytes, stream);
}

// ----------------------------------------------------------------------------- half sweep without a workspace argument
size_t als_static_scratch_bytes(void) { return ALS_STATIC_SCRATCH_BYTES; }

int32_t als_static_scratch_init(void) {
    void *p;
    return static_scratch_ptr(&p);
}

int32_t als_half_sweep_csr_static_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                                      const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                                      const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                                      const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, int32_t f,
                                      float alpha, float *Xown, int64_t ldx, double *absdiff_out, int32_t *info_out,
                                      const als_sweep_opts_t *opts, void *stream) {
    const size_t need = als_half_sweep_workspace_bytes(n_rows, n_slots, f);
    if (need == 0) return fail(ALS_ERR_UNSUPPORTED, "als_half_sweep_csr_static_f32: f=%d outside [1, %d] or negative sizes", f,
                               ALS_MAX_FACTORS_EXACT);
    if (need > ALS_STATIC_SCRATCH_BYTES)
        return fail(ALS_ERR_WORKSPACE, "als_half_sweep_csr_static_f32: n_rows=%lld, n_slots=%lld at f=%d need %zu bytes of scratch, "
                    "the built-in buffer holds %zu; use als_half_sweep_csr_f32 with a workspace", (long long)n_rows,
                    (long long)n_slots, f, need, (size_t)ALS_STATIC_SCRATCH_BYTES);
    void *ws;
    if (const int32_t rc = static_scratch_ptr(&ws)) return rc;
    // Nothing in the scratch has to start at zero: every partial slot, every row's |delta| and every partial sum is
    // written by this call before this call reads it, so the previous call's contents are never seen.
    return als_half_sweep_tiles_(rowptr, col, val, n_rows, row_ids, items, n_items, long_rows, n_long, n_slots, chunk_nnz,
                                 Yopp, n_opp, ldy, gram_tiles, f, alpha, 0, Xown, ldx, absdiff_out, info_out, opts, ws,
                                 ALS_STATIC_SCRATCH_BYTES, stream);
}

}  // extern "C"

// The half-sweep on the accumulator tiles: cg_steps == 0 solves every row exactly (the C entry point
// above), cg_steps > 0 takes that many CG steps on the formed matrix (als_half_sweep_cg_f32, als_cg.hip).
int32_t als_half_sweep_tiles_(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                              const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                              const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                              const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, int32_t f,
                              float alpha, int32_t cg_steps, float *Xown, int64_t ldx, double *absdiff_out,
                              int32_t *info_out, const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes,
                              void *stream) {
    SweepOpts so;
    if (const int32_t rc = read_opts(opts, so)) return rc;
    if (!rowptr || !items || !Yopp || !gram_tiles || !Xown || !workspace)
        return fail(ALS_ERR_INVALID_ARG, "als_half_sweep_csr_f32: null pointer");
    if (n_rows < 1 || n_rows > INT32_MAX || n_items < n_rows || n_long < 0 || n_slots < 0 || n_opp < 1)
        return fail(ALS_ERR_INVALID_ARG, "als_half_sweep_csr_f32: bad sizes (n_rows=%lld n_items=%lld n_opp=%lld)",
                    (long long)n_rows, (long long)n_items, (long long)n_opp);
    if ((n_long > 0 && !long_rows) || (n_long > 0) != (n_slots > 0) || n_items != n_rows - n_long + n_slots)
        return fail(ALS_ERR_INVALID_ARG, "plan sizes are inconsistent (n_items=%lld n_long=%lld n_slots=%lld)",
                    (long long)n_items, (long long)n_long, (long long)n_slots);
    if (chunk_nnz < 64 || chunk_nnz % 64) return fail(ALS_ERR_INVALID_ARG, "chunk_nnz must be a positive multiple of 64");
    if (f < 1 || f > ALS_MAX_FACTORS_EXACT)
        return fail(ALS_ERR_UNSUPPORTED, "f=%d outside [1, %d] (the exact solve; als_half_sweep_cg_f32 goes to %d)", f,
                    ALS_MAX_FACTORS_EXACT, ALS_MAX_FACTORS);
    if (ldy < f || ldx < f) return fail(ALS_ERR_INVALID_ARG, "leading dimension smaller than f");
    if (n_items > 0x7fffffffLL) return fail(ALS_ERR_UNSUPPORTED, "too many work items for one launch");
    if (ws_bytes < als_half_sweep_workspace_bytes(n_rows, n_slots, f))
        return fail(ALS_ERR_WORKSPACE, "half-sweep workspace too small: %zu < %zu", ws_bytes,
                    als_half_sweep_workspace_bytes(n_rows, n_slots, f));
    if ((!col || !val) && rowptr) {
        // a CSR without stored entries is legal only if every row is empty; the kernels never
        // dereference col/val then, but a null with entries would fault: refuse it here.
        return fail(ALS_ERR_INVALID_ARG, "col/val are null");
    }
    const int vw = vw_for(f);
    const size_t pq = (size_t)(vw * (vw + 1) / 2 * 4 + vw) * 64 * sizeof(float);
    char *ws = (char *)workspace;
    SweepParams p;
    p.rowptr = rowptr; p.col = col; p.val = val; p.row_ids = row_ids; p.items = items; p.long_rows = long_rows;
    p.Y = Yopp; p.gtiles = gram_tiles; p.X = Xown;
    p.partials = (float *)ws; ws += align_up((size_t)n_slots * pq, 256);
    p.rowdiff = (float *)ws; ws += align_up((size_t)n_rows * sizeof(float), 256);
    double *diff_partials = (double *)ws;
    ws += align_up((size_t)(ceil_div(n_rows, DIFF_BLOCK_ROWS) + 1) * sizeof(double), 256);
    p.stream_next = (int32_t *)ws;          // in the 256 spare bytes that end the workspace
    p.info = info_out; p.ldy = ldy; p.ldx = ldx; p.f = f; p.chunk_nnz = chunk_nnz; p.alpha = alpha;
    // The library's own choice of the fp16 x 2 accumulate (2^-21 per product against 2^-23) is limited to fixed factors
    // with at least 16 rows per factor: with fewer the Gram is close to singular (at 2 rows per factor and lamb = 0.01
    // the float32 path is at the limit of the 1e-4 bar already, see tests/core/test_ALS.py:27-29) and the larger
    // product error shows in the solution.
    if (so.f16 && so.f16_auto && n_opp < 16 * (int64_t)f) so.f16 = false;
    p.cg_steps = cg_steps; p.rmax = so.rmax;
    const bool vec = vec_ok(Yopp, ldy, f, n_opp);
    hipStream_t st = (hipStream_t)stream;
    switch (vw) {
        case 1: return half_sweep_impl<1>(p, n_rows, n_items, n_long, absdiff_out, diff_partials, vec, so, st);
        case 2: return half_sweep_impl<2>(p, n_rows, n_items, n_long, absdiff_out, diff_partials, vec, so, st);
        case 4: return half_sweep_impl<4>(p, n_rows, n_items, n_long, absdiff_out, diff_partials, vec, so, st);
        default: return half_sweep_impl<8>(p, n_rows, n_items, n_long, absdiff_out, diff_partials, vec, so, st);
    }
}

// 128 < f <= 256: CG on the formed matrix, one row per 4-wave workgroup (als_wide_row.inc).
size_t als_half_sweep_wide_workspace_bytes_(int64_t n_rows, int64_t n_slots) {
    size_t b = align_up((size_t)n_slots * WR_SLOT_FLOATS * sizeof(float), 256);
    b += align_up((size_t)n_rows * sizeof(float), 256);
    b += align_up((size_t)(ceil_div(n_rows, DIFF_BLOCK_ROWS) + 1) * sizeof(double), 256);
    return b + 256;
}
int32_t als_half_sweep_wide_(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                             const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                             const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots, int32_t chunk_nnz,
                             const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram_tiles, float lamb, int32_t f,
                             float alpha, int32_t cg_steps, float *Xown, int64_t ldx, double *absdiff_out,
                             int32_t *info_out, const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes,
                             void *stream) {
    SweepOpts so;
    if (const int32_t rc = read_opts(opts, so)) return rc;
    if (!rowptr || !col || !val || !items || !Yopp || !gram_tiles || !Xown || !workspace)
        return fail(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: null pointer");
    if (n_rows < 1 || n_rows > INT32_MAX || n_items < n_rows || n_long < 0 || n_slots < 0 || n_opp < 1)
        return fail(ALS_ERR_INVALID_ARG, "als_half_sweep_cg_f32: bad sizes (n_rows=%lld n_items=%lld n_opp=%lld)",
                    (long long)n_rows, (long long)n_items, (long long)n_opp);
    if ((n_long > 0 && !long_rows) || (n_long > 0) != (n_slots > 0) || n_items != n_rows - n_long + n_slots)
        return fail(ALS_ERR_INVALID_ARG, "plan sizes are inconsistent (n_items=%lld n_long=%lld n_slots=%lld)",
                    (long long)n_items, (long long)n_long, (long long)n_slots);
    if (chunk_nnz < 64 || chunk_nnz % 64) return fail(ALS_ERR_INVALID_ARG, "chunk_nnz must be a positive multiple of 64");
    if (f <= ALS_MAX_FACTORS_EXACT || f > ALS_MAX_FACTORS) return fail(ALS_ERR_UNSUPPORTED, "f=%d outside (128, 256]", f);
    if (ldy < f || ldx < f) return fail(ALS_ERR_INVALID_ARG, "leading dimension smaller than f");
    if (n_items > 0x7fffffffLL) return fail(ALS_ERR_UNSUPPORTED, "too many work items for one launch");
    if (ws_bytes < als_half_sweep_wide_workspace_bytes_(n_rows, n_slots))
        return fail(ALS_ERR_WORKSPACE, "cg workspace too small: %zu < %zu", ws_bytes,
                    als_half_sweep_wide_workspace_bytes_(n_rows, n_slots));
    char *ws = (char *)workspace;
    SweepParams p;
    p.rowptr = rowptr; p.col = col; p.val = val; p.row_ids = row_ids; p.items = items; p.long_rows = long_rows;
    p.Y = Yopp; p.gtiles = gram_tiles; p.X = Xown;
    p.partials = (float *)ws; ws += align_up((size_t)n_slots * WR_SLOT_FLOATS * sizeof(float), 256);
    p.rowdiff = (float *)ws; ws += align_up((size_t)n_rows * sizeof(float), 256);
    double *diff_partials = (double *)ws;
    p.info = info_out; p.ldy = ldy; p.ldx = ldx; p.f = f; p.chunk_nnz = chunk_nnz; p.alpha = alpha;
    p.cg_steps = cg_steps; p.lamb = lamb; p.rmax = 0.f; p.stream_next = nullptr;
    // 16-byte loads: whole vectors inside a row, aligned base, 32-bit row offsets in vector units
    const bool vec = (f % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)Yopp % 16 == 0) && (n_opp * (ldy / 4) < 0xffffffffLL);
    hipStream_t st = (hipStream_t)stream;
    if (so.tap0) ALS_HIP_CHECK(hipEventRecord(so.tap0, st));
    if (vec) hipLaunchKernelGGL(wide_row_items_kernel<true>, dim3((unsigned)n_items), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(wide_row_items_kernel<false>, dim3((unsigned)n_items), dim3(256), 0, st, p);
    ALS_HIP_CHECK(hipGetLastError());
    if (n_long > 0) {
        hipLaunchKernelGGL(sum_slots_kernel, dim3((unsigned)n_long, (unsigned)((WR_SLOT_FLOATS + 255) / 256)), dim3(256), 0,
                           st, p.partials, p.long_rows, WR_SLOT_FLOATS);
        ALS_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(wide_row_long_kernel, dim3((unsigned)n_long), dim3(256), 0, st, p);
        ALS_HIP_CHECK(hipGetLastError());
    }
    if (so.tap1) ALS_HIP_CHECK(hipEventRecord(so.tap1, st));
    if (absdiff_out) return als_sum_rowdiff_(p.rowdiff, n_rows, diff_partials, absdiff_out, st);
    return ALS_OK;
}
