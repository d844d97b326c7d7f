// Scoring side of libals_hip.so: dense user x item scores on the fp32 matrix cores, exact
// per-row top-k by radix select, pairwise prediction and row normalisation (cosine similarity).
//
//   scores kernel   one workgroup = 4 waves = a 64-user x 256-item tile; each wave owns 64 x 64
//                   outputs as 4 x 4 tiles of v_mfma_f32_16x16x4_f32.  Both operands are fetched
//                   with one 16-byte load per lane: lane (g, i) reads elements 16s+4g .. 16s+4g+3
//                   of its row, which are 4 consecutive k-steps for lane group g - the k order is
//                   a permutation shared by both operands, so no staging or shuffling is needed.
//   top-k kernel    one workgroup per score row: 4 x 8-bit radix-select passes find the k-th
//                   largest key, a count pass and an ordered ballot compaction collect the
//                   winners (ties by ascending item id), a bitonic sort in LDS orders them.
//                   Integer LDS atomics only: the result is deterministic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "als_hip.h"

int als_fail_(int code, const char *fmt, ...);  // defined in als_kernels.hip

#define ALS_HIP_CHECK2(expr)                                                                      \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return als_fail_(ALS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------- scores
// 4 consecutive elements k0..k0+3 of a factor row, zero beyond f.  VEC: one 16-byte load.
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float *__restrict__ row, const int k0, const int f) {
    if constexpr (VEC) {
        const bool ok = k0 < f;   // f % 4 == 0: the vector is entirely inside or outside
        const f32x4 x = *reinterpret_cast<const f32x4 *>(row + (ok ? k0 : 0));
        return ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        f32x4 x;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = k0 + t < f;
            const float v = row[ok ? k0 + t : 0];
            x[t] = ok ? v : 0.f;
        }
        return x;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void scores_kernel(const float *__restrict__ X, const int64_t ldx,
                                                    const int64_t *__restrict__ user_ids, const int64_t nu,
                                                    const float *__restrict__ Y, const int64_t ldy, const int64_t M,
                                                    const int f, float *__restrict__ S, const int64_t lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int64_t u0 = (int64_t)blockIdx.y * 64;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + wave * 64;
    if (i0 >= M) return;
    const float *xrow[4];
    const float *yrow[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t u = min(u0 + 16 * a + j, nu - 1);              // clamped: stores are masked below
        xrow[a] = X + (user_ids ? user_ids[u] : u) * ldx;
        yrow[a] = Y + min(i0 + 16 * a + j, M - 1) * ldy;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < f; s += 16) {
        f32x4 xa[4], yb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            xa[a] = load4<VEC>(xrow[a], s + 4 * g, f);
            yb[a] = load4<VEC>(yrow[a], s + 4 * g, f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[a][t], yb[b][t], acc[a][b], 0, 0, 0);
    }
    // C layout: lane (g, j), register r holds (row 4g + r, column j) of each 16 x 16 tile
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t u = u0 + 16 * a + 4 * g + r;
            if (u >= nu) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t i = i0 + 16 * b + j;
                if (i < M) S[u * lds + i] = acc[a][b][r];
            }
        }
}

// ----------------------------------------------------------------------------- top-k
__device__ __forceinline__ uint32_t order_key(const float x) {
    // monotone map float -> uint32 (larger float = larger key); NaN sorts below everything
    if (x != x) return 0u;
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(const uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

constexpr int TOPK_MAX = 1024;

__global__ __launch_bounds__(256) void topk_rows_kernel(float *__restrict__ S, const int64_t lds, const int64_t M,
                                                       const int64_t *__restrict__ ex_rowptr,
                                                       const int32_t *__restrict__ ex_col, const int k,
                                                       int32_t *__restrict__ idx_out, float *__restrict__ score_out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_key[TOPK_MAX];
    __shared__ int32_t sel_idx[TOPK_MAX];
    __shared__ uint32_t sh_prefix, sh_remaining;
    __shared__ uint32_t wave_gt[4], wave_eq[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *row = S + (int64_t)blockIdx.x * lds;

    // 0. excluded items (e.g. already rated, RecommenderService.py:16-29) drop to -inf
    if (ex_rowptr) {
        const int64_t s = ex_rowptr[blockIdx.x], e = ex_rowptr[blockIdx.x + 1];
        for (int64_t q = s + tid; q < e; q += 256) {
            const int c = ex_col[q];
            if (c >= 0 && c < M) row[c] = -INFINITY;
        }
        __threadfence_block();
        __syncthreads();
    }
    // 1. radix select: the key of rank k (1-based, descending)
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int64_t i = tid; i < M; i += 256) {
            const uint32_t key = order_key(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t above = 0;
            int b = 255;
            for (; b > 0; --b) {
                if (above + hist[b] >= remaining) break;
                above += hist[b];
            }
            sh_prefix = prefix | ((uint32_t)b << shift);
            sh_remaining = remaining - above;
        }
        __syncthreads();
        prefix = sh_prefix;
        remaining = sh_remaining;
        mask |= 0xffu << shift;
        __syncthreads();
    }
    const uint32_t T = prefix;               // k-th largest key; take `remaining` of the ties, lowest ids first
    const uint32_t n_gt = (uint32_t)k - remaining;
    // 2. each wave owns a contiguous quarter of the row: count, then ordered compaction
    const int64_t seg = (M + 3) / 4;
    const int64_t s0 = min(M, seg * wave), s1 = min(M, s0 + seg);
    uint32_t cgt = 0, ceq = 0;
    for (int64_t i = s0 + lane; i < s1; i += 64) {
        const uint32_t key = order_key(row[i]);
        cgt += key > T;
        ceq += key == T;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        cgt += __shfl_xor(cgt, o);
        ceq += __shfl_xor(ceq, o);
    }
    if (lane == 0) {
        wave_gt[wave] = cgt;
        wave_eq[wave] = ceq;
    }
    __syncthreads();
    uint32_t off_gt = 0, off_eq = 0;
    for (int w = 0; w < wave; ++w) {
        off_gt += wave_gt[w];
        off_eq += wave_eq[w];
    }
    for (int64_t base = s0; base < s1; base += 64) {
        const int64_t i = base + lane;
        const uint32_t key = i < s1 ? order_key(row[i]) : 0u;
        const bool gt = i < s1 && key > T, eq = i < s1 && key == T;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (gt) {
            const uint32_t p = off_gt + (uint32_t)__popcll(bg & below);
            sel_key[p] = key;
            sel_idx[p] = (int32_t)i;
        }
        if (eq) {
            const uint32_t p = off_eq + (uint32_t)__popcll(be & below);
            if (p < remaining) {
                sel_key[n_gt + p] = key;
                sel_idx[n_gt + p] = (int32_t)i;
            }
        }
        off_gt += (uint32_t)__popcll(bg);
        off_eq += (uint32_t)__popcll(be);
    }
    // 3. bitonic sort, descending by (key, -idx)
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int q = k + tid; q < n2; q += 256) {
        sel_key[q] = 0u;
        sel_idx[q] = INT32_MAX;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = tid; q < n2 / 2; q += 256) {
                const int lo = 2 * q - (q & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint32_t ka = sel_key[lo], kb = sel_key[hi];
                const int32_t ia = sel_idx[lo], ib = sel_idx[hi];
                const bool a_first = (ka > kb) || (ka == kb && ia < ib);   // a ranks before b
                if (a_first != desc) {
                    sel_key[lo] = kb; sel_key[hi] = ka;
                    sel_idx[lo] = ib; sel_idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int q = tid; q < k; q += 256) {
        idx_out[(int64_t)blockIdx.x * k + q] = sel_idx[q];
        score_out[(int64_t)blockIdx.x * k + q] = key_to_float(sel_key[q]);
    }
}

// ----------------------------------------------------------------------------- small ops
__global__ void predict_pairs_kernel(const float *__restrict__ X, const int64_t ldx, const float *__restrict__ Y,
                                     const int64_t ldy, const int64_t *__restrict__ u, const int64_t *__restrict__ i,
                                     const int64_t n, const int f, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *x = X + u[p] * ldx, *y = Y + i[p] * ldy;
    float acc = 0.f;
    for (int k = 0; k < f; ++k) acc = fmaf(x[k], y[k], acc);
    out[p] = acc;
}

// out[r] = E[r] / ||E[r]||_2 ; rows of zero norm become 0 (the reference replaces NaN by 0)
__global__ __launch_bounds__(64) void normalize_rows_kernel(const float *__restrict__ E, const int64_t lde,
                                                           const int64_t n, const int f, float *__restrict__ out,
                                                           const int64_t ldo) {
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;
    float ss = 0.f;
    for (int k = lane; k < f; k += 64) {
        const float v = E[r * lde + k];
        ss = fmaf(v, v, ss);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    for (int k = lane; k < f; k += 64) {
        const float q = E[r * lde + k] / nrm;
        out[r * ldo + k] = (q != q) ? 0.f : q;
    }
}

}  // namespace

extern "C" {

int32_t als_scores_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y, int64_t ldy,
                       int64_t M, int32_t f, float *scores, int64_t lds, void *stream) {
    if (!X || !Y || !scores) return als_fail_(ALS_ERR_INVALID_ARG, "als_scores_f32: null pointer");
    if (nu < 1 || M < 1 || f < 1 || ldx < f || ldy < f || lds < M)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_scores_f32: bad sizes (nu=%lld M=%lld f=%d)", (long long)nu,
                         (long long)M, f);
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)((nu + 63) / 64)), block(256);
    if (grid.y > 65535) return als_fail_(ALS_ERR_UNSUPPORTED, "als_scores_f32: more than 65535*64 users per call");
    const bool vec = (f % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                     ((uintptr_t)Y % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(scores_kernel<true>, grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, M, f, scores, lds);
    else
        hipLaunchKernelGGL(scores_kernel<false>, grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, M, f, scores, lds);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

int32_t als_topk_rows_f32(float *scores, int64_t lds, int64_t n_rows, int64_t M, const int64_t *ex_rowptr,
                          const int32_t *ex_col, int32_t k, int32_t *idx_out, float *score_out, void *stream) {
    if (!scores || !idx_out || !score_out) return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: null pointer");
    if (n_rows < 1 || M < 1 || lds < M || M > INT32_MAX)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: bad sizes");
    if (k < 1 || k > TOPK_MAX || k > M)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_topk_rows_f32: k=%d outside [1, min(%d, M)]", k, TOPK_MAX);
    if ((ex_rowptr == nullptr) != (ex_col == nullptr))
        return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: pass both exclusion arrays or neither");
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, lds, M,
                       ex_rowptr, ex_col, k, idx_out, score_out);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

int32_t als_predict_pairs_f32(const float *X, int64_t ldx, const float *Y, int64_t ldy, const int64_t *u,
                              const int64_t *i, int64_t n, int32_t f, float *out, void *stream) {
    if (!X || !Y || !u || !i || !out) return als_fail_(ALS_ERR_INVALID_ARG, "als_predict_pairs_f32: null pointer");
    if (n < 1 || f < 1 || ldx < f || ldy < f) return als_fail_(ALS_ERR_INVALID_ARG, "als_predict_pairs_f32: bad sizes");
    hipLaunchKernelGGL(predict_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       ldx, Y, ldy, u, i, n, f, out);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

int32_t als_normalize_rows_f32(const float *E, int64_t lde, int64_t n, int32_t f, float *out, int64_t ldo,
                               void *stream) {
    if (!E || !out) return als_fail_(ALS_ERR_INVALID_ARG, "als_normalize_rows_f32: null pointer");
    if (n < 1 || f < 1 || lde < f || ldo < f) return als_fail_(ALS_ERR_INVALID_ARG, "als_normalize_rows_f32: bad sizes");
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, E, lde, n, f, out,
                       ldo);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

}  // extern "C"
