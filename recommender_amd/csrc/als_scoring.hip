// Scoring side of libals_hip.so: dense user x item scores on the matrix cores, exact per-row
// top-k by radix select, the fused top-k that never stores the score matrix, pairwise prediction
// and row normalisation (cosine similarity).
//
//   scores_split_kernel  f <= 128, 16-byte aligned rows: operands cut exactly into three bf16 pieces,
//                   6 piece products per block on v_mfma_f32_16x16x32_bf16; item fragments staged
//                   through LDS (described at the kernel).  FILTER form: the epilogue keeps only
//                   scores above the user's threshold (fused top-k).
//   scores_kernel   any f, any alignment: one workgroup = 4 waves = a 64-user x 256-item tile; each
//                   wave owns 64 x 64 outputs as 4 x 4 tiles of v_mfma_f32_16x16x4_f32.  Both
//                   operands are fetched with one 16-byte load per lane: lane (g, i) reads elements
//                   16s+4g .. 16s+4g+3 of its row, which are 4 consecutive k-steps for lane group g -
//                   the k order is a permutation shared by both operands, so nothing is staged.
//   top-k kernel    one workgroup per score row: 4 x 8-bit radix-select passes find the k-th
//                   largest key, a count pass and an ordered ballot compaction collect the
//                   winners (ties by ascending item id), a bitonic sort in LDS orders them.
//                   Integer LDS atomics only: the result is deterministic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <type_traits>

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

// Candidate lists of the fused top-k: row u holds cap (score, item) slots; slots [0, k) are filled
// beforehand with the top-k of the first item_begin items, count[u] starts at k.
struct CandArgs {
    float *score;
    int32_t *item;
    int32_t *count;
    int32_t *overflow;
    const int64_t *ex_rowptr;
    const int32_t *ex_col;
    const int64_t *ex_rows;     // CSR row of output row u (NULL: u itself)
    int32_t cap, k;
};

__device__ __forceinline__ bool excluded_item(const CandArgs &c, const int64_t u, const int item) {
    if (!c.ex_rowptr) return false;
    const int64_t er = c.ex_rows ? c.ex_rows[u] : u;
    int64_t lo = c.ex_rowptr[er], hi = c.ex_rowptr[er + 1];
    while (lo < hi) {          // the exclusion rows are sorted (canonical CSR)
        const int64_t mid = (lo + hi) >> 1;
        const int v = c.ex_col[mid];
        if (v == item) return true;
        if (v < item) lo = mid + 1;
        else hi = mid;
    }
    return false;
}

// FILTER = false: S[u][i] = <X[u], Y[i]> for items [0, M).
// FILTER = true:  nothing is stored; items are [item_begin, M) and a score strictly above the
//                 user's threshold (slot k-1 of its candidate row: the k-th best of the earlier
//                 items) is appended to the candidate row.  An item tied with the threshold cannot
//                 displace the k earlier items that reach it, because ties rank by ascending id.
template <bool VEC, bool FILTER>
__global__ __launch_bounds__(256) void scores_kernel(const float *__restrict__ X, const int64_t ldx,
                                                    const int64_t *__restrict__ user_ids, const int64_t nu,
                                                    const float *__restrict__ Y, const int64_t ldy, const int64_t M,
                                                    const int f, float *__restrict__ S, const int64_t lds,
                                                    const int64_t item_begin, const CandArgs cand) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int64_t u0 = (int64_t)blockIdx.y * 64;
    const int64_t i0 = item_begin + (int64_t)blockIdx.x * 256 + wave * 64;
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
            if constexpr (FILTER) {
                const float tau = cand.score[u * cand.cap + cand.k - 1];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t i = i0 + 16 * b + j;
                    const float sc = acc[a][b][r];
                    if (i < M && sc > tau) {       // excluded items are dropped when the candidates are ordered
                        const int pos = atomicAdd(&cand.count[u], 1);      // integer: which slot, not what value
                        if (pos < cand.cap) {
                            cand.score[u * cand.cap + pos] = sc;
                            cand.item[u * cand.cap + pos] = (int32_t)i;
                        } else {
                            atomicAdd(cand.overflow, 1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t i = i0 + 16 * b + j;
                    if (i < M) S[u * lds + i] = acc[a][b][r];
                }
            }
        }
}

// ----------------------------------------------------------------------------- scores on the bf16 pipe
// The same scores with every operand cut into three bf16 pieces by truncation (x = hi + mid + lo
// exactly, 3 x 8 mantissa bits) and the 6 piece products of relative size >= 2^-16 issued as
// v_mfma_f32_16x16x32_bf16 (fp32 accumulate; the dropped products are at fp32 rounding).  Six of
// these cost 96 matrix-pipe cycles per 16 x 16 x 32 block against 256 for the fp32 form.
//   workgroup   4 waves x 32 users = 128 users; a wave keeps its users' fragments (KS k-steps of
//               32, 3 pieces) in registers for the whole launch and walks a range of items
//   stage       32 items: the workgroup loads them once (one row segment of 32 bytes per thread),
//               splits them and writes the fragments to LDS in operand order - fragment
//               (item tile, k-step, piece) is 64 consecutive 16-byte lane slots, so writes and reads
//               are linear in the lane index; two stages, one barrier per stage
//   grid        workgroup ids go round the 8 XCDs, so the ids are laid out for the workgroups that
//               stream the same items (one per 128-user block) to sit on the same XCD and share its L2
// A score depends only on its user row and item row (fixed k order), not on the tiling.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_trunc(const float a, const float b) {   // two truncated bf16: a low, b high
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(const float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

struct Frag3 {
    u32x4 h, m, l;
};
// 8 consecutive k of one row -> the lane's operand fragment of each piece
__device__ __forceinline__ Frag3 split8(const f32x4 &a, const f32x4 &b) {
    Frag3 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? a[2 * p] : b[2 * p - 4], x1 = p < 2 ? a[2 * p + 1] : b[2 * p - 3];
        const float r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
        const float l0 = r0 - trunc_bf16(r0), l1 = r1 - trunc_bf16(r1);
        o.h[p] = pack_trunc(x0, x1);
        o.m[p] = pack_trunc(r0, r1);
        o.l[p] = pack_trunc(l0, l1);
    }
    return o;
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is reserved, and written here on purpose: the LDS address of the DMA
// 64 lanes x 16 bytes from g (per lane) to LDS bytes [lds, lds + 1024), lane l at 16 l
__device__ __forceinline__ void score_dma16(const void *g, const unsigned lds) {
    // (the address is wave uniform; readfirstlane makes the compiler keep it in a scalar register where it cannot tell)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "s"(__builtin_amdgcn_readfirstlane((int)lds)) : "memory", "m0");
}
#pragma clang diagnostic pop

#include "als_fp16.inc"
#include "als_scores_prepared.inc"

constexpr int SB_USERS = 128;   // users per workgroup
constexpr int SB_ITEMS = 32;    // items per stage

template <int KS, bool FILTER>
__global__ __launch_bounds__(256, 2) void scores_split_kernel(const float *__restrict__ X, const int64_t ldx,
                                                             const int64_t *__restrict__ user_ids, const int64_t nu,
                                                             const float *__restrict__ Y, const int64_t ldy,
                                                             const int64_t M, const int f, float *__restrict__ S,
                                                             const int64_t lds, const int64_t item_begin,
                                                             const int64_t stages_per_chunk, const int64_t n_user_blocks,
                                                             const int chunks_per_round,
                                                             const CandArgs cand) {
    constexpr int FRAGS = 2 * KS * 3;                  // fragments per stage
    constexpr int PAIRS = 2 * KS * 64;                 // (item, 8 k) row segments per stage
    constexpr int NP = (PAIRS + 255) / 256;            // per thread
    __shared__ u32x4 stage[2][FRAGS * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int64_t w = blockIdx.x, per_round = chunks_per_round * n_user_blocks;
    const int64_t ub = (w % per_round) / chunks_per_round, chunk = (w / per_round) * chunks_per_round + w % chunks_per_round;
    const int64_t n_stages = (M - item_begin + SB_ITEMS - 1) / SB_ITEMS;
    const int64_t t0 = chunk * stages_per_chunk, t1 = min(t0 + stages_per_chunk, n_stages);
    if (t0 >= t1) return;                              // the whole workgroup
    const int64_t u0 = ub * SB_USERS + wave * 32;

    u32x4 AH[2][KS], AM[2][KS], AL[2][KS];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int64_t u = min(u0 + 16 * a + j, nu - 1);              // clamped: results are masked below
        const float *xrow = X + (user_ids ? user_ids[u] : u) * ldx;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const Frag3 fr = split8(load4<true>(xrow, 32 * s + 8 * g, f), load4<true>(xrow, 32 * s + 8 * g + 4, f));
            AH[a][s] = fr.h; AM[a][s] = fr.m; AL[a][s] = fr.l;
        }
    }
    float tau[2][4];
    if constexpr (FILTER) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tau[a][r] = cand.score[min(u0 + 16 * a + 4 * g + r, nu - 1) * cand.cap + cand.k - 1];
    }

    // The item rows travel global -> LDS by LDS-DMA (no destination registers) into a per-thread parking area (`raw`:
    // a thread later reads back exactly the 16-byte pieces its own lane loaded), two stages ahead of the matrix work.
    // The DMA is inline assembly and its wait an explicit count: left to the compiler, the conditional atomics and
    // stores of the FILTER epilogue (they share the vector-memory counter) make it drain every load before the
    // epilogue ends - a prefetch one stage deep whatever the source says - and a known LDS-DMA makes it drain the
    // counter before every LDS read.  Unknown to it, the loads of stage t + 2 stay in flight through all of stage
    // t + 1.  Operations issued behind a load (the epilogue's) only make the counted wait stricter, never wrong.
    // Every stage issues and commits, also past the range's end (rows clamp to the last item, their fragments go to
    // the LDS buffer nobody reads): one fixed shape.
    __shared__ f32x4 raw[2][2 * NP][256];
    const unsigned raw_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&raw[0][0][0] + wave * 1024;
    auto issue = [&](auto set_c, const int64_t t) {
        constexpr int set = decltype(set_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // commit's reads of this set have returned
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const int p = tid + 256 * n;               // wave uniform validity: PAIRS is a multiple of 64
            if (p < PAIRS) {
                const int c = p >> 6, it = c / KS, s = c % KS, k0 = 32 * s + 8 * g;
                const int64_t item = min(item_begin + t * SB_ITEMS + it * 16 + j, M - 1);
                const float *yrow = Y + item * ldy;
                score_dma16(yrow + (k0 < f ? k0 : 0), raw_base + ((set * 2 * NP + 2 * n) * 256) * 16);
                score_dma16(yrow + (k0 + 4 < f ? k0 + 4 : 0), raw_base + ((set * 2 * NP + 2 * n + 1) * 256) * 16);
            }
        }
    };
    auto commit = [&](auto set_c, const int buf) {
        constexpr int set = decltype(set_c)::value;
        // behind this set's loads the wave issued the other set's: 2 per row segment it owns (wave uniform: the last
        // segment index exists only for the first PAIRS - 256 (NP - 1) threads)
        static_assert(NP == 1 || NP == 2, "wait counts below");
        const bool all_segments = tid + 256 * (NP - 1) < PAIRS;
        if (NP == 2 && all_segments) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if ((NP == 2) != all_segments) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const int p = tid + 256 * n;
            if (p < PAIRS) {
                const int k0 = 32 * ((p >> 6) % KS) + 8 * g;
                const f32x4 zero{0.f, 0.f, 0.f, 0.f};
                const f32x4 ya = raw[set][2 * n][tid], yb = raw[set][2 * n + 1][tid];
                const Frag3 fr = split8(k0 < f ? ya : zero, k0 + 4 < f ? yb : zero);
                u32x4 *dst = &stage[buf][(p >> 6) * 3 * 64 + lane];
                dst[0] = fr.h; dst[64] = fr.m; dst[128] = fr.l;
            }
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    issue(Set0{}, t0);
    issue(Set1{}, t0 + 1);
    commit(Set0{}, 0);
    issue(Set0{}, t0 + 2);
    __syncthreads();
    // FILTER: a hit reserves its slot in the user's candidate row with an integer atomic, whose answer
    // takes about as long as a stage.  The lane parks the hit and writes it after the next stage's
    // products (or at the end); only a lane's second hit within one stage waits for the first one's slot.
    bool pend = false;
    float pend_sc = 0.f;
    int32_t pend_item = 0, pend_pos = 0;
    int64_t pend_row = 0;
    auto flush = [&]() {
        if (pend) {
            if (pend_pos < cand.cap) {
                cand.score[pend_row + pend_pos] = pend_sc;
                cand.item[pend_row + pend_pos] = pend_item;
            } else {
                atomicAdd(cand.overflow, 1);
            }
            pend = false;
        }
    };
    // one stage: stage t sits in LDS buffer `par`; the rows of stage t + 1 are in set par ^ 1, those of t + 2 in set par
    auto step = [&](auto par_c, const int64_t t) {
        constexpr int par = decltype(par_c)::value;
        constexpr int buf = par;
        using Other = std::integral_constant<int, par ^ 1>;
        f32x4 acc[2][2];                               // [item tile][user tile]
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[it][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 BH[2], BM[2], BL[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const u32x4 *src = &stage[buf][(it * KS + s) * 3 * 64 + lane];
                BH[it] = src[0]; BM[it] = src[64]; BL[it] = src[128];
            }
            // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AL[a][s], BH[it], acc[it][a]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AH[a][s], BL[it], acc[it][a]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AM[a][s], BM[it], acc[it][a]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AM[a][s], BH[it], acc[it][a]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AH[a][s], BM[it], acc[it][a]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[it][a] = mfma_bf16(AH[a][s], BH[it], acc[it][a]);
        }
        // C layout: lane (g, j), register r holds (user 4g + r, item j) of each 16 x 16 tile
        if constexpr (FILTER) {
            flush();                 // last stage's hit: its slot number has arrived by now
            // most (user row, stage) pairs hold no score above the row's threshold: one wave-wide test
            // per register (4 user rows x 32 items), the per-score code only behind it
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!__any(fmaxf(acc[0][a][r], acc[1][a][r]) > tau[a][r])) continue;
                    const int64_t u = u0 + 16 * a + 4 * g + r;
                    if (u >= nu) continue;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int64_t i = item_begin + t * SB_ITEMS + it * 16 + j;
                        const float sc = acc[it][a][r];
                        if (i < M && sc > tau[a][r]) {      // excluded items are dropped when the candidates are ordered
                            flush();             // only a lane's second hit within one stage waits here
                            pend = true;
                            pend_pos = atomicAdd(&cand.count[u], 1);      // integer: which slot, not what value
                            pend_sc = sc; pend_item = (int32_t)i; pend_row = u * cand.cap;
                        }
                    }
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t u = u0 + 16 * a + 4 * g + r;
                    if (u >= nu) continue;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int64_t i = item_begin + t * SB_ITEMS + it * 16 + j;
                        if (i < M) S[u * lds + i] = acc[it][a][r];
                    }
                }
        }
        commit(Other{}, buf ^ 1);
        issue(Other{}, t + 3);                       // into the parking area commit has just read
        __syncthreads();
    };
    for (int64_t t = t0; t < t1; t += 2) {
        step(Set0{}, t);
        if (t + 1 < t1) step(Set1{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of this wave is in flight towards its LDS when it ends
    if constexpr (FILTER) flush();
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
constexpr int TOPK_REG_ITEMS = 16384;

__global__ __launch_bounds__(256) void topk_rows_kernel(float *__restrict__ S, const int64_t lds, const int64_t M,
                                                       const int64_t *__restrict__ ex_rowptr,
                                                       const int32_t *__restrict__ ex_col,
                                                       const int64_t *__restrict__ ex_rows, const int k,
                                                       int32_t *__restrict__ idx_out, float *__restrict__ score_out,
                                                       const int64_t out_stride) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_key[TOPK_MAX];
    __shared__ int32_t sel_idx[TOPK_MAX];
    __shared__ uint32_t sh_prefix, sh_remaining;
    __shared__ uint32_t wave_gt[4], wave_eq[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *row = S + (int64_t)blockIdx.x * lds;
    auto key_at = [&](const int64_t i) { return order_key(row[i]); };

    // 0. excluded items (e.g. already rated, RecommenderService.py:16-29) drop to -inf
    if (ex_rowptr) {
        const int64_t er = ex_rows ? ex_rows[blockIdx.x] : (int64_t)blockIdx.x;
        const int64_t s = ex_rowptr[er], e = ex_rowptr[er + 1];
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
            const uint32_t key = key_at(i);
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
        const uint32_t key = key_at(i);
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
        const uint32_t key = i < s1 ? key_at(i) : 0u;
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
        idx_out[(int64_t)blockIdx.x * out_stride + q] = sel_idx[q];
        score_out[(int64_t)blockIdx.x * out_stride + q] = key_to_float(sel_key[q]);
    }
}

// The same selection for rows of at most TOPK_REG_ITEMS scores (the first range of the fused top-k): a thread keeps its
// 64 keys (items q 256 + tid) in registers and the row is read once.  The k-th largest key is found bit by bit - T takes
// bit b if at least k keys are >= T | 1 << b: 32 counting rounds over registers, no histogram, no atomics (on scores of
// one magnitude the top-byte histogram of the kernel above funnels the whole row through a handful of LDS counters).
// Then the keys above T and the first `remaining` equal to it, in item order: per (q, wave) segment the counts go
// through LDS, one scan gives every segment its offset.  Same sort, same output as topk_rows_kernel.
__global__ __launch_bounds__(256) void topk_rows_reg_kernel(float *__restrict__ S, const int64_t lds, const int64_t M,
                                                           const int64_t *__restrict__ ex_rowptr,
                                                           const int32_t *__restrict__ ex_col,
                                                           const int64_t *__restrict__ ex_rows, const int k,
                                                           int32_t *__restrict__ idx_out, float *__restrict__ score_out,
                                                           const int64_t out_stride) {
    constexpr int Q = TOPK_REG_ITEMS / 256;
    __shared__ uint32_t sel_key[TOPK_MAX];
    __shared__ int32_t sel_idx[TOPK_MAX];
    __shared__ uint32_t wave_cnt[2][4];
    __shared__ uint32_t seg_gt[Q * 4], seg_eq[Q * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *row = S + (int64_t)blockIdx.x * lds;
    if (ex_rowptr) {
        const int64_t er = ex_rows ? ex_rows[blockIdx.x] : (int64_t)blockIdx.x;
        const int64_t s = ex_rowptr[er], e = ex_rowptr[er + 1];
        for (int64_t q = s + tid; q < e; q += 256) {
            const int c = ex_col[q];
            if (c >= 0 && c < M) row[c] = -INFINITY;
        }
        __threadfence_block();
        __syncthreads();
    }
    uint32_t key[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = q * 256 + tid;
        const float v = row[i < M ? i : (int)M - 1];
        key[q] = i < M ? order_key(v) : 0u;           // past the row: the lowest key, behind every item of equal key
        if (q % 8 == 7) __builtin_amdgcn_sched_barrier(0);      // 8 loads in flight; not 64 lane masks alive at once
    }
    // 1. the k-th largest key
    auto count_all = [&](uint32_t c, const int par) {      // sum over the workgroup; two LDS slots by parity: one barrier
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) wave_cnt[par][wave] = c;
        __syncthreads();
        return wave_cnt[par][0] + wave_cnt[par][1] + wave_cnt[par][2] + wave_cnt[par][3];
    };
    uint32_t T = 0;
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) c += key[q] >= cand;
        if (count_all(c, b & 1) >= (uint32_t)k) T = cand;
    }
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) c += key[q] > T;
    __syncthreads();
    const uint32_t n_gt = count_all(c, 0), remaining = (uint32_t)k - n_gt;
    // 2. ordered compaction: segment (q, wave) = items q 256 + 64 wave .. + 63.  (T is made opaque before each of the
    // two loops: knowing it to be one value the compiler keeps all 128 lane masks of the comparisons alive from one
    // loop to the next, in scalar registers it does not have.)
    asm volatile("" : "+v"(T));
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const unsigned long long bg = __ballot(key[q] > T), be = __ballot(key[q] == T);
        if (lane == 0) {
            seg_gt[q * 4 + wave] = (uint32_t)__popcll(bg);
            seg_eq[q * 4 + wave] = (uint32_t)__popcll(be);
        }
        __builtin_amdgcn_sched_barrier(0);         // one segment at a time: 128 lane masks do not fit the scalar registers
    }
    __syncthreads();
    {   // exclusive scan of the Q * 4 = 256 segment counts, one per thread
        uint32_t g = seg_gt[tid], e = seg_eq[tid];
        const uint32_t g0 = g, e0 = e;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t tg = __shfl_up(g, o), te = __shfl_up(e, o);
            if (lane >= o) { g += tg; e += te; }
        }
        __syncthreads();                       // every count has been read
        if (lane == 63) { wave_cnt[0][wave] = g; wave_cnt[1][wave] = e; }
        __syncthreads();
        for (int w = 0; w < wave; ++w) { g += wave_cnt[0][w]; e += wave_cnt[1][w]; }
        seg_gt[tid] = g - g0;
        seg_eq[tid] = e - e0;
        __syncthreads();
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    asm volatile("" : "+v"(T));
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const unsigned long long bg = __ballot(key[q] > T), be = __ballot(key[q] == T);
        if ((bg | be) == 0) continue;              // nearly every segment: nothing of it is selected
        if (key[q] > T) {
            const uint32_t p = seg_gt[q * 4 + wave] + (uint32_t)__popcll(bg & below);
            sel_key[p] = key[q];
            sel_idx[p] = q * 256 + tid;
        } else if (key[q] == T) {
            const uint32_t p = seg_eq[q * 4 + wave] + (uint32_t)__popcll(be & below);
            if (p < remaining) {
                sel_key[n_gt + p] = key[q];
                sel_idx[n_gt + p] = q * 256 + tid;
            }
        }
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
        idx_out[(int64_t)blockIdx.x * out_stride + q] = sel_idx[q];
        score_out[(int64_t)blockIdx.x * out_stride + q] = key_to_float(sel_key[q]);
    }
}

// Final selection of the fused top-k: one workgroup per user sorts the user's candidates
// (score descending, item id ascending - the order does not depend on the order of arrival)
// in LDS and writes the first k.
constexpr int CAND_CAP = 8192;
// idx_out == nullptr: the first k go back to the head of the candidate row and the row's count
// becomes k (the row's threshold, slot k-1, tightens for the items still to come).
// Two instances run one after the other: LCAP = CAND_SMALL takes the rows with at most CAND_SMALL candidates (nearly
// all: a range adds about 3 k) and the rows to be given up, with 16 KB of LDS - eight workgroups to a CU instead of the
// two that sorting space for a full row allows; LCAP = CAND_CAP takes the rest (it finds count = k on every row
// the first instance handled and leaves those alone).
constexpr int CAND_SMALL = 2048;
template <int LCAP>
__global__ __launch_bounds__(256) void select_candidates_kernel(const CandArgs c, int32_t *__restrict__ idx_out,
                                                               float *__restrict__ score_out) {
    __shared__ uint32_t key[LCAP];
    __shared__ int32_t item[LCAP];
    float *__restrict__ cscore = c.score;
    int32_t *__restrict__ citem = c.item;
    int32_t *__restrict__ count = c.count;
    const int cap = c.cap, k = c.k;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    // A user whose candidates did not fit the row in some range (count > cap: e.g. thousands of excluded items that
    // all score above the threshold) is given up HERE, alone: its threshold becomes +inf, so that later ranges add
    // nothing to it, and its output row is item -1 throughout - the caller redoes such users by the two-call path.
    const bool dead = count[row] > cap || citem[row * cap] < 0;
    if (LCAP == CAND_SMALL ? (!dead && count[row] > CAND_SMALL) : (dead || count[row] <= CAND_SMALL)) return;
    if (dead) {
        for (int q = tid; q < k; q += 256) {
            if (idx_out) {
                idx_out[row * k + q] = -1;
                score_out[row * k + q] = -__builtin_inff();
            } else {
                citem[row * cap + q] = -1;
                cscore[row * cap + q] = __builtin_inff();
            }
        }
        if (tid == 0) count[row] = k;
        return;
    }
    const int n = min(count[row], cap);
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    // Slots [0, k) hold the best of the earlier ranges, exclusions applied.  The score kernel appends every score
    // above the threshold; a user's excluded items among them (at most as many as it has) are dropped here, one
    // binary search per thread, instead of behind every hit inside the matrix-core loop.
    for (int q = tid; q < n2; q += 256) {
        bool ok = q < n;
        const int32_t it = ok ? citem[row * cap + q] : INT32_MAX;
        if (ok && q >= k && excluded_item(c, row, it)) ok = false;
        key[q] = ok ? order_key(cscore[row * cap + q]) : 0u;
        item[q] = ok ? it : INT32_MAX;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = tid; q < n2 / 2; q += 256) {
                const int lo = 2 * q - (q & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint32_t ka = key[lo], kb = key[hi];
                const int32_t ia = item[lo], ib = item[hi];
                const bool a_first = (ka > kb) || (ka == kb && ia < ib);
                if (a_first != desc) {
                    key[lo] = kb; key[hi] = ka;
                    item[lo] = ib; item[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    if (idx_out) {
        for (int q = tid; q < k; q += 256) {
            idx_out[row * k + q] = item[q];
            score_out[row * k + q] = key_to_float(key[q]);
        }
    } else {
        for (int q = tid; q < k; q += 256) {
            citem[row * cap + q] = item[q];
            cscore[row * cap + q] = key_to_float(key[q]);
        }
        if (tid == 0) count[row] = k;
    }
}

__global__ void fill_i32_kernel(int32_t *__restrict__ p, const int64_t n, const int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
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

// Launches the bf16-split score kernel if the operands allow it (16-byte vectors, f <= 128).
// Returns false if they do not; the fp32 kernel then takes the call.
template <bool FILTER>
static bool launch_scores_split(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y,
                                int64_t ldy, int64_t M, int32_t f, float *S, int64_t lds, int64_t item_begin,
                                const CandArgs &cand, hipStream_t st) {
    const bool vec = (f % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                     ((uintptr_t)Y % 16 == 0);
    if (!vec || f > 128 || M <= item_begin) return false;
    const int64_t n_stages = (M - item_begin + SB_ITEMS - 1) / SB_ITEMS;
    const int64_t nub = (nu + SB_USERS - 1) / SB_USERS;
    // enough workgroups to fill the chip a few times over, at least 16 stages each (a workgroup
    // first loads and splits its 128 user rows); the chunk count is a multiple of the 8 XCDs
    // (8 consecutive workgroup ids = 8 different chunks); few items: one chunk, no XCD layout needed
    int64_t n_chunks = std::max<int64_t>(8, (4096 / nub + 7) / 8 * 8);
    n_chunks = std::min(n_chunks, ((n_stages + 15) / 16 + 7) / 8 * 8);
    if (n_stages <= 16) n_chunks = 1;
    const int cpr = n_chunks >= 8 ? 8 : 1;
    const int64_t per_chunk = (n_stages + n_chunks - 1) / n_chunks;
    if (n_chunks * nub > 0x7fffffffLL) return false;
    const dim3 grid((unsigned)(n_chunks * nub)), block(256);
    const int ks = (f + 31) / 32;
#define ALS_SCORES_SPLIT_CASE(n)                                                                                       \
    if (ks == n)                                                                                                       \
        hipLaunchKernelGGL((scores_split_kernel<n, FILTER>), grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, M, f, S, lds, \
                           item_begin, per_chunk, nub, cpr, cand);
    ALS_SCORES_SPLIT_CASE(1) ALS_SCORES_SPLIT_CASE(2) ALS_SCORES_SPLIT_CASE(3) ALS_SCORES_SPLIT_CASE(4)
#undef ALS_SCORES_SPLIT_CASE
    return true;
}

// The prepared-table score kernel over items [item_begin, M) (als_scores_prepared.inc).
static size_t prepared_bytes(int64_t M, int32_t f) {
    if (M < 1 || f < 1 || f > 128 || f % 4) return 0;
    const int64_t ks = (f + 31) / 32, stages = (M + 31) / 32;
    return PREP_HEADER + (size_t)stages * ks * 4096;
}
template <bool FILTER>
static int32_t launch_scores_prepared(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const void *P,
                                      int64_t M_table, int64_t M, int32_t f, float *S, int64_t lds, int64_t item_begin,
                                      const CandArgs &cand, hipStream_t st) {
    if (M <= item_begin) return ALS_OK;
    const int64_t n_stages = (M - item_begin + 31) / 32;
    const int64_t nub = (nu + PB_USERS - 1) / PB_USERS;
    // as launch_scores_split, but about two workgroups per resident slot (512) only: a workgroup's first act is to
    // load and split its 256 user rows, about as much as 6 stages cost; at least 16 stages each; the chunk count a
    // multiple of the 8 XCDs so that the workgroups streaming the same items share an L2
    int64_t n_chunks = std::max<int64_t>(8, (1024 / nub + 7) / 8 * 8);
    n_chunks = std::min(n_chunks, ((n_stages + 15) / 16 + 7) / 8 * 8);
    if (n_stages <= 16) n_chunks = 1;
    const int cpr = n_chunks >= 8 ? 8 : 1;
    const int64_t per_chunk = (n_stages + n_chunks - 1) / n_chunks;
    if (n_chunks * nub > 0x7fffffffLL) return als_fail_(ALS_ERR_UNSUPPORTED, "too many users for one prepared-score launch");
    const dim3 grid((unsigned)(n_chunks * nub)), block(256);
    const int ks = (f + 31) / 32;
    const int64_t table_stages = (M_table + 31) / 32;
    const char *Pc = (const char *)P;
#define ALS_SCORES_PREP_CASE(n)                                                                                          \
    if (ks == n)                                                                                                         \
        hipLaunchKernelGGL((scores_prepared_kernel<n, FILTER>), grid, block, 0, st, X, ldx, user_ids, nu, Pc, M, f, S, lds, \
                           item_begin, per_chunk, nub, cpr, table_stages, cand);
    ALS_SCORES_PREP_CASE(1) ALS_SCORES_PREP_CASE(2) ALS_SCORES_PREP_CASE(3) ALS_SCORES_PREP_CASE(4)
#undef ALS_SCORES_PREP_CASE
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}
static int32_t check_prepared_args(const char *who, const float *X, int64_t ldx, int64_t nu, const void *P, int64_t M,
                                   int32_t f) {
    if (!X || !P) return als_fail_(ALS_ERR_INVALID_ARG, "%s: null pointer", who);
    if (nu < 1 || M < 1 || f < 1 || ldx < f) return als_fail_(ALS_ERR_INVALID_ARG, "%s: bad sizes (nu=%lld M=%lld f=%d)", who,
                                                            (long long)nu, (long long)M, f);
    if (prepared_bytes(M, f) == 0) return als_fail_(ALS_ERR_UNSUPPORTED, "%s: prepared items need f <= 128, f %% 4 == 0", who);
    if (ldx % 4 || (uintptr_t)X % 16 || (uintptr_t)P % 16)
        return als_fail_(ALS_ERR_UNSUPPORTED, "%s: user rows and the table must be 16-byte aligned (ldx a multiple of 4)", who);
    return ALS_OK;
}

extern "C" {

size_t als_items_prepared_bytes(int64_t M, int32_t f) { return prepared_bytes(M, f); }

int32_t als_items_prepare_f32(const float *Y, int64_t ldy, int64_t M, int32_t f, void *prepared, size_t bytes,
                              void *stream) {
    if (!Y || !prepared) return als_fail_(ALS_ERR_INVALID_ARG, "als_items_prepare_f32: null pointer");
    if (M < 1 || f < 1 || ldy < f) return als_fail_(ALS_ERR_INVALID_ARG, "als_items_prepare_f32: bad sizes");
    const size_t need = prepared_bytes(M, f);
    if (need == 0) return als_fail_(ALS_ERR_UNSUPPORTED, "als_items_prepare_f32: f=%d (needs f <= 128, f %% 4 == 0)", f);
    if (ldy % 4 || (uintptr_t)Y % 16 || (uintptr_t)prepared % 16)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_items_prepare_f32: rows and the table must be 16-byte aligned");
    if (bytes < need) return als_fail_(ALS_ERR_WORKSPACE, "als_items_prepare_f32: table too small: %zu < %zu", bytes, need);
    hipStream_t st = (hipStream_t)stream;
    ALS_HIP_CHECK2(hipMemsetAsync(prepared, 0, PREP_HEADER, st));
    const int64_t quads = M * (f / 4);
    hipLaunchKernelGGL(items_absmax_kernel, dim3((unsigned)std::min<int64_t>(2048, (quads + 255) / 256)), dim3(256), 0, st, Y,
                       ldy, M, f, (unsigned *)prepared);
    ALS_HIP_CHECK2(hipGetLastError());
    const int ks = (f + 31) / 32;
    const int64_t n_tiles = (M + 31) / 32 * 2, threads = n_tiles * ks * 64;
    if ((threads + 255) / 256 > 0x7fffffffLL) return als_fail_(ALS_ERR_UNSUPPORTED, "als_items_prepare_f32: too many items");
    hipLaunchKernelGGL(items_prepare_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, Y, ldy, M, f, ks,
                       n_tiles, (char *)prepared);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

int32_t als_scores_prepared_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const void *prepared,
                                int64_t M_table, int64_t M, int32_t f, float *scores, int64_t lds, void *stream) {
    int32_t rc = check_prepared_args("als_scores_prepared_f32", X, ldx, nu, prepared, M_table, f);
    if (rc) return rc;
    if (!scores || M < 1 || M > M_table || lds < M) return als_fail_(ALS_ERR_INVALID_ARG, "als_scores_prepared_f32: bad output sizes");
    const CandArgs none{};
    return launch_scores_prepared<false>(X, ldx, user_ids, nu, prepared, M_table, M, f, scores, lds, 0, none, (hipStream_t)stream);
}

int32_t als_scores_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y, int64_t ldy,
                       int64_t M, int32_t f, float *scores, int64_t lds, void *stream) {
    if (!X || !Y || !scores) return als_fail_(ALS_ERR_INVALID_ARG, "als_scores_f32: null pointer");
    if (nu < 1 || M < 1 || f < 1 || ldx < f || ldy < f || lds < M)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_scores_f32: bad sizes (nu=%lld M=%lld f=%d)", (long long)nu,
                         (long long)M, f);
    hipStream_t st = (hipStream_t)stream;
    const CandArgs none{};
    if (launch_scores_split<false>(X, ldx, user_ids, nu, Y, ldy, M, f, scores, lds, 0, none, st)) {
        ALS_HIP_CHECK2(hipGetLastError());
        return ALS_OK;
    }
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)((nu + 63) / 64)), block(256);
    if (grid.y > 65535) return als_fail_(ALS_ERR_UNSUPPORTED, "als_scores_f32: more than 65535*64 users per call");
    const bool vec = (f % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                     ((uintptr_t)Y % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((scores_kernel<true, false>), grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, M, f, scores, lds,
                           (int64_t)0, none);
    else
        hipLaunchKernelGGL((scores_kernel<false, false>), grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, M, f, scores,
                           lds, (int64_t)0, none);
    ALS_HIP_CHECK2(hipGetLastError());
    return ALS_OK;
}

// ----------------------------------------------------------------------------- fused top-k
// The fused top-k walks the items in ranges [S, 4S), [4S, 16S), ...: the first S items are scored
// into memory and selected exactly; every later range is filtered against the user's k-th best so
// far and the candidate rows are cut back to their k best between ranges.  A range is expected to
// add k * (range length) / (items before it) = 3k candidates per user, far below CAND_CAP.
constexpr int FUSED_RANGE_GROWTH = 4;
static int64_t fused_subset_items(int64_t M, int32_t k) {
    const int64_t s = std::max<int64_t>(16384, ((int64_t)k * 16 + 255) / 256 * 256);
    return std::min(M, s);
}
static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t als_recommend_topk_workspace_bytes(int64_t nu, int64_t M, int32_t k) {
    if (nu < 1 || M < 1 || k < 1 || k > TOPK_MAX || k > M) return 0;
    const int64_t S = fused_subset_items(M, k);
    size_t b = align256((size_t)nu * S * sizeof(float));                  // scores of the early items
    if (S < M) b += 2 * align256((size_t)nu * CAND_CAP * 4) + align256((size_t)nu * 4);
    return b + 256;
}

// Y (with ldy) or a prepared table P of the M items: the same steps either way
static int32_t recommend_topk_impl(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y,
                                   int64_t ldy, const void *P, int64_t M, int32_t f, int32_t k, const int64_t *ex_rowptr,
                                   const int32_t *ex_col, int32_t ex_by_user_id, int32_t *idx_out, float *score_out,
                                   int32_t *overflow_out, void *workspace, size_t ws_bytes, void *stream) {
    if (!X || (!Y && !P) || !idx_out || !score_out || !overflow_out || !workspace)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_recommend_topk_f32: null pointer");
    if (nu < 1 || M < 1 || M > INT32_MAX || f < 1 || ldx < f || (Y && ldy < f))
        return als_fail_(ALS_ERR_INVALID_ARG, "als_recommend_topk_f32: bad sizes (nu=%lld M=%lld f=%d)", (long long)nu,
                         (long long)M, f);
    if (k < 1 || k > TOPK_MAX || k > M)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_recommend_topk_f32: k=%d outside [1, min(%d, M)]", k, TOPK_MAX);
    if ((ex_rowptr == nullptr) != (ex_col == nullptr))
        return als_fail_(ALS_ERR_INVALID_ARG, "als_recommend_topk_f32: pass both exclusion arrays or neither");
    if (nu > 65535LL * 64) return als_fail_(ALS_ERR_UNSUPPORTED, "als_recommend_topk_f32: more than 65535*64 users per call");
    const size_t need = als_recommend_topk_workspace_bytes(nu, M, k);
    if (ws_bytes < need) return als_fail_(ALS_ERR_WORKSPACE, "top-k workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    // exclusion CSR indexed by user id: output row u reads CSR row user_ids[u] (no user_ids: u, as before)
    const int64_t *ex_rows = (ex_rowptr && ex_by_user_id) ? user_ids : nullptr;
    const int64_t S = fused_subset_items(M, k);
    char *ws = (char *)workspace;
    float *sub = (float *)ws; ws += align256((size_t)nu * S * sizeof(float));
    // 1. the early items, materialised: scores, exclusions, exact top-k
    int32_t rc = P ? als_scores_prepared_f32(X, ldx, user_ids, nu, P, M, S, f, sub, S, stream)
                   : als_scores_f32(X, ldx, user_ids, nu, Y, ldy, S, f, sub, S, stream);
    if (rc) return rc;
    if (S == M) {
        ALS_HIP_CHECK2(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), st));
        return als_topk_rows_f32(sub, S, nu, S, ex_rowptr, ex_col, ex_rows, k, idx_out, score_out, stream);
    }
    CandArgs c;
    c.score = (float *)ws; ws += align256((size_t)nu * CAND_CAP * 4);
    c.item = (int32_t *)ws; ws += align256((size_t)nu * CAND_CAP * 4);
    c.count = (int32_t *)ws;
    c.overflow = overflow_out; c.ex_rowptr = ex_rowptr; c.ex_col = ex_col; c.ex_rows = ex_rows; c.cap = CAND_CAP; c.k = k;
    ALS_HIP_CHECK2(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, c.count, nu, k);
    ALS_HIP_CHECK2(hipGetLastError());
    if (S <= TOPK_REG_ITEMS)
        hipLaunchKernelGGL(topk_rows_reg_kernel, dim3((unsigned)nu), dim3(256), 0, st, sub, S, S, ex_rowptr, ex_col, ex_rows,
                           k, c.item, c.score, (int64_t)CAND_CAP);
    else
        hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)nu), dim3(256), 0, st, sub, S, S, ex_rowptr, ex_col,
                           ex_rows, k, c.item, c.score, (int64_t)CAND_CAP);
    ALS_HIP_CHECK2(hipGetLastError());
    // 2. the later items, range by range: score tiles are compared with the user's threshold in
    //    registers, never stored; then each user's candidates are ordered and cut back to k
    const bool vec = (f % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                     ((uintptr_t)Y % 16 == 0);
    for (int64_t begin = S; begin < M;) {
        int64_t end = (begin > M / FUSED_RANGE_GROWTH) ? M : begin * FUSED_RANGE_GROWTH;
        if (M - end < end / FUSED_RANGE_GROWTH) end = M;      // no short last range
        const dim3 grid((unsigned)((end - begin + 255) / 256), (unsigned)((nu + 63) / 64)), block(256);
        if (P) {
            rc = launch_scores_prepared<true>(X, ldx, user_ids, nu, P, M, end, f, nullptr, 0, begin, c, st);
            if (rc) return rc;
        } else if (launch_scores_split<true>(X, ldx, user_ids, nu, Y, ldy, end, f, nullptr, 0, begin, c, st)) {
        } else if (vec)
            hipLaunchKernelGGL((scores_kernel<true, true>), grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, end, f, nullptr,
                               (int64_t)0, begin, c);
        else
            hipLaunchKernelGGL((scores_kernel<false, true>), grid, block, 0, st, X, ldx, user_ids, nu, Y, ldy, end, f, nullptr,
                               (int64_t)0, begin, c);
        ALS_HIP_CHECK2(hipGetLastError());
        const bool last = end == M;
        hipLaunchKernelGGL(select_candidates_kernel<CAND_SMALL>, dim3((unsigned)nu), dim3(256), 0, st, c,
                           last ? idx_out : nullptr, last ? score_out : nullptr);
        hipLaunchKernelGGL(select_candidates_kernel<CAND_CAP>, dim3((unsigned)nu), dim3(256), 0, st, c,
                           last ? idx_out : nullptr, last ? score_out : nullptr);
        ALS_HIP_CHECK2(hipGetLastError());
        begin = end;
    }
    return ALS_OK;
}

int32_t als_recommend_topk_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y,
                               int64_t ldy, int64_t M, int32_t f, int32_t k, const int64_t *ex_rowptr,
                               const int32_t *ex_col, int32_t ex_by_user_id, int32_t *idx_out, float *score_out,
                               int32_t *overflow_out, void *workspace, size_t ws_bytes, void *stream) {
    if (!Y) return als_fail_(ALS_ERR_INVALID_ARG, "als_recommend_topk_f32: null pointer");
    return recommend_topk_impl(X, ldx, user_ids, nu, Y, ldy, nullptr, M, f, k, ex_rowptr, ex_col, ex_by_user_id, idx_out,
                               score_out, overflow_out, workspace, ws_bytes, stream);
}

int32_t als_recommend_topk_prepared_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu,
                                        const void *prepared, int64_t M, int32_t f, int32_t k, const int64_t *ex_rowptr,
                                        const int32_t *ex_col, int32_t ex_by_user_id, int32_t *idx_out, float *score_out,
                                        int32_t *overflow_out, void *workspace, size_t ws_bytes, void *stream) {
    int32_t rc = check_prepared_args("als_recommend_topk_prepared_f32", X, ldx, nu, prepared, M, f);
    if (rc) return rc;
    return recommend_topk_impl(X, ldx, user_ids, nu, nullptr, 0, prepared, M, f, k, ex_rowptr, ex_col, ex_by_user_id,
                               idx_out, score_out, overflow_out, workspace, ws_bytes, stream);
}

int32_t als_topk_rows_f32(float *scores, int64_t lds, int64_t n_rows, int64_t M, const int64_t *ex_rowptr,
                          const int32_t *ex_col, const int64_t *ex_row_ids, int32_t k, int32_t *idx_out, float *score_out,
                          void *stream) {
    if (!scores || !idx_out || !score_out) return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: null pointer");
    if (n_rows < 1 || M < 1 || lds < M || M > INT32_MAX)
        return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: bad sizes");
    if (k < 1 || k > TOPK_MAX || k > M)
        return als_fail_(ALS_ERR_UNSUPPORTED, "als_topk_rows_f32: k=%d outside [1, min(%d, M)]", k, TOPK_MAX);
    if ((ex_rowptr == nullptr) != (ex_col == nullptr))
        return als_fail_(ALS_ERR_INVALID_ARG, "als_topk_rows_f32: pass both exclusion arrays or neither");
    if (M <= TOPK_REG_ITEMS)
        hipLaunchKernelGGL(topk_rows_reg_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, lds, M,
                           ex_rowptr, ex_col, ex_rowptr ? ex_row_ids : nullptr, k, idx_out, score_out, (int64_t)k);
    else
        hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, lds, M,
                           ex_rowptr, ex_col, ex_rowptr ? ex_row_ids : nullptr, k, idx_out, score_out, (int64_t)k);
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
