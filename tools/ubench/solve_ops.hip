// Micro-benchmark: cost of the instruction kinds on the per-row solve's dependent chains, one wave per SIMD
// (what the f = 128 kernels run at) and three (f = 64).  Every loop body is inline assembly, so the instruction
// stream is exactly what is written; times are per loop iteration, and in units of a dependent v_fma_f32.
//   hipcc --offload-arch=gfx950 -O2 -o solve_ops solve_ops.hip && ./solve_ops
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 4000

template <int MODE>
__global__ __launch_bounds__(768) void k(float *out, float s) {
    float a = threadIdx.x * 0.001f + 1.0f, b = s, c = s * 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
    float g0 = a + 4.f, g1 = a + 5.f, g2 = a + 6.f, g3 = a + 7.f;
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    bf16x4 pa = {1, 2, 3, 4};
    bf16x8 pb = {1, 2, 3, 4, 5, 6, 7, 8};
    f32x4 accs[8];
    for (int q = 0; q < 8; ++q) accs[q] = f32x4{0, 0, 0, 0};
    int addr = ((threadIdx.x & 15) + 16) * 4;
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE == 0) {          // 8 dependent fma
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 1) {   // 8 independent fmac_dpp row_newbcast (4 registers, twice)
            asm volatile("v_fmac_f32_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                         : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(c));
        } else if constexpr (MODE == 2) {   // 8 independent plain fmac
            asm volatile("v_fmac_f32 %0, %0, %4\n v_fmac_f32 %1, %1, %4\n v_fmac_f32 %2, %2, %4\n v_fmac_f32 %3, %3, %4\n"
                         "v_fmac_f32 %0, %0, %4\n v_fmac_f32 %1, %1, %4\n v_fmac_f32 %2, %2, %4\n v_fmac_f32 %3, %3, %4"
                         : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(c));
        } else if constexpr (MODE == 3) {   // 4 dependent rcp
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0" : "+v"(a));
        } else if constexpr (MODE == 4) {   // 4 x (readlane -> dependent mul by the sgpr)
            asm volatile("v_readlane_b32 s20, %0, 5\n v_mul_f32 %0, s20, %0\n v_readlane_b32 s20, %0, 6\n v_mul_f32 %0, s20, %0\n"
                         "v_readlane_b32 s20, %0, 7\n v_mul_f32 %0, s20, %0\n v_readlane_b32 s20, %0, 8\n v_mul_f32 %0, s20, %0" : "+v"(a) :: "s20");
        } else if constexpr (MODE == 5) {   // 4 x (mov, mov, swap32, mov, swap16): a lane group into all four
            for (int q = 0; q < 4; ++q)
                asm volatile("v_mov_b32 %1, %0\n v_mov_b32 %2, %0\n s_nop 1\n v_permlane32_swap_b32 %1, %2\n v_mov_b32 %2, %1\n s_nop 1\n v_permlane16_swap_b32 %1, %2\n s_nop 1\n v_add_f32 %0, %1, %2"
                             : "+v"(a), "+v"(d), "+v"(e));
        } else if constexpr (MODE == 6) {   // 4 dependent ds_bpermute round trips
            for (int q = 0; q < 4; ++q)
                asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr));
        } else if constexpr (MODE == 7) {   // 8 dependent accvgpr write/read pairs
            for (int q = 0; q < 8; ++q)
                asm volatile("v_accvgpr_write_b32 a0, %0\n s_nop 1\n v_accvgpr_read_b32 %0, a0" : "+v"(a) :: "a0");
        } else if constexpr (MODE == 8) {   // 8 dependent bf16 16x16x16 MFMAs (same accumulator)
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pa, acc, 0, 0, 0);
        } else if constexpr (MODE == 9) {   // 8 fp32 16x16x4 MFMAs (same accumulator)
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, acc, 0, 0, 0);
        } else if constexpr (MODE == 10) {  // 8 bf16 MFMAs to alternating accumulators + 16 independent fma in between
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pa, acc, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(g0), "+v"(g1) : "v"(b), "v"(c));
                acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pa, acc2, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(g2), "+v"(g3) : "v"(b), "v"(c));
            }
        } else if constexpr (MODE == 11) {  // elimination step as it was: bpermute + readlane/rcp/newton + 8 fmac_dpp
            asm volatile("ds_bpermute_b32 %5, %6, %0\n v_readlane_b32 s20, %0, 21\n v_rcp_f32 %4, s20\n s_nop 0\n v_fma_f32 %7, -s20, %4, 2.0\n v_mul_f32 %4, %4, -%7\n"
                         "s_waitcnt lgkmcnt(0)\n v_mul_f32 %4, %4, %5\n v_cndmask_b32 %4, 0, %4, vcc\n"
                         "v_fmac_f32_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %8, %8, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %9, %9, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %10, %10, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %11, %11, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                         : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(d), "+v"(e), "+v"(addr), "+v"(f), "+v"(a), "+v"(b), "+v"(c), "+v"(acc2[0]) :: "s20", "vcc");
        } else if constexpr (MODE == 12) {  // merged step with swaps: readlane/rcp + movs/swaps + mul/cndmask + 4 fmac_dpp
            asm volatile("v_readlane_b32 s20, %0, 21\n v_mov_b32 %5, %0\n v_mov_b32 %6, %0\n v_rcp_f32 %4, -s20\n s_nop 0\n v_permlane32_swap_b32 %5, %6\n v_mov_b32 %6, %5\n s_nop 1\n"
                         "v_permlane16_swap_b32 %5, %6\n v_mul_f32 %4, %4, %6\n v_cndmask_b32 %4, 0, %4, vcc\n"
                         "v_fmac_f32_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                         : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(d), "+v"(e), "+v"(f) :: "s20", "vcc");
        } else if constexpr (MODE == 13) {  // merged step with bpermute: bpermute + readlane/rcp + mul/cndmask + 4 fmac_dpp
            asm volatile("ds_bpermute_b32 %5, %6, %0\n v_readlane_b32 s20, %0, 21\n v_rcp_f32 %4, -s20\n"
                         "s_waitcnt lgkmcnt(0)\n v_mul_f32 %4, %4, %5\n v_cndmask_b32 %4, 0, %4, vcc\n"
                         "v_fmac_f32_dpp %0, %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                         : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(d), "+v"(e), "+v"(addr) :: "s20", "vcc");
        } else if constexpr (MODE == 15) {  // 8 independent bf16 16x16x16 MFMAs (8 accumulators)
#pragma unroll
            for (int q = 0; q < 8; ++q) accs[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pa, accs[q], 0, 0, 0);
        } else if constexpr (MODE == 16) {  // 8 independent bf16 16x16x32 MFMAs (8 accumulators)
#pragma unroll
            for (int q = 0; q < 8; ++q) accs[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb, pb, accs[q], 0, 0, 0);
        } else if constexpr (MODE == 17) {  // 8 independent bf16 16x16x32 MFMAs + 16 independent fma interleaved
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                accs[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb, pb, accs[q], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(g0), "+v"(g1) : "v"(b), "v"(c));
            }
        } else if constexpr (MODE == 18) {  // 8 independent bf16 16x16x16 MFMAs + 16 independent fma interleaved
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                accs[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pa, accs[q], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(g0), "+v"(g1) : "v"(b), "v"(c));
            }
        } else if constexpr (MODE == 19) {  // 8 x (s_mov + s_nop): scalar instructions in the one wave's stream
            asm volatile("s_mov_b32 s20, 1\n s_nop 0\n s_mov_b32 s20, 2\n s_nop 0\n s_mov_b32 s20, 3\n s_nop 0\n s_mov_b32 s20, 4\n s_nop 0\n"
                         "s_mov_b32 s20, 1\n s_nop 0\n s_mov_b32 s20, 2\n s_nop 0\n s_mov_b32 s20, 3\n s_nop 0\n s_mov_b32 s20, 4\n s_nop 0" ::: "s20");
        } else if constexpr (MODE == 20) {  // 4 x (2 dot2c bf16), dependent
            asm volatile("v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2\n"
                         "v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 14) {  // 4 fmac_dpp only, dependent through the DPP source (2 registers alternating)
            asm volatile("v_fmac_f32_dpp %0, %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %1, %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                         "v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %1, %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_nop 1"
                         : "+v"(g0), "+v"(g1) : "v"(c));
        }
    }
    for (int q = 0; q < 8; ++q) a += accs[q][0] + accs[q][3];
    float r = a + d + e + f + g0 + g1 + g2 + g3 + b + c + acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + (float)addr;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, int threads, float *d, double &unit) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ITERS;
    if (MODE == 0 && threads == 256) unit = ns / 8.0;
    printf("%-64s waves/SIMD=%d  %8.1f ns / iteration = %6.1f dependent-fma times\n", name, threads / 256, ns, ns / unit);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    double unit = 1.0;
    for (int threads : {256, 768}) {
        run<0>("8 dependent fma", threads, d, unit);
        run<2>("8 independent fmac", threads, d, unit);
        run<1>("8 independent fmac_dpp row_newbcast", threads, d, unit);
        run<14>("4 fmac_dpp dependent through the DPP source (+s_nop 1)", threads, d, unit);
        run<3>("4 dependent rcp", threads, d, unit);
        run<4>("4 x (readlane -> mul by the sgpr)", threads, d, unit);
        run<5>("4 x (2 mov, swap32, mov, swap16, add)", threads, d, unit);
        run<6>("4 dependent ds_bpermute round trips", threads, d, unit);
        run<7>("8 dependent accvgpr write/read pairs", threads, d, unit);
        run<8>("8 dependent bf16 16x16x16 MFMA", threads, d, unit);
        run<9>("8 dependent fp32 16x16x4 MFMA", threads, d, unit);
        run<10>("8 bf16 16x16x16 MFMA (2 accumulators) + 16 fma interleaved", threads, d, unit);
        run<15>("8 independent bf16 16x16x16 MFMA", threads, d, unit);
        run<18>("8 independent bf16 16x16x16 MFMA + 16 fma interleaved", threads, d, unit);
        run<16>("8 independent bf16 16x16x32 MFMA", threads, d, unit);
        run<17>("8 independent bf16 16x16x32 MFMA + 16 fma interleaved", threads, d, unit);
        run<19>("8 x (s_mov, s_nop 0)", threads, d, unit);
        run<20>("8 dependent v_dot2c_f32_bf16", threads, d, unit);
        run<11>("elimination step, two tiles: bpermute, newton, 8 fmac_dpp", threads, d, unit);
        run<13>("elimination step, one tile: bpermute, 4 fmac_dpp", threads, d, unit);
        run<12>("elimination step, one tile: swaps, 4 fmac_dpp", threads, d, unit);
    }
    return 0;
}
