// Micro-benchmark: VALU / MFMA issue rates and co-execution on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 2000

template <int MODE>   // 0 fma, 1 pk_fma, 2 mfma16x16x4, 3 mixed-in-wave (10 mfma + 30 fma), 4 split waves (w<4 mfma, else fma), 5 split (w<4 mfma else pkfma)
__global__ __launch_bounds__(512) void k(float *out, float s) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a[16]; f32x2 p[16]; f32x4 acc[10];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float x = s, y = s * 0.5f;
    const f32x2 xs = f32x2{x, x}, ys = f32x2{y, y};
    bool do_mfma = (MODE == 2) || (MODE == 3) || ((MODE == 4 || MODE == 5) && wave < 4);
    bool do_fma = (MODE == 0) || (MODE == 3) || (MODE == 4 && wave >= 4);
    bool do_pk = (MODE == 1) || (MODE == 5 && wave >= 4);
    for (int it = 0; it < ITERS; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
        }
        if (do_fma) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 15; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
        if (do_pk) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 15; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xs), "v"(ys));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 10; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, int threads, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s threads=%3d  %.3f ms  -> %.1f ns per loop iteration (ITERS=%d)\n", name, threads, ms, ms * 1e6 / ITERS, ITERS);
}
int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    // one wave per SIMD (256 threads), two (512)
    run<0>("30 fma / iter", 256, d);  run<0>("30 fma / iter", 512, d);
    run<1>("30 pk_fma / iter", 256, d); run<1>("30 pk_fma / iter", 512, d);
    run<2>("10 mfma16x16x4 / iter", 256, d); run<2>("10 mfma16x16x4 / iter", 512, d);
    run<3>("10 mfma + 30 fma same wave", 256, d); run<3>("10 mfma + 30 fma same wave", 512, d);
    run<4>("split: w<4 10 mfma | w>=4 30 fma", 512, d);
    run<5>("split: w<4 10 mfma | w>=4 30 pkfma", 512, d);
    return 0;
}
