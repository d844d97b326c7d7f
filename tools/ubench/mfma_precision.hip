// Accuracy of the matrix-core accumulate on gfx950: C = A B^T (16 x 16, K = 32 per instruction, `reps` instructions
// chained into one accumulator) from fp16 / bf16 inputs against a float64 sum of the same products, with an
// accumulator start value `c0` that is large against the products (as a row's running sum is).
// Prints the largest error relative to the largest |C|.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__global__ void k(const unsigned short *A, const unsigned short *B, float *C, int reps, float c0) {
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    f32x4 acc{c0, c0, c0, c0};
    for (int r = 0; r < reps; ++r) {
        unsigned short a[8], b[8];
        for (int q = 0; q < 8; ++q) {      // lane (g, i): row i, k-slots 8 g + q of instruction r
            a[q] = A[(r * 16 + i) * 32 + 8 * g + q];
            b[q] = B[(r * 16 + i) * 32 + 8 * g + q];
        }
        if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(f16x8 *)a, *(f16x8 *)b, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8 *)a, *(bf16x8 *)b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = acc[r];
}

static float half_to_float(unsigned short h) { _Float16 x; __builtin_memcpy(&x, &h, 2); return (float)x; }
static unsigned short float_to_half(float f) { _Float16 x = (_Float16)f; unsigned short h; __builtin_memcpy(&h, &x, 2); return h; }
static float bf_to_float(unsigned short h) { unsigned u = (unsigned)h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
static unsigned short float_to_bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main() {
    const int reps = 8;
    for (int f16 = 0; f16 < 2; ++f16)
        for (float amp : {1.f, 200.f, 0.05f})
            for (float c0 : {0.f, 1e4f}) {
                std::vector<unsigned short> A(reps * 16 * 32), B(reps * 16 * 32);
                srand(3);
                for (auto *v : {&A, &B})
                    for (auto &x : *v) {
                        const float r = amp * ((rand() % 2001) / 1000.f - 1.f);
                        x = f16 ? float_to_half(r) : float_to_bf(r);
                    }
                unsigned short *dA, *dB; float *dC;
                hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 256 * 4);
                hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
                hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
                if (f16) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, dA, dB, dC, reps, c0 * amp * amp);
                else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, dA, dB, dC, reps, c0 * amp * amp);
                float C[256];
                hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
                double worst = 0, big = 0;
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) {
                        double s = (double)(c0 * amp * amp);
                        for (int r = 0; r < reps; ++r)
                            for (int kk = 0; kk < 32; ++kk) {
                                const double a = f16 ? half_to_float(A[(r * 16 + i) * 32 + kk]) : bf_to_float(A[(r * 16 + i) * 32 + kk]);
                                const double b = f16 ? half_to_float(B[(r * 16 + j) * 32 + kk]) : bf_to_float(B[(r * 16 + j) * 32 + kk]);
                                s += a * b;
                            }
                        worst = fmax(worst, fabs(s - C[i * 16 + j]));
                        big = fmax(big, fabs(s));
                    }
                printf("%s amp %-6g c0 %-6g: max |err| / max |C| = %.3g  (2^%.1f)\n", f16 ? "fp16" : "bf16", amp, c0, worst / big, log2(worst / big));
                hipFree(dA); hipFree(dB); hipFree(dC);
            }
    return 0;
}
