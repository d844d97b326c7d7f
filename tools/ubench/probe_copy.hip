// A stand-in for a collective's kernel, for tools/overlap_probe.py: a copy by a few resident workgroups with the
// resource footprint of an RCCL kernel (256 threads, 288 vector registers, 16 KB of LDS per workgroup) - unlike the
// runtime's own copy kernel, whose handful of registers fits beside anything.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libprobe_copy.so tools/ubench/probe_copy.hip
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                        const size_t n16, const int passes, const int keep_lds) {
    __shared__ uint4 pad[1024];
    // the kernel is allocated 288 vector registers (256 + 32 accumulation registers): one wave per SIMD, one workgroup per
    // compute unit - rcclGenericKernel of this ROCm's librccl.so is built for gfx950 with 256 threads, 261-280 registers
    // and 19.7 KB of LDS (read from the code object's metadata)
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a31, 0" ::: "v255", "a31");
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int pass = 0; pass < passes; ++pass) {             // (several passes: a kernel that stays resident that long)
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {     // four loads in flight per thread
            const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
            dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
        }
        for (; i < n16; i += stride) dst[i] = src[i];
    }
    if (keep_lds) {                                     // never taken: keeps the LDS allocation
        pad[threadIdx.x] = src[threadIdx.x];      // (threads <= 256)
        __syncthreads();
        dst[threadIdx.x] = pad[(threadIdx.x * 7) & 1023];
    }
}

extern "C" int probe_copy_launch(const void *src, void *dst, size_t bytes, int blocks, int threads, int passes, void *stream) {
    hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream, (const uint4 *)src,
                       (uint4 *)dst, bytes / 16, passes, 0);
    return (int)hipGetLastError();
}
