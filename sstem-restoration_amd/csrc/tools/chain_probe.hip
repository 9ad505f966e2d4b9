// Developer probe: throughput of v_mfma_f32_4x4x1 as a function of the number of independent accumulator
// chains per wave (1, 2, 3) and of waves per SIMD (1..4), operands in registers, 54-step dependent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int NCH>
__global__ __launch_bounds__(1024) void chains(float* out, int iters, float seed)
{
    float hs[54];
#pragma unroll
    for (int t = 0; t < 54; ++t) hs[t] = seed + t + threadIdx.x;
    float a[NCH];
    for (int c = 0; c < NCH; ++c) a[c] = seed * (c + 1);
    float o = 0.f;
    for (int it = 0; it < iters; ++it) {
        f32x4 acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 54; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[c], hs[t], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) o += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}

template <int NCH>
static void run(float* dout, int waves_per_simd)
{
    const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = 3000 / NCH;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chains<NCH>, dim3(blocks), dim3(threads), 0, 0, dout, 3, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chains<NCH>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)waves_per_simd * iters * 54.0 * NCH;
    printf("chains/wave %d  waves/SIMD %d : %.2f ns per MFMA per SIMD\n", NCH, waves_per_simd, ms * 1e6 / per_simd);
}

int main()
{
    float* dout; CK(hipMalloc(&dout, 256 * 1024 * 4));
    for (int w = 1; w <= 4; ++w) { run<1>(dout, w); run<2>(dout, w); run<3>(dout, w); }
    return 0;
}
