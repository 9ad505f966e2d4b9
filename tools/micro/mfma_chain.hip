// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 as a function of how many independent accumulators the stream rotates over
// (1 = every MFMA reads the previous one's result as SrcC), with 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float seed)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int q = 0; q < 16; ++q) s += acc[k][q];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs_per_cu, float* out)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // occupancy through dynamic LDS: 1 workgroup per CU needs > 80 KB each, 2 per CU > 53 KB
    const int lds = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 60 * 1024 : 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(chain<NACC>, dim3(grid), dim3(256), lds, 0, out, 10, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NACC>, dim3(grid), dim3(256), lds, 0, out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * 24 * wgs_per_cu;
    printf("accumulators %d, waves per SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns), %.0f TFLOP/s\n", NACC,
           wgs_per_cu, ms, ms * 1e6 / mfmas_per_simd, mfmas_per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    float* out;
    hipMalloc(&out, 4096);
    for (int w = 1; w <= 2; ++w) { run<1>(w, out); run<2>(w, out); run<3>(w, out); run<4>(w, out); run<8>(w, out); }
    return 0;
}
