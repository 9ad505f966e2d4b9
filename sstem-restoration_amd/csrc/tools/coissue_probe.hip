// Developer probe: do v_mfma_f32_4x4x1 (8-cycle) instructions issued by SOME waves overlap with fp32 VALU
// FMAs issued by OTHER waves of the same SIMD?  (Inside one wave they serialise -- mfma_probe.)
// Workgroup = 16 waves (4 per SIMD): the first NM waves per SIMD run an MFMA stream, the rest a VALU FMA stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MFMA_WAVES>   // waves with (wave_in_wg < MFMA_WAVES) run MFMAs
__global__ __launch_bounds__(1024) void mix(float* out, int it_mfma, int it_valu, float seed)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r = 0.f;
    if (wave < MFMA_WAVES) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
        const float a = seed + threadIdx.x, b = seed * 0.5f;
        for (int i = 0; i < it_mfma; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, a2, 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) r += a0[i] + a1[i] + a2[i];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
        const float a = seed * 0.999f, b = 1e-3f;
        for (int i = 0; i < it_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 24; ++u) v[u % 8] = fmaf(v[u % 8], a, b);
        }
        for (int i = 0; i < 8; ++i) r += v[i];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int MW>
static float run(float* dout, int it_mfma, int it_valu)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mix<MW>, dim3(256), dim3(1024), 0, 0, dout, 10, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix<MW>, dim3(256 * 2), dim3(1024), 0, 0, dout, it_mfma, it_valu, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    float* dout; CK(hipMalloc(&dout, 512 * 1024 * 4));
    const int N = 4000;
    // 16 waves per WG, waves w -> SIMD w%4 (cyclic): the first 8 waves put 2 on every SIMD
    float t_m = run<8>(dout, N, 0);      // 8 MFMA waves busy, 8 idle
    float t_v = run<8>(dout, 0, N);      // 8 VALU waves busy
    float t_b = run<8>(dout, N, N);      // both
    printf("8 MFMA waves alone: %.3f ms | 8 VALU waves alone: %.3f ms | both together: %.3f ms  (sum %.3f, max %.3f)\n",
           t_m, t_v, t_b, t_m + t_v, t_m > t_v ? t_m : t_v);
    float t_m16 = run<16>(dout, N, 0);
    printf("16 MFMA waves: %.3f ms (same per-wave work: 2x the MFMAs)\n", t_m16);
    const double mf = 512.0 * 8 * N * 24 * 512;      // blocks * waves * iters * mfma/iter * flop
    const double vf = 512.0 * 8 * N * 24 * 128;
    printf("MFMA alone %.1f TF, VALU alone %.1f TF, together %.1f TF\n", mf / t_m / 1e9, vf / t_v / 1e9, (mf + vf) / t_b / 1e9);
    return 0;
}
