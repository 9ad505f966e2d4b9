// Developer probe: isolates the sepconv inner loop (A from an LDS ring by ds_read_b128, B from 54
// registers, 3 accumulator chains of v_mfma_f32_4x4x1_16b_f32) from everything else, at the
// kernel's occupancy (one workgroup per CU, WAVES waves), to see what the loop itself sustains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int WAVES, int USE_LDS, int DIST>
__global__ __launch_bounds__(WAVES * 64) void loop(float* out, int iters, const float* seed)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 144, CH = 3, RS = CH * P;
    for (int i = threadIdx.x; i < 80 * RS; i += WAVES * 64) lds[i] = seed[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane >> 2, sub = lane & 3;
    float hs[54];
#pragma unroll
    for (int t = 0; t < 54; ++t) hs[t] = seed[t + lane];
    const float* arow = lds + ((wave & 7) + sub) * RS + blk * 4;
    f32x4 ar[3][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { ar[0][c] = *(const f32x4*)(arow + c * P); ar[1][c] = *(const f32x4*)(arow + c * P + 4); ar[2][c] = ar[0][c]; }
    float o[CH] = {0, 0, 0};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        f32x4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0, 0, 0, 0};
        const float* abase = arow + (it % 13) * 4 * RS;
#pragma unroll
        for (int tq = 0; tq < 14; ++tq) {
            if (USE_LDS) {
                const float* src = abase + ((tq + DIST) % 14) * 4;
#pragma unroll
                for (int c = 0; c < CH; ++c) ar[(tq + DIST) % 3][c] = *(const f32x4*)(src + c * P);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = tq * 4 + e;
                if (t < 54) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[tq % 3][c][e], hs[t], acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = o[0] + o[1] + o[2];
}

template <int WAVES, int USE_LDS, int DIST>
static void run(float* dout, const float* dseed, const char* name)
{
    const int iters = 13 * 16, blocks = 256 * 4;
    const size_t ldsb = 80 * 432 * 4;
    auto k = loop<WAVES, USE_LDS, DIST>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), ldsb, 0, dout, 13, dseed);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), ldsb, 0, dout, iters, dseed);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)blocks * WAVES * iters * 162.0 / 1024.0;
    printf("%-44s %8.3f ms   %.2f ns per MFMA per SIMD\n", name, ms, ms * 1e6 / mfma_per_simd);
}

int main()
{
    float *dout, *dseed; CK(hipMalloc(&dout, 1024 * 768 * 4)); CK(hipMalloc(&dseed, 2048 * 4));
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = (float)rand() / RAND_MAX;
    CK(hipMemcpy(dseed, h, sizeof(h), hipMemcpyHostToDevice));
    run<8, 0, 2>(dout, dseed, "8 waves, A in registers (no LDS)");
    run<8, 1, 1>(dout, dseed, "8 waves, LDS ring, prefetch distance 1");
    run<8, 1, 2>(dout, dseed, "8 waves, LDS ring, prefetch distance 2");
    run<12, 0, 2>(dout, dseed, "12 waves, A in registers (no LDS)");
    run<12, 1, 2>(dout, dseed, "12 waves, LDS ring, prefetch distance 2");
    run<16, 0, 2>(dout, dseed, "16 waves, A in registers (no LDS)");
    run<16, 1, 2>(dout, dseed, "16 waves, LDS ring, prefetch distance 2");
    return 0;
}
