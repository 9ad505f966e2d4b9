// Micro-benchmark: SUSTAINED rate of the exact-fp32 matrix instructions under the socket's power cap, random fp32 operands, ~1 s per line:
// v_mfma_f32_4x4x1_16b_f32 (the sepconv kernels' instruction: 16 independent 4x4 outer products, 512 flop), v_mfma_f32_16x16x4_f32
// (2048 flop) and v_mfma_f32_32x32x2_f32 (4096 flop) -- which shape costs the least energy per flop?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_shape_power mfma_f32_shape_power.hip && ./mfma_f32_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline float rnd(uint32_t seed, bool zero)
{
    const uint32_t h = hash(seed);
    // random sign and mantissa, exponent 120..127 (|x| in 2^-7 .. 2)
    return zero ? 0.f : __builtin_bit_cast(float, (h & 0x807fffffu) | ((120u + ((h >> 23) & 7u)) << 23));
}

template <int KIND, bool ZERO>
__global__ __launch_bounds__(256) void stream(float* out, int iters)
{
    float a[8], b[8];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    for (int i = 0; i < 8; ++i) { a[i] = rnd(t * 16u + i, ZERO); b[i] = rnd(t * 16u + 8u + i, ZERO); }
    f32x4 c4[8]; f32x16 c16[4];
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) c4[k][q] = 0.f;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) c16[k][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if constexpr (KIND == 0) c4[u % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u % 8], b[(u / 8) % 8], c4[u % 8], 0, 0, 0);
            else if constexpr (KIND == 1) c4[u % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u % 8], b[(u / 8) % 8], c4[u % 8], 0, 0, 0);
            else c16[u % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % 8], b[(u / 8) % 8], c16[u % 4], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) s += c4[k][q];
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) s += c16[k][q];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND, bool ZERO>
static void run(const char* name, float* out, double flop_per_inst, int iters)
{
    const int lds = 60 * 1024, grid = 512;       // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stream<KIND, ZERO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((stream<KIND, ZERO>), dim3(grid), dim3(256), lds, 0, out, iters);
    const int launches = 1200;
    (void)hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((stream<KIND, ZERO>), dim3(grid), dim3(256), lds, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)launches * iters * 64 * grid * 4;
    printf("%-44s %7.1f ms  %6.1f TFLOP/s sustained  (%.2f ns per instruction and SIMD)\n", name, ms, insts * flop_per_inst / (ms * 1e-3) / 1e12,
           ms * 1e6 / ((double)launches * iters * 64 * 2));
    fflush(stdout);
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 4096);
    run<0, true>("4x4x1 f32, all-zero operands", out, 512.0, 800);
    run<0, false>("4x4x1 f32, random operands", out, 512.0, 800);
    run<1, true>("16x16x4 f32, all-zero operands", out, 2048.0, 200);
    run<1, false>("16x16x4 f32, random operands", out, 2048.0, 200);
    run<2, true>("32x32x2 f32, all-zero operands", out, 4096.0, 100);
    run<2, false>("32x32x2 f32, random operands", out, 4096.0, 100);
    run<0, false>("4x4x1 f32, random operands (again)", out, 512.0, 800);
    return 0;
}
