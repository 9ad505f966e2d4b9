// Micro-benchmark: SUSTAINED rate of v_mfma_f32_32x32x16_f16 under the socket's power cap as a function of how often the A and B operands
// change between consecutive instructions (random fp16 data in both; all-zero operands for reference).  Every variant runs for about a
// second (back-to-back launches), so the figure is the rate the power management settles on, not a burst.  Question behind it: does the
// ORDER of a convolution tile's MFMAs (which operand stays, which rotates) change the energy per instruction?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_operand_power mfma_operand_power.hip && ./mfma_operand_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline f16x8 frag(uint32_t seed, bool zero)
{
    u32x4 v;
    for (int i = 0; i < 4; ++i) {
        uint32_t h = hash(seed * 4u + i);
        // two fp16 values with random sign and mantissa, exponent 8..15 (|x| in 2^-7 .. 2): no inf / nan, no denormals
        uint32_t lo = (h & 0x83ffu) | ((8u + ((h >> 10) & 7u)) << 10), hi = ((h >> 16) & 0x83ffu) | ((8u + ((h >> 26) & 7u)) << 10);
        v[i] = zero ? 0u : (lo | (hi << 16));
    }
    return __builtin_bit_cast(f16x8, v);
}

// PA / PB: the operand changes every PA / PB instructions (0 = never); 8 fragments of each in registers, 8 accumulators
// KIND 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_32x32x16_bf16 (the same random 16-bit patterns read as bf16), 2: v_mfma_f32_16x16x32_f16
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PA, int PB, bool ZERO, int KIND = 0>
__global__ __launch_bounds__(256) void stream(float* out, int iters)
{
    f16x8 a[8], b[8];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    for (int i = 0; i < 8; ++i) { a[i] = frag(t * 16u + i, ZERO); b[i] = frag(t * 16u + 8u + i, ZERO); }
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int ia = PA ? (u / PA) % 8 : 0, ib = PB ? (u / PB) % 8 : 0;
            if constexpr (KIND == 0) acc[u % 8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[u % 8], 0, 0, 0);
            else if constexpr (KIND == 1)
                acc[u % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ia]), __builtin_bit_cast(bf16x8, b[ib]), acc[u % 8], 0, 0, 0);
            else {      // two 16x16x32 per slot: the same multiply count as one 32x32x16
                f32x4 lo = {acc[u % 8][0], acc[u % 8][1], acc[u % 8][2], acc[u % 8][3]}, hi = {acc[u % 8][4], acc[u % 8][5], acc[u % 8][6], acc[u % 8][7]};
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ib], b[ia], hi, 0, 0, 0);
                for (int q = 0; q < 4; ++q) { acc[u % 8][q] = lo[q]; acc[u % 8][4 + q] = hi[q]; }
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 16; ++q) s += acc[k][q];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int PA, int PB, bool ZERO, int KIND = 0>
static void run(const char* name, float* out)
{
    const int iters = 400, lds = 60 * 1024, grid = 512;       // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<PA, PB, ZERO, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((stream<PA, PB, ZERO, KIND>), dim3(grid), dim3(256), lds, 0, out, iters);     // settle
    const int launches = 1200;
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((stream<PA, PB, ZERO, KIND>), dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)launches * iters * 64 * grid * 4;
    printf("%-44s %7.1f ms  %6.0f TFLOP/s sustained\n", name, ms, mfmas * 32768.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    float* out;
    hipMalloc(&out, 4096);
    run<0, 0, true>("all-zero operands", out);
    run<0, 0, false>("A and B never change", out);
    run<1, 0, false>("A changes every instruction, B never", out);
    run<0, 1, false>("B changes every instruction, A never", out);
    run<1, 1, false>("A and B change every instruction", out);
    run<1, 3, false>("A every instruction, B every 3 (the kernel)", out);
    run<4, 1, false>("A every 4, B every instruction", out);
    run<4, 4, false>("A and B every 4", out);
    run<8, 1, false>("A every 8, B every instruction", out);
    run<1, 1, false>("A and B change every instruction (again)", out);
    run<1, 1, false, 1>("bf16 32x32x16, A and B every instruction", out);
    run<1, 3, false, 1>("bf16 32x32x16, A every instruction, B every 3", out);
    run<0, 0, true, 1>("bf16 32x32x16, all-zero operands", out);
    run<1, 1, false, 2>("f16 16x16x32 (two per slot), A and B every slot", out);
    run<1, 1, false>("f16 32x32x16, A and B every instruction (third)", out);
    return 0;
}
