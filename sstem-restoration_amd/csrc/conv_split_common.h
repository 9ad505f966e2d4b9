// Shared by conv_split_kernels.hip (forward / data gradient) and conv_split_wgrad.hip (weight gradient): piece splitting, amax words,
// tile constants.  Everything here has internal linkage (one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint64_t u64x2v __attribute__((ext_vector_type(2)));

// where a step's MFMA loop commits staged pixels (split + LDS stores between the MFMAs): N pixels, the lane's pixels J0 .. J0 + N - 1,
// one per item from item I0 on; I0 < 0: in the middle of the step (the classic kernel's rule)
template <int I0_, int N_, int J0_> struct CommitPlan { static constexpr int I0 = I0_, N = N_, J0 = J0_; };
constexpr int SKC = 16;                       // input channels per K chunk
constexpr int STH = 8, STW = 32;              // output tile (rows x columns)
constexpr int SIN_PW = STW + 2;
// One piece image of the input tile in LDS: [channel half (8 channels = 16 B)][tile row][column], 16-byte slots, rows PITCH = PW | 1
// slots apart (odd).  Round 4, from SQ_LDS_BANK_CONFLICT: the former [pixel][16 channels] image cost 66 % of the LDS cycles in conflicts --
// the eight lanes a ds_write_b128 group holds stored pixels 4 columns = 128 B apart (8-way), and the 16 lanes of a ds_read_b128 group
// read one half of 32-byte pixels (2-way).  Here a fragment read's 16 lanes read 16 consecutive slots, and the staging lanes are dealt
// 4 rows x 2 column groups per store group: slots r PITCH + 4 q + j cover all eight residues (profiles/r04/n_*).
constexpr int SIN_BYTES = 11200;              // 2 halves x 10 rows x 35 slots x 16 B (32-wide tiles; 18 x 19 slots x 2 for 16-wide: 10944)
constexpr uint32_t S_OOB = 0x80000000u;

typedef __attribute__((address_space(1))) float gfloat_t;
typedef __attribute__((address_space(1))) uint8_t gbyte_t;
template <typename T>
__device__ __forceinline__ void pin_uptr(T*& p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ void st_lane(float* ubase, uint32_t lane_byte_off, float v)
{
    *reinterpret_cast<gfloat_t*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off) = v;
}
__device__ __forceinline__ float ld_lane(const float* ubase, uint32_t lane_byte_off)
{
    return *reinterpret_cast<const gfloat_t*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off);
}
__device__ __forceinline__ void pin_s(uint32_t& v) { asm volatile("" : "+s"(v)); }

__device__ __forceinline__ float act_s(float v, int act, float slope)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

// x -> P bf16 pieces with x = sum of the pieces (exactly for P = 3; to 2^-17 relative for P = 2)
template <int P>
__device__ __forceinline__ void split_pieces(float x, __bf16 (&o)[P])
{
    float r = x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        o[p] = (__bf16)r;
        if (p + 1 < P) r -= (float)o[p];
    }
}

// ---- two fp16 pieces (SSTEM_CONV_MFMA_F16X3) -----------------------------------------------------------------------------------------
// x * s = h0 + h1 with h0 = fp16(x * s), h1 = fp16(x * s - h0) (the subtraction is exact in fp32): 11 + 11 significant bits, and the
// three products h0 g0 + h0 g1 + h1 g0 (each exact in fp32, summed by the MFMA's fp32 accumulator) give x * y to 2^-22 relative per product
// -- 45x finer than the two-piece bf16 id at the same three MFMAs per term (v_mfma_f32_32x32x16_f16), half the MFMAs of X6.  fp16 has
// fp32's precision problem turned around: 5 exponent bits.  Every tensor therefore carries a power-of-two scale taken from an upper
// bound of its largest magnitude (an "amax word": 1024 float slots, the bound is their maximum; producers atomicMax into slot
// (workgroup & 1023), a consumer reduces them): s = 2^(141 - e), e = biased exponent of the bound, puts the largest value in
// [2^14, 2^15) and leaves 2^-14 .. 2^15 (18 binades below the bound keep the full 22 bits; smaller values fade out with an absolute
// error of 2^-25 of the bound).  Scales are exact (powers of two) and are taken out of the accumulators by one v_ldexp per value.
// Not a bit copy under one-hot weights (22 of fp32's 24 bits survive); inference only (no masks, no weight gradient).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int AMAX_SLOTS = 1024;        // 64 lines of 64 bytes: a launch's atomics (one per workgroup) spread over all of them

__device__ __forceinline__ void split_pieces_f16(float xs, __bf16 (&o)[2])
{
    const _Float16 h0 = (_Float16)xs;
    const _Float16 h1 = (_Float16)(xs - (float)h0);
    o[0] = __builtin_bit_cast(__bf16, h0);
    o[1] = __builtin_bit_cast(__bf16, h1);
}
// biased exponent e of a bound, clamped so that 2^(141 - e) and its inverse are normal floats; non-finite bound: scale 1
__host__ __device__ inline int amax_exponent(float amax)
{
    int e = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu);
    if (e == 255) e = 141;
    return e < 16 ? 16 : (e > 250 ? 250 : e);
}
__device__ __forceinline__ float scale_of_exponent(int e) { return __builtin_bit_cast(float, (uint32_t)(268 - e) << 23); }
// maximum of the slots of an amax word, by the calling wave (uniform result): 4 KB, four 16-byte loads per lane
__device__ __forceinline__ float amax_word_max(const float* __restrict__ word)
{
    const f32x4v* w4 = reinterpret_cast<const f32x4v*>(word) + (threadIdx.x & 63);
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < AMAX_SLOTS / 256; ++k) {
        const f32x4v v = w4[k * 64];
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}
// The calling WORKGROUP's contribution to an amax word: every thread of the workgroup calls it (the reduction uses a barrier and four
// floats of shared memory nobody else is using); ONE atomic per workgroup.  Device-scope atomics execute at the memory side, about 11 ns
// each and one after the other per 64-byte line (the guide's 'fanin' row): with one per wave, the 16k atomics of a slice-sum launch took
// 80 us where the launch takes 6.  Non-negative floats order like their bit patterns.
__device__ __forceinline__ void amax_word_update(float* __restrict__ word, float lane_max, uint32_t slot, float* red4)
{
#pragma unroll
    for (int off = 32; off; off >>= 1) lane_max = fmaxf(lane_max, __shfl_xor(lane_max, off));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = lane_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red4[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, red4[i]);
        atomicMax(reinterpret_cast<unsigned int*>(word) + (slot & (AMAX_SLOTS - 1)), __builtin_bit_cast(uint32_t, m));
    }
}

// per kernel instance (`done` belongs to the call site) and device, once: the kernels' dynamic LDS is above the 64 KB default
inline hipError_t wgrad_split_lds(const void* kernel, int bytes, bool (&done)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return hipSuccess;
}

}  // namespace
}  // namespace sstem
