// fp32 3x3 convolutions on the bf16 matrix cores of gfx950 by operand splitting (algorithm ids SSTEM_CONV_MFMA_BF16X3 / _BF16X6).
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32, conv_kernels.hip) runs at 1/16 of the MAC rate of v_mfma_f32_32x32x16_bf16, and gfx950 has
// no xf32 form.  An fp32 value is the exact sum of three bf16 pieces,
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)          (8 + 8 + 8 significant bits; both subtractions
// are exact in fp32), and a product of two bf16 values is exact in fp32.  So
//     x * y = hh + (hm + mh) + (hl + mm + lh) + [ml + lm + ll]
// and the six products outside the brackets, each an exact fp32 number summed by the MFMA's fp32 accumulator, give x * y to
// 2 * 2^-27 relative -- below half an fp32 ulp (2^-24): the arithmetic of an fp32 convolution (exact products, fp32 sums) at
// 6/16 of the fp32 MFMA's pipe time.  P = 3 pieces / 6 products is SSTEM_CONV_MFMA_BF16X6.  P = 2 pieces / 3 products
// (hh + hm + mh; SSTEM_CONV_MFMA_BF16X3) drops terms of <= 3 * 2^-18 = 1.1e-5 relative per product (random signs: far below that on a
// sum) at 3/16 of the pipe time.  The C-ABI's SSTEM_CONV_AUTO selects neither; hipnn's ALGO_AUTO runs the layers where X6 is the faster
// kernel under it (hipnn/functional.py, _auto_algo / _wgrad_algo) and never picks X3.
// Limits of the split: |x| near FLT_MAX rounds h to infinity (and an infinite input gives NaN where the fp32 kernel gives inf);
// pieces below the bf16 denormal range are lost (absolute 1e-38-ish).
//
// Same layers, same entry, same tiling, staging, XCD-aware tile order, split over K and epilogue as conv3x3_bf16_mfma
// (conv_bf16_kernels.hip: model_interp.py:121-143, networks.py:179-186, model_unet.py:11-48, model_fusionnet.py:12-43); what differs:
//   * LDS holds P images of the input tile (one per piece, each [2 channel halves][10 rows][35 slots of 8 channels] 16-bit, double-buffered: SIN_BYTES below);
//   * the weights are packed [co block][chunk][piece][tap][co][16 ci] bf16; a wave keeps the nine A fragments of ONE piece in
//     registers and fetches the next piece's (or next chunk's) nine during the MFMAs of the current one;
//   * per chunk the wave walks weight piece pa = 0..P-1 and, per input row, the input pieces pb with pa + pb < P.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv_kernels.h"
#include "conv_split_common.h"

// developer ablation builds of conv3x3_split_mfma (wrong results; tools/build_ablate_split.sh): 1 no per-item LDS reads of the B operand,
// 2 no staging commit (split + LDS stores), 4 no staging loads, 8 no A-fragment loads after the first, 64 no epilogue stores (whole tiles);
// such builds also read SSTEM_SPLIT_LDS_PAD (extra dynamic LDS bytes per workgroup: occupancy experiments)
#ifndef SSTEM_SPLIT_AUTOWAIT
#define SSTEM_SPLIT_AUTOWAIT 1   // 0: every step starts with s_waitcnt vmcnt(0) (A/B builds)
#endif
#ifndef SSTEM_SPLIT_ABLATE
#define SSTEM_SPLIT_ABLATE 0
#endif
#ifndef SSTEM_SPLIT_DEV
#define SSTEM_SPLIT_DEV 0
#endif

namespace sstem {
namespace {

// max |x| of a tensor into an amax word (zeroed by the caller): the pre-pass for tensors no producer has bounded
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ word)
{
    float m = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4v v = reinterpret_cast<const f32x4v*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[4];
    amax_word_update(word, m, blockIdx.x, red);
}

// A ragged last chunk of 1..4 input channels (51 = 3 x 16 + 3: the kernel heads of the IFNet) would spend nine K steps of 16 on 3 live
// columns.  Under the three-piece id it is packed by tap ROW instead: K = kx * 4 + channel, one K step per tap row (the kernel stages that
// chunk as [pixel][4 channels], so a pixel's K vector is the 8 bytes of itself and of its two right-hand neighbours): 3 K steps instead of
// 9 for that chunk.  The two-piece fp16 id (F16X3) does the same.  A pure function of the channel count and the id, so packing and launches
// agree without a flag.
#ifndef SSTEM_SPLIT_TAIL
#define SSTEM_SPLIT_TAIL 1          // 0: A/B builds without the tap-row chunk (tools/build_ablate_split.sh)
#endif
__host__ __device__ inline bool split_tail_chunk(int cin, int P, bool f16 = false)
{
    return SSTEM_SPLIT_TAIL && (P == 3 || f16) && cin > 16 && cin % 16 >= 1 && cin % 16 <= 4;
}

// one element of a packed weight image [chunk][piece][tap][output channel, padded to COP][16 input channels]: the layout does not
// depend on the output-channel block a launch chooses (32 or 64 per workgroup, by grid size)
template <int P, bool F16 = false>
__device__ __forceinline__ __bf16 packed_weight(const float* __restrict__ w, int64_t idx, int cin, int cout, int COP, int nchunks,
                                                bool transposed_flipped, float wscale = 1.f)
{
    const int cl = idx % SKC;
    int64_t r = idx / SKC;
    const int co = r % COP; r /= COP;
    const int tap = r % 9; r /= 9;
    const int piece = r % P; r /= P;
    const int chunk = (int)r;
    int ci = chunk * SKC + cl;
    int wtap = tap;
    bool live = true;
    if (split_tail_chunk(cin, P, F16) && chunk == nchunks - 1) {         // slots 0..2 of the tap axis hold the tap rows, the rest is not read
        const int kx = cl >> 2;
        ci = chunk * SKC + (cl & 3);
        wtap = tap * 3 + kx;
        live = tap < 3 && kx < 3;
    }
    float v = 0.f;
    if (live && ci < cin && co < cout)
        v = transposed_flipped ? w[((int64_t)ci * cout + co) * 9 + (8 - wtap)] : w[((int64_t)co * cin + ci) * 9 + wtap];
    __bf16 pc[P];
    if constexpr (F16) split_pieces_f16(v * wscale, pc); else split_pieces<P>(v, pc);
    __bf16 res = pc[0];
#pragma unroll
    for (int p = 1; p < P; ++p) if (piece == p) res = pc[p];
    return res;
}

// both fp16 pieces of ONE weight in one go (the group launch after an optimiser step: the weight is read and its slot decoded once, not
// once per piece): j counts the slots of a packed image without the piece axis, [chunk][tap][co, padded to COP][16 input channels];
// the two elements go to wp[first] and wp[first + 9 COP 16] -- the same values packed_weight<2, true> gives those two indices
__device__ __forceinline__ void pack_pair_f16(const float* __restrict__ w, __bf16* __restrict__ wp, int64_t j, int cin, int cout, int COP,
                                              int nchunks, bool transposed_flipped, float wscale)
{
    const int cl = j % SKC;
    int64_t r = j / SKC;
    const int co = r % COP; r /= COP;
    const int tap = r % 9;
    const int chunk = (int)(r / 9);
    int ci = chunk * SKC + cl;
    int wtap = tap;
    bool live = true;
    if (split_tail_chunk(cin, 2, true) && chunk == nchunks - 1) {
        const int kx = cl >> 2;
        ci = chunk * SKC + (cl & 3);
        wtap = tap * 3 + kx;
        live = tap < 3 && kx < 3;
    }
    float v = 0.f;
    if (live && ci < cin && co < cout)
        v = transposed_flipped ? w[((int64_t)ci * cout + co) * 9 + (8 - wtap)] : w[((int64_t)co * cin + ci) * 9 + wtap];
    __bf16 pc[2];
    split_pieces_f16(v * wscale, pc);
    const int64_t first = (((int64_t)chunk * 2) * 9 + tap) * COP * SKC + (int64_t)co * SKC + cl;
    wp[first] = pc[0];
    wp[first + (int64_t)9 * COP * SKC] = pc[1];
}

// fp16 pieces of [Cout,Cin,3,3] weights: wp[0..7] is a 16-byte header (the weights' bound as a float, written here from the amax word the
// launcher filled), the packed image follows
__global__ void pack_weights_3x3_split_f16(const float* __restrict__ w, __bf16* __restrict__ wp, const float* __restrict__ w_amax_word,
                                           int Cin, int Cout, int COP, int nchunks, int64_t n, int transposed)
{
    const float bound = amax_word_max(w_amax_word);
    const float wscale = scale_of_exponent(amax_exponent(bound));
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(wp) = bound;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        wp[8 + i] = packed_weight<2, true>(w, i, Cin, Cout, COP, nchunks, transposed != 0, wscale);
}

// both orientations of a recorded layer's fp16 pieces (forward image and the transposed + flipped image of its data gradient; either may
// be null), each behind its own header, under ONE bound
__global__ void pack_weights_3x3_split_f16_both(const float* __restrict__ w, __bf16* __restrict__ wp_f, __bf16* __restrict__ wp_t,
                                                const float* __restrict__ w_amax_word, int Cin, int Cout, int COP_f, int nchunks_f,
                                                int64_t n_fwd, int COP_t, int nchunks_t, int64_t n_t)
{
    const float bound = amax_word_max(w_amax_word);
    const float wscale = scale_of_exponent(amax_exponent(bound));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_fwd) *reinterpret_cast<float*>(wp_f) = bound;
        if (n_t) *reinterpret_cast<float*>(wp_t) = bound;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_t; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= n_fwd) wp_t[8 + i - n_fwd] = packed_weight<2, true>(w, i - n_fwd, Cout, Cin, COP_t, nchunks_t, true, wscale);
        else wp_f[8 + i] = packed_weight<2, true>(w, i, Cin, Cout, COP_f, nchunks_f, false, wscale);
    }
}

// many layers in one go (fp16 pieces): table entries as pack_weights_3x3_split_group's, plus [14] = the entry's first block in the
// bound launch (4096 weights per block) and [15] = the address of the float that launch raises to the layer's largest magnitude
// (zeroed by the launcher; non-negative floats order like their bit patterns)
constexpr int AMAX_GROUP_ELEMS = 4096;
__global__ __launch_bounds__(256) void amax_weights_group(const int64_t* __restrict__ table, int n_entries)
{
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 16 + 14] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* en = table + (int64_t)lo * 16;
    const float* w = reinterpret_cast<const float*>(en[0]);
    const int64_t n = en[3] * en[4] * 9;
    const int64_t base = ((int64_t)blockIdx.x - en[14]) * AMAX_GROUP_ELEMS;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < AMAX_GROUP_ELEMS / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n) m = fmaxf(m, fabsf(w[i]));
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned int*>(en[15]), __builtin_bit_cast(uint32_t, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
__global__ __launch_bounds__(256) void pack_weights_3x3_split_f16_group(const int64_t* __restrict__ table, int n_entries)
{
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 16 + 13] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* en = table + (int64_t)lo * 16;
    const float* w = reinterpret_cast<const float*>(en[0]);
    __bf16* wp_f = reinterpret_cast<__bf16*>(en[1]);
    __bf16* wp_t = reinterpret_cast<__bf16*>(en[2]);
    const int Cin = (int)en[3], Cout = (int)en[4];
    const int64_t n_fwd = en[8], n_t = en[12];
    const float bound = *reinterpret_cast<const float*>(en[15]);
    const float wscale = scale_of_exponent(amax_exponent(bound));
    const int64_t j = ((int64_t)blockIdx.x - en[13]) * 256 + threadIdx.x;          // one thread per WEIGHT slot: both pieces
    if (j == 0) { *reinterpret_cast<float*>(wp_f) = bound; *reinterpret_cast<float*>(wp_t) = bound; }
    const int64_t h_fwd = n_fwd / 2, h_t = n_t / 2;
    if (j >= h_fwd + h_t) return;
    if (j >= h_fwd) pack_pair_f16(w, wp_t + 8, j - h_fwd, Cout, Cin, (int)en[9], (int)en[10], true, wscale);
    else pack_pair_f16(w, wp_f + 8, j, Cin, Cout, (int)en[5], (int)en[6], false, wscale);
}

// forward packing of [Cout,Cin,3,3] and / or the transposed + flipped packing its data gradient uses (either may be null: n = 0)
template <int P>
__global__ void pack_weights_3x3_split_both(const float* __restrict__ w, __bf16* __restrict__ wp_f, __bf16* __restrict__ wp_t, int Cin,
                                            int Cout, int COP_f, int nchunks_f, int64_t n_fwd, int COP_t, int nchunks_t, int64_t n_t,
                                            int fwd_is_transposed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const bool t = i >= n_fwd;
        if (t) wp_t[i - n_fwd] = packed_weight<P>(w, i - n_fwd, Cout, Cin, COP_t, nchunks_t, true);
        else wp_f[i] = packed_weight<P>(w, i, Cin, Cout, COP_f, nchunks_f, fwd_is_transposed != 0);
    }
}

// many layers in one launch: the table of pack_weights_3x3_group (conv_kernels.hip), entries from pack_group_entry_split
template <int P>
__global__ __launch_bounds__(256) void pack_weights_3x3_split_group(const int64_t* __restrict__ table, int n_entries)
{
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 16 + 13] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* en = table + (int64_t)lo * 16;
    const float* w = reinterpret_cast<const float*>(en[0]);
    __bf16* wp_f = reinterpret_cast<__bf16*>(en[1]);
    __bf16* wp_t = reinterpret_cast<__bf16*>(en[2]);
    const int Cin = (int)en[3], Cout = (int)en[4];
    const int64_t n_fwd = en[8], n_t = en[12];
    const int64_t i = ((int64_t)blockIdx.x - en[13]) * 256 + threadIdx.x;
    if (i >= n_fwd + n_t) return;
    if (i >= n_fwd) wp_t[i - n_fwd] = packed_weight<P>(w, i - n_fwd, Cout, Cin, (int)en[9], (int)en[10], true);
    else wp_f[i] = packed_weight<P>(w, i, Cin, Cout, (int)en[5], (int)en[6], false);
}

// MASKED: in_mask (nullable, [N,Cin,H,W] bytes) zeroes the input elements whose byte is 0 while they are staged -- the data gradient of
// a layer whose incoming gradient still has to pass the ReLU of the layer's output (g * (out > 0) without a pass of its own);
// out_mask (nullable, [N,Cout,H,W] bytes) receives (stored activation > 0) -- that mask, written by the forward launch.
// WT = 16 (16-byte staging only): maps up to 16 pixels wide -- the 32 pixel columns of an MFMA row are TWO image rows of 16, the tile is
// 16 x 16 pixels (18 x 18 with its halo: 324 tile pixels in the same LDS image), so a 16 x 16 map is one whole tile instead of a tile
// whose right half is padding (the deep levels of the U-Nets and of the IFNet at 256 x 256 inputs).
// TAIL: the last chunk is a tap-row chunk (split_tail_chunk): staged as [pixel][4 channels], 3 K steps.
// F16: the two pieces are fp16 (split_pieces_f16): in_amax = the input's amax word, wp = the packed image behind its header
// (w_bound = the header: the weights' bound).  out_amax (any piece format, nullable): the launch adds the largest magnitude it stores to
// that amax word, for the next layer.
// DEEP (round 4; fp16 pieces, 16-byte staging, 32-wide tiles, no tap-row chunk, no K slices): the workgroup WALKS `walk` tiles down the
// image and treats their chunks as ONE stream -- the tile loads run TWO stream steps ahead of the MFMAs (a second set of staging
// registers), the staging commit one step ahead, across tile boundaries; a tile's epilogue sits between two steps of the stream.  For
// the 32-output-channel block (two MFMA rows per wave: 54 MFMAs = 0.8 us per chunk) one step of lead did not cover a memory latency and
// every tile began with an exposed one: a 2-chunk layer (8 x 32 -> 32 at 1024^2) spent 7 us per tile on 1.6 us of MFMAs.
// CT (round 4): the launch is the sub-pixel form of a ConvTranspose2d(k3, s2, p1, op1) (model_fusionnet.py:21-27, model_unet.py:32,70).
// An output pixel (2y + py, 2x + px) of the transposed convolution receives 1, 2, 2 or 4 taps, all from the input window
// in[y .. y+1][x .. x+1]: a 2 x 2 convolution with 4 C output channels (parity-major: co' = (2 py + px) C + co), stored with a pixel
// shuffle.  The 2 x 2 window sits in taps (ky, kx) in {1, 2}^2 of this kernel's 3 x 3 machinery (same tile, same staging): the
// caller packs weights [4C, Cin, 3, 3] with W'[(py,px) co][ci][1+dy][1+dx] = wT[ci][co][kyT(py,dy)][kxT(px,dx)] (kyT(0,0) = 1,
// kyT(1,0) = 2, kyT(1,1) = 0, no tap for (0,1)), the taps with ky == 0 or kx == 0 are neither loaded nor multiplied, and the whole-tile
// store writes out[n][co][2y + py][2x + px] (a residual is read from the same place).  9 of the 16 issued taps are real, on the fp16
// two-piece id: a third of the fp32 MFMA kernel's matrix-pipe time.  C must be a multiple of 32 (a wave's 32 channels share a parity).
template <int WCO, int WR, int P, bool VEC, bool MASKED = false, int WT = 32, bool TAIL = false, bool F16 = false, bool DEEP = false,
          bool CT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_split_mfma(
    const float* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int act, float slope, int ksplit, float* __restrict__ slab,
    int xcd_remap, const float* __restrict__ residual, float res_scale, int COP, const uint8_t* __restrict__ in_mask = nullptr,
    uint8_t* __restrict__ out_mask = nullptr, const float* __restrict__ in_amax = nullptr, const float* __restrict__ w_bound = nullptr,
    float* __restrict__ out_amax = nullptr, int out_blocked = 0, int walk = 1, int64_t out_img = 0, float* __restrict__ pool_out = nullptr,
    int pool_kind = 0)
{
    static_assert(WCO * WR == 4, "four waves");
    static_assert(!DEEP || (F16 && VEC && WT == 32 && !TAIL && !MASKED), "tile-walking stream: the fp16 inference instances");
    static_assert(!CT || (F16 && VEC && WT == 32 && !TAIL && !MASKED), "sub-pixel ConvTranspose: an fp16 inference instance");
    static_assert(P == 2 || P == 3, "two or three pieces");
    static_assert(!F16 || P == 2, "fp16 pieces: two of them");      // (MASKED && F16, round 5: recorded launches on the fp16 id)
    static_assert(WT == 32 || (WT == 16 && VEC), "16-wide tiles: 16-byte staging only");
    static_assert(!TAIL || P == 3 || F16, "tap-row chunks: the three-piece and the fp16 ids");
    constexpr int CO = 32 * WCO, R = STH / WR;                   // R MFMA rows (32 pixels each) per wave
    constexpr int RS = WT == 32 ? 1 : 2;                         // image rows per MFMA row
    constexpr int TROWS = STH * RS;                              // image rows per tile (8 / 16)
    constexpr int PW = WT + 2, NPX = PW * (TROWS + 2);           // tile with its halo: 34 x 10 / 18 x 18 pixels
    constexpr int PITCH = PW | 1;                                // slots per tile row (35 / 19)
    constexpr int HOFF = (TROWS + 2) * PITCH * 16;               // bytes from channel half 0 to half 1
    static_assert(2 * HOFF <= SIN_BYTES && NPX * 8 <= SIN_BYTES, "LDS image");
    auto px_off = [](int row, int col, int half) { return (row * PITCH + col) * 16 + half * HOFF; };
    // 2 buffers x P piece images of the input tile + one 16-byte slot per thread where lanes without a pixel park their staging stores
    // (an unconditional store keeps the staging commit straight-line code that can be scheduled between the MFMAs)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PARK = 2 * P * SIN_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int wco = wave % WCO, wr = wave / WCO;
    // XCD-aware tile order (see conv3x3_bf16_mfma): XCD k owns a contiguous run of (channel block, x, y, K slice, image)
    int bx = blockIdx.x, by = blockIdx.y, ks = blockIdx.z % ksplit, n = (blockIdx.z / ksplit) / ncb, cb = (blockIdx.z / ksplit) % ncb;
    if (xcd_remap) {
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const uint32_t k = lin & 7u, q = total >> 3, rem = total & 7u;
        uint32_t t = k * q + (k < rem ? k : rem) + (lin >> 3);
        cb = (int)(t % (uint32_t)ncb); t /= (uint32_t)ncb;
        bx = (int)(t % gx); t /= gx;
        by = (int)(t % gy); t /= gy;
        ks = (int)(t % (uint32_t)ksplit); n = (int)(t / (uint32_t)ksplit);
    }
    const int X0 = bx * WT;
    int Y0 = by * TROWS * (DEEP ? walk : 1);                     // DEEP: the first of the tiles this workgroup walks (updated per tile)
    const int cpk = nchunks / ksplit;
    const int c_first = ks * cpk, c_end = c_first + cpk;
    const int64_t plane = (int64_t)H * W;
    const uint32_t plane4 = (uint32_t)plane * 4u;
    float sx = 1.f;              // F16: the input's scale, and the exponent that takes both scales out of the sums again
    int descale = 0;
    if constexpr (F16) {
        const int ex = amax_exponent(amax_word_max(in_amax)), ew = amax_exponent(*w_bound);
        sx = scale_of_exponent(ex);
        descale = ex + ew - 282;
    }
    // F16, whole-tile stores: v = act(acc * mul[co] + add[co]) with mul = scale 2^descale, add = bias scale + shift -- the same two numbers
    // for every tile, wave and lane of the workgroup, so they are made ONCE per workgroup (CO channels x 8 bytes of LDS behind the
    // parking slots; published by the first barrier) instead of per accumulator register of every tile: the store phase was ~37
    // instructions per stored value with two MFMA rows per wave (scalar loads, dead-channel selects, lane-half selects, ldexp, fma per
    // channel pair) -- 40 % of a 2-chunk tile's instructions (round 4).  A channel behind the last one gets (0, 0).
    // The bf16-piece ids keep their arithmetic, v = act((acc + bias) scale + shift), and read the three numbers from the same table.
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    f32x2v* ptab = reinterpret_cast<f32x2v*>(lds + PARK + 256 * 16);
    float* ptab3 = reinterpret_cast<float*>(lds + PARK + 256 * 16);          // !F16: [3][CO] = bias, scale, shift
    if (tid < CO) {
        const int co = cb * CO + tid;
        const bool live = co < Cout;
        const float bsv = (bias && live) ? bias[live ? co : 0] : 0.f;
        const float scv = live ? (scale ? scale[co] : 1.f) : 0.f;
        const float shv = (shift && live) ? shift[live ? co : 0] : 0.f;
        if constexpr (F16) ptab[tid] = (f32x2v){__builtin_ldexpf(scv, descale), __builtin_fmaf(bsv, scv, shv)};
        else { ptab3[tid] = bsv; ptab3[CO + tid] = scv; ptab3[2 * CO + tid] = shv; }
    }

    // ---- dword staging (any W): 12 wave-items (2 channel halves x 6 groups of 64 tile pixels), 3 per wave
    const rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)n * Cin * plane), 0,
                                                         (int)((uint32_t)Cin * plane4), 0x00020000);
    uint32_t voff[3];
    int lds_off[3], lds_tail[3];                                 // 16-channel image / tap-row image ([pixel][4 channels])
    int half_of[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int wi = wave * 3 + k;
        const int half = wi / 6;
        const int px = (wi % 6) * 64 + lane;
        const int row = px / PW, col = px - row * PW;
        const int y = Y0 - 1 + row, x = X0 - 1 + col;
        const bool inside = px < NPX && y >= 0 && y < H && x >= 0 && x < W;
        voff[k] = inside ? (uint32_t)(y * W + x) * 4u : S_OOB;
        lds_off[k] = px < NPX ? px_off(row, col, half) : -1;
        lds_tail[k] = px * 8;
        half_of[k] = half;
    }
    float stg[VEC ? 1 : 3][8];
    auto issue_in = [&](int chunk) {
        const int cl_lim = Cin - chunk * SKC;
        const uint32_t sbase = (uint32_t)(chunk * SKC) * plane4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = half_of[k] * 8 + i;
                float v = 0.f;
                if (c < cl_lim)
                    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, (int)voff[k], (int)(sbase + (uint32_t)c * plane4), 0));
                if constexpr (MASKED) {
                    if (in_mask && c < cl_lim && voff[k] != S_OOB &&
                        in_mask[((int64_t)n * Cin + chunk * SKC + c) * plane + (voff[k] >> 2)] == 0) v = 0.f;
                }
                stg[VEC ? 0 : k][i] = v;
            }
    };
    auto commit_in = [&](int buf, int chunk) {
        const bool tailfmt = TAIL && chunk == nchunks - 1;          // uniform
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bf16x8 pk[P];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 pc[P];
                if constexpr (F16) split_pieces_f16(stg[VEC ? 0 : k][i] * sx, pc); else split_pieces<P>(stg[VEC ? 0 : k][i], pc);
#pragma unroll
                for (int p = 0; p < P; ++p) pk[p][i] = pc[p];
            }
            if (lds_off[k] >= 0) {
                if (!tailfmt) {
#pragma unroll
                    for (int p = 0; p < P; ++p) *reinterpret_cast<bf16x8*>(lds + (buf * P + p) * SIN_BYTES + lds_off[k]) = pk[p];
                } else if (half_of[k] == 0) {                       // [pixel][4 channels]
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        *reinterpret_cast<uint64_t*>(lds + (buf * P + p) * SIN_BYTES + lds_tail[k]) = __builtin_bit_cast(u64x2v, pk[p])[0];
                }
            }
        }
    };

    // ---- 16-byte staging (W % 4 == 0, 16-B aligned input): see conv3x3_bf16_mfma
    const int vhalf = wave & 1;
    int vdst[4], vtail[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { vdst[j] = PARK + tid * 16; vtail[j] = 0; }
    int vrow = -1, vdx = 0;          // the lane's tile row and the column of its 16 bytes against the tile's first (both the lane's own)
    {
        int first_col = 0, only = -1;
        if constexpr (WT == 32) {
            // waves 0, 1: tile rows 0..7 x eight 4-pixel groups; eight consecutive lanes (one ds_write_b128 group) = 4 rows x 2 groups
            if (wave < 2) {
                const int q = 2 * (lane >> 4) + (lane & 1);
                vrow = 4 * ((lane >> 3) & 1) + ((lane & 7) >> 1); vdx = 4 * q; first_col = 1 + 4 * q;
            }
            else if (lane < 16) { vrow = 8 + (lane >> 3); vdx = 4 * (lane & 7); first_col = 1 + 4 * (lane & 7); }
            else if (lane < 36) {
                const int hl = lane - 16; vrow = hl >> 1;
                if (hl & 1) { vdx = WT; first_col = PW - 1; only = 0; }
                else { vdx = -4; first_col = 0 - 3; only = 3; }
            }
        } else {   // 18 tile rows x 4 groups: waves 0, 1 rows 0..15; waves 2, 3: lanes 0..7 rows 16, 17, lanes 8..43 the two halo columns
            if (wave < 2) {                                  // rows 0..15 x four groups, dealt 4 rows x 2 groups per eight lanes as above
                const int q = 2 * ((lane >> 3) & 1) + (lane & 1);
                vrow = 4 * (lane >> 4) + ((lane & 7) >> 1); vdx = 4 * q; first_col = 1 + 4 * q;
            }
            else if (lane < 8) { vrow = 16 + (lane >> 2); vdx = 4 * (lane & 3); first_col = 1 + 4 * (lane & 3); }
            else if (lane < 44) {
                const int hl = lane - 8; vrow = hl >> 1;
                if (hl & 1) { vdx = WT; first_col = PW - 1; only = 0; }
                else { vdx = -4; first_col = 0 - 3; only = 3; }
            }
        }
        if (vrow >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (only < 0 || only == j) { vdst[j] = px_off(vrow, first_col + j, vhalf); vtail[j] = (vrow * PW + first_col + j) * 8; }
        }
    }
    // byte offset of the lane's 16 bytes inside a channel plane for the tile at (x0, y0); S_OOB: outside the image (or no pixel)
    auto tile_voff = [&](int x0, int y0) -> uint32_t {
        const int y = y0 - 1 + vrow, xg = x0 + vdx;
        return (vrow >= 0 && y >= 0 && y < H && xg >= 0 && xg < W) ? (uint32_t)(y * W + xg) * 4u : S_OOB;
    };
    const uint32_t vvoff = tile_voff(X0, Y0);
    const char* in_n = reinterpret_cast<const char*>(in) + (int64_t)n * Cin * plane * 4;
    f32x4v stg4[DEEP ? 2 : 1][VEC ? 8 : 1];                                         // DEEP: two sets, a tile's loads two steps ahead
    uint32_t mk4[(VEC && MASKED) ? 8 : 1];                                         // the mask bytes of the lane's four pixels, per channel
    // in_img: the image's first byte (uniform), voff_t: tile_voff of the tile the chunk belongs to
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    auto issue_in_v = [&](int chunk, const char* in_img, uint32_t voff_t, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const uint32_t vsafe = voff_t != S_OOB ? voff_t : 0u;
        const int cl_lim = Cin - chunk * SKC;
        const char* pc = in_img + (int64_t)(chunk * SKC + vhalf * 8) * plane4;      // uniform
        if constexpr (MASKED) {
            const uint8_t* pm = in_mask + ((int64_t)n * Cin + chunk * SKC + vhalf * 8) * plane;      // uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool chan = in_mask != nullptr && (cl_lim >= SKC || vhalf * 8 + i < cl_lim);
                uint32_t m = 0x01010101u;
                if (chan) m = *reinterpret_cast<const uint32_t*>(pm + (int64_t)i * plane + (vsafe >> 2));
                mk4[VEC ? i : 0] = m;
            }
        }
        if constexpr (!F16) {       // (the bf16-piece instances lost 7-13 % with the buffer form below: measured, profiles/r03/m_*)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool chan = cl_lim >= SKC || vhalf * 8 + i < cl_lim;             // uniform
                f32x4v v = {0.f, 0.f, 0.f, 0.f};
                if (chan) v = *reinterpret_cast<const f32x4v*>(pc + (int64_t)i * plane4 + vsafe);
                stg4[SET][VEC ? i : 0] = v;
            }
            return;
        }
        // F16: a buffer resource over the lane half's (up to) eight channel planes of this chunk: lanes without a pixel inside the image
        // carry offset 2^31 and channels behind the last one lie behind the resource's end -- the range check returns zeros for both,
        // nothing to select when the values are split (the launcher keeps eight planes below 2^31 bytes on this path).
        const int live_ch = __builtin_amdgcn_readfirstlane(min(max(cl_lim - vhalf * 8, 0), 8));
        const uint64_t pcu = reinterpret_cast<uint64_t>(pc);             // uniform by construction: say so (no waterfall loop around the loads)
        const uint64_t pcs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pcu >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pcu);
        const rsrc_t rch = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(pcs), 0, (int)((uint32_t)live_ch * plane4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            stg4[SET][VEC ? i : 0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rch, (int)voff_t, (int)((uint32_t)i * plane4), 0));
    };
    auto commit_px_v = [&](int buf, int j, int chunk, uint32_t voff_t, auto set_tag) __attribute__((always_inline)) {      // pixel j of the lane's four
        constexpr int SET = decltype(set_tag)::value;
        const bool tailfmt = TAIL && chunk == nchunks - 1;          // uniform
        const bool vok = voff_t != S_OOB;
        {
            bf16x8 pk[P];
            if constexpr (F16) {
                // two channels per register: head = fp16(x s) and tail = fp16(fma(x, s, -head)) (x s is exact: s is a power of two) written
                // straight into the halves of the packed registers by the mixed-precision fma -- 4 instructions per pair, spelled out
                // because the compiler's own selection for the same arithmetic takes 7-9
                uint32_t hd[4], tl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v0 = stg4[SET][VEC ? 2 * i : 0][j], v1 = stg4[SET][VEC ? 2 * i + 1 : 0][j];
                    if constexpr (MASKED) {          // the ReLU mask of the layer this data gradient passes through, applied while staging
                        v0 = ((mk4[VEC ? 2 * i : 0] >> (8 * j)) & 0xffu) == 0u ? 0.f : v0;
                        v1 = ((mk4[VEC ? 2 * i + 1 : 0] >> (8 * j)) & 0xffu) == 0u ? 0.f : v1;
                    }
                    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hd[i]) : "v"(v0), "v"(sx));
                    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hd[i]) : "v"(v1), "v"(sx));
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(tl[i]) : "v"(v0), "v"(sx), "v"(hd[i]));
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(tl[i]) : "v"(v1), "v"(sx), "v"(hd[i]));
                }
                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                const u32x4v h4 = {hd[0], hd[1], hd[2], hd[3]}, t4 = {tl[0], tl[1], tl[2], tl[3]};
                pk[0] = __builtin_bit_cast(bf16x8, h4);
                pk[1] = __builtin_bit_cast(bf16x8, t4);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __bf16 pc[P];
                    float v = vok ? stg4[SET][VEC ? i : 0][j] : 0.f;
                    if constexpr (MASKED) v = ((mk4[VEC ? i : 0] >> (8 * j)) & 0xffu) == 0u ? 0.f : v;
                    split_pieces<P>(v, pc);
#pragma unroll
                    for (int p = 0; p < P; ++p) pk[p][i] = pc[p];
                }
            }
            if constexpr (!TAIL) {
                const int base = vdst[j] >= PARK ? 0 : buf * P * SIN_BYTES;       // parked stores: the slot itself
#pragma unroll
                for (int p = 0; p < P; ++p) *reinterpret_cast<bf16x8*>(lds + base + (vdst[j] >= PARK ? 0 : p * SIN_BYTES) + vdst[j]) = pk[p];
            } else {        // both formats are stored, the one that does not apply into the lane's parking slot (straight-line code)
                const bool live = vdst[j] < PARK;
                const bool st16 = live && !tailfmt, st8 = live && tailfmt && vhalf == 0;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    *reinterpret_cast<bf16x8*>(lds + (st16 ? (buf * P + p) * SIN_BYTES + vdst[j] : PARK + tid * 16)) = pk[p];
                    *reinterpret_cast<uint64_t*>(lds + (st8 ? (buf * P + p) * SIN_BYTES + vtail[j] : PARK + tid * 16)) =
                        __builtin_bit_cast(u64x2v, pk[p])[0];
                }
            }
        }
    };
    auto commit_in_v = [&](int buf, int chunk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) commit_px_v(buf, j, chunk, vvoff, S0());
    };

    // weights of this wave's 32 output channels: fragment (chunk, piece, tap) = 16 B per lane at [tap][co = wco*32 + r][h*8 ..]
    const __bf16* wp_lane0 = wp + (int64_t)(wco * 32 + r) * SKC + h * 8;       // + the channel block's cb * CO * SKC
    const int tap_stride = COP * SKC;
    auto load_a = [&](bf16x8 (&a)[9], int chunk, int piece, int cb_t) {
        const __bf16* p = wp_lane0 + (int64_t)cb_t * (CO * SKC) + (int64_t)(chunk * P + piece) * 9 * tap_stride;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (CT && (t / 3 == 0 || t % 3 == 0)) continue;      // sub-pixel ConvTranspose: taps (ky, kx) in {1, 2}^2 only
            a[t] = *reinterpret_cast<const bf16x8*>(p + t * tap_stride);
        }
    };

    f32x16 acc[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[rr][q] = 0.f;

    const int b_row = WT == 32 ? wr * R : 2 * wr * R + (r >> 4), b_col = WT == 32 ? r : (r & 15);     // the lane's pixel of the wave's first MFMA row
    const int b_lane = px_off(b_row, b_col, h);
    // MFMAs of weight piece PA against the input pieces pb < P - PA: items (input row ro, pb), fragments read one item ahead
    // 16-byte staging, weight piece 1: the next chunk's tile is split and stored to LDS buffer cbuf BETWEEN the MFMAs of the
    // four middle items, one of the lane's four pixels each (about 7 VALU instructions per MFMA: they issue while the matrix pipe works
    // on the wave's own MFMA).  As one block in front of the barrier the ~200 instructions cost 12 % of the kernel: both workgroups of a
    // CU run in step, so neither covered the other's commit phase (ablation builds, tools/build_ablate_split.sh).
    auto mfmas = [&](auto pa_tag, const bf16x8 (&a)[9], int buf, int cbuf, int cnext, uint32_t voff_next, auto set_tag, auto plan) {
        constexpr int PA = decltype(pa_tag)::value;
        constexpr int NPB = P - PA;
        constexpr int NU = RS * R + 2;                            // input rows (of the lane's row phase) the wave's MFMA rows read
        constexpr int NIT = NU * NPB;
        typedef decltype(plan) PL;
        constexpr int CN = PL::I0 < 0 ? (PA == 1 ? 4 : 0) : PL::N;                    // pixels committed in this step
        constexpr int IT0 = PL::I0 < 0 ? (NIT - 4) / 2 : PL::I0;
        constexpr int CJ0 = PL::I0 < 0 ? 0 : PL::J0;
        static_assert(IT0 >= 0 && IT0 + CN <= NIT, "commit plan");
        const unsigned char* bp = lds + buf * P * SIN_BYTES + b_lane;
        bf16x8 b[2][3];
#pragma unroll
        for (int kx = CT ? 1 : 0; kx < 3; ++kx) b[0][kx] = *reinterpret_cast<const bf16x8*>(bp + kx * 16);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT && !(SSTEM_SPLIT_ABLATE & 1)) {
                const int ro1 = (it + 1) / NPB, pb1 = (it + 1) % NPB;
#pragma unroll
                for (int kx = CT ? 1 : 0; kx < 3; ++kx)
                    b[(it + 1) & 1][kx] = *reinterpret_cast<const bf16x8*>(bp + pb1 * SIN_BYTES + (ro1 * PITCH + kx) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ro = it / NPB;
            const bool slice = VEC && it >= IT0 && it < IT0 + CN;
            if (slice && !(SSTEM_SPLIT_ABLATE & 2)) commit_px_v(cbuf, CJ0 + it - IT0, cnext, voff_next, set_tag);   // unconditional: behind the last chunk it stores stale values nobody reads
#pragma unroll
            for (int kx = CT ? 1 : 0; kx < 3; ++kx) {
#pragma unroll
                for (int ky = CT ? 1 : 0; ky < 3; ++ky) {
                    const int d = ro - ky;                        // input row ro feeds MFMA row d / RS through tap row ky
                    if (d >= 0 && d % RS == 0 && d / RS < R) {
                        if constexpr (F16)
                            acc[d / RS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ky * 3 + kx]),
                                                                                __builtin_bit_cast(f16x8, b[(SSTEM_SPLIT_ABLATE & 1) ? 0 : (it & 1)][kx]),
                                                                                acc[d / RS], 0, 0, 0);
                        else
                            acc[d / RS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky * 3 + kx], b[(SSTEM_SPLIT_ABLATE & 1) ? 0 : (it & 1)][kx],
                                                                                 acc[d / RS], 0, 0, 0);
                    }
                }
            }
            if (slice) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);          // seven VALU
                }
                __builtin_amdgcn_sched_group_barrier(0x200, TAIL ? 2 * P : P, 0);   // the pixel's LDS stores
            }
        }
    };

    // the tap-row chunk: one fragment per (input row, piece) covers the three tap columns, a[ky] holds tap row ky
    const int b_lane_tail = (b_row * PW + b_col) * 8 + h * 16;
    auto mfmas_tail = [&](auto pa_tag, const bf16x8 (&a)[9], int buf) {
        constexpr int PA = decltype(pa_tag)::value;
        constexpr int NPB = P - PA;
        constexpr int NU = RS * R + 2;
        const unsigned char* bp = lds + buf * P * SIN_BYTES + b_lane_tail;
#pragma unroll
        for (int ro = 0; ro < NU; ++ro) {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const uint64_t* q = reinterpret_cast<const uint64_t*>(bp + pb * SIN_BYTES + ro * PW * 8);
                u64x2v v;
                v[0] = q[0];
                v[1] = h ? 0ull : q[1];            // K 12..15 belong to the pixel three to the right: outside the window (and 0 x inf = NaN)
                const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = ro - ky;
                    if (d >= 0 && d % RS == 0 && d / RS < R) {
                        if constexpr (F16)
                            acc[d / RS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ky]), __builtin_bit_cast(f16x8, b), acc[d / RS], 0, 0, 0);
                        else
                            acc[d / RS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky], b, acc[d / RS], 0, 0, 0);
                    }
                }
            }
        }
    };

    bf16x8 a0[9], a1[9];
    if constexpr (!DEEP) {
        if constexpr (VEC) issue_in_v(c_first, in_n, vvoff, S0()); else issue_in(c_first);
        load_a(a0, c_first, 0, cb);
        if constexpr (VEC) commit_in_v(0, c_first); else commit_in(0, c_first);
        __syncthreads();
    }

    // one step = (chunk c, weight piece PA): settle this step's fragments, request the next step's, run the MFMAs
    auto step = [&](auto pa_tag, int c, const bf16x8 (&acur)[9], bf16x8 (&anxt)[9]) {
        constexpr int PA = decltype(pa_tag)::value;
        const bool more = (c + 1 < c_end);
        const int buf = (c - c_first) & 1;
#if SSTEM_SPLIT_AUTOWAIT
        // the weight fragments are requested in FRONT of the next chunk's tile: loads return in order, so the wait for the fragments at
        // the next step's first MFMA (the compiler's, a vmcnt(n) with the tile's loads still counted) leaves the tile's loads in flight
        // until the staging commit in the middle of that step reads them
        if (!(SSTEM_SPLIT_ABLATE & 8)) {
            if constexpr (PA + 1 < P) load_a(anxt, c, PA + 1, cb);
            else if (more) load_a(anxt, c + 1, 0, cb);
        }
        if constexpr (PA == 0) { if (more && !(SSTEM_SPLIT_ABLATE & 4)) { if constexpr (VEC) issue_in_v(c + 1, in_n, vvoff, S0()); else issue_in(c + 1); } }
#else
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): acur (requested a step ago)
        if constexpr (PA == 0) { if (more && !(SSTEM_SPLIT_ABLATE & 4)) { if constexpr (VEC) issue_in_v(c + 1, in_n, vvoff, S0()); else issue_in(c + 1); } }
        if (!(SSTEM_SPLIT_ABLATE & 8)) {
            if constexpr (PA + 1 < P) load_a(anxt, c, PA + 1, cb);
            else if (more) load_a(anxt, c + 1, 0, cb);
        }
#endif
        mfmas(pa_tag, (SSTEM_SPLIT_ABLATE & 8) ? a0 : acur, buf, buf ^ 1, c + 1, vvoff, S0(), CommitPlan<-1, 0, 0>());
        if constexpr (PA + 1 == P) {
            if constexpr (!VEC) { if (more && !(SSTEM_SPLIT_ABLATE & 2)) commit_in(buf ^ 1, c + 1); }
            __syncthreads();
        }
    };
    typedef std::integral_constant<int, 0> T0;
    typedef std::integral_constant<int, 1> T1;
    typedef std::integral_constant<int, 2> T2;
    float vmax = 0.f;            // largest magnitude this lane stores (out_amax)
    auto epilogue = [&]() __attribute__((always_inline)) {
    // DEEP runs the epilogue inside its tile loop: without these two opaque copies the optimiser hoists everything that does not
    // depend on the tile (per-channel multipliers and offsets of all 16 accumulator registers: ~300 VGPRs, i.e. spills) out of the loop
    int cb_e = cb, descale_e = descale;
    if constexpr (DEEP) { cb_e = __builtin_amdgcn_readfirstlane(cb_e); asm volatile("" : "+s"(cb_e)); asm volatile("" : "+v"(descale_e)); }
    // ---- acc[rr][q] = out[co = cb*CO + wco*32 + (q&3) + 8*(q>>2) + 4*h][y = Y0 + wr*R + rr][x = X0 + r]
    // F16: both power-of-two scales leave the sums here (exact) -- except on the whole-tile store path, which folds them into the channel's
    // multiplier
    auto descale_acc = [&]() __attribute__((always_inline)) {
        if constexpr (F16) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[rr][q] = __builtin_ldexpf(acc[rr][q], descale_e);
        }
    };
    const int x = X0 + (WT == 32 ? r : (r & 15));
    const int yl = WT == 32 ? 0 : (r >> 4);                      // the lane's image row inside its MFMA row
    // Output addressing.  NCHW: element (n, co, y, x) at ((n Cout + co) H + y) W + x.  out_blocked (the row-segment layout the sepconv
    // apply reads, include/sstem_sepconv.h): [N][H][ceil(W/64)][Cout][64], element at (((n H + y) TX + x/64) Cout + co) 64 + x%64 --
    // the same store instructions with other strides (a tile is 32 or 16 columns wide and starts on a multiple of its width, so it
    // never crosses a 64-column segment); no K slices, residual or mask with it (the launcher sees to that).
    // CT (sub-pixel ConvTranspose): channel co' = par * C + co of this launch is parity par = 2 py + px of real channel co; the output is
    // [N, C, 2H, 2W]: real channels 4 planes apart, an input row is two output rows (4W floats), a lane's pixel two floats wide.
    const int ct_c = Cout >> 2;
    const int64_t o_ch = CT ? 4 * plane : (out_blocked ? 64 : plane);                                        // floats between channels
    const int64_t o_row = CT ? 4 * (int64_t)W : (out_blocked ? (int64_t)((W + 63) >> 6) * Cout * 64 : W);    // ... rows
    const int64_t o_img = (!CT && out_blocked) ? o_row * H : (int64_t)Cout * plane;                          // floats of one image (CT: C * 4 * plane)
    // out_img (round 4): floats between the images of `out` when that is a channel block of a larger tensor -- a producer storing
    // straight into the tensor its consumer concatenates (model_unet.py:86); 0 = the images are back to back
    const int64_t o_stride = out_img ? out_img : o_img;
    const int64_t o_x0 = CT ? 2 * (int64_t)X0 : (out_blocked ? (int64_t)(X0 >> 6) * Cout * 64 + (X0 & 63) : X0);     // the tile's first column
    const bool whole = Y0 + TROWS <= H && X0 + WT <= W && o_img * 4 < ((int64_t)1 << 32);
    const bool cpart = cb_e * CO + CO > Cout;
    if (whole) {
        // One buffer resource per image (o_img * 4 < 2^32, see `whole`) whose size is the image's exactly, the lane's byte offset per MFMA
        // row in a VGPR, the channel's in the instruction's SGPR offset (part of the range check on gfx950: tools/micro/
        // buffer_soffset_range.hip): a value costs its arithmetic and the store.  Lanes whose channel lies behind the last one (cpart) need
        // no predicate in NCHW: their offset is behind the image's last byte and the hardware drops the store (returns 0 for the
        // residual); in the row-segment layout such an offset would be another segment's, so there the channel's offset is added in a
        // VGPR and those lanes get 2^32 - 1.  Per-store predicates, 64-bit pointer arithmetic and per-lane parameter loads made this phase cost
        // 3.4 us per tile and wave -- as much as the MFMAs of four chunks (profiles/r03/m_*).
        const int co0 = cb_e * CO + wco * 32;
        const int ct_par = CT ? co0 / ct_c : 0;                                 // the wave's 32 channels share a parity (C % 32 == 0)
        const int co_a = CT ? co0 - ct_par * ct_c : co0;                        // the channel the ADDRESS is made of (parameters: co0)
        const uint32_t lane_off = (uint32_t)(((int64_t)(4 * h) * o_ch + (int64_t)(Y0 + RS * wr * R + yl) * o_row + o_x0 +
                                              (CT ? (int64_t)(ct_par >> 1) * 2 * W + (ct_par & 1) + 2 * (x - X0) : (int64_t)(x - X0))) * 4);
        uint32_t voff_rr[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) voff_rr[rr] = lane_off + (uint32_t)(RS * rr) * (uint32_t)o_row * 4u;
        const uint32_t ch_step = (uint32_t)o_ch * 4u;                          // bytes between channels
        if (ksplit > 1) {
            descale_acc();
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab + ((int64_t)ks * N + n) * Cout * plane, 0,
                                                                (int)((uint32_t)Cout * plane4), 0x00020000);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t soff = (uint32_t)(co0 + (q & 3) + 8 * (q >> 2)) * ch_step;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const float a = acc[rr][q];          // (a bit_cast of the vector element itself compiled to element 0 for every q)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, a), rs, (int)voff_rr[rr], (int)soff, 0);
                }
            }
            return;
        }
        // out == nullptr (with pool_out): the caller wants the pooled copy alone (the IFNet's first block, model_interp.py:60-61, whose
        // full-resolution result nobody reads) -- a resource of zero bytes, the range check drops the full-resolution stores
        const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out ? out + (int64_t)n * o_stride : pool_out, 0, out ? (int)(uint32_t)(o_img * 4) : 0, 0x00020000);
        // the residual and the mask are NCHW tensors (never with a blocked store): same offsets, the mask's in bytes
        const rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual ? residual + (int64_t)n * Cout * plane : out), 0,
                                                              (int)((uint32_t)Cout * plane4), 0x00020000);
        const rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(MASKED && out_mask ? out_mask + (int64_t)n * Cout * plane : (uint8_t*)out, 0,
                                                               (int)((uint32_t)Cout * (uint32_t)plane), 0x00020000);
        // the pooled copy of the output (pool_out, nullable): [N, Cout, H/2, W/2]; a channel is plane/4 floats -- the SGPR offset of the
        // full-resolution store divided by four --, odd lanes carry an offset behind the resource's end (dropped by the range check)
        constexpr bool POOLS = F16 && !CT && !MASKED && (R % 2 == 0) && WT == 32;
        const rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc(pool_out ? pool_out + (int64_t)n * Cout * (plane >> 2) : out, 0,
                                                               pool_out ? (int)((uint32_t)Cout * (uint32_t)plane) : 0, 0x00020000);
        const uint32_t pool_row = (uint32_t)(W >> 1) * 4u;
        const uint32_t pool_voff = (uint32_t)(((int64_t)(4 * h) * (plane >> 2) + (int64_t)((Y0 + wr * R) >> 1) * (W >> 1) + ((X0 + r) >> 1)) * 4);
        // mode 0: NCHW, 1: NCHW with a residual, 2: row segments
        auto store_all = [&](auto actf, auto mode_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            float vm = vmax;
            // MODE 1: the residual's values are requested a batch of channels ahead of the arithmetic that uses them (all 16 channels of the
            // lane; 4 in the walking kernel, whose next tile's fragments and staging sets are live here).  Read next to the store that
            // follows them, each channel's loads waited out a full memory latency -- the compiler cannot move them above the earlier
            // stores to `out`, which may alias the residual for all it knows: 16 exposed latencies per tile, +0.05 .. +0.4 ms per launch of
            // the flow network's residual blocks (profiles/r04/n_*).
            constexpr int QB = MODE == 1 ? (DEEP ? 4 : 16) : 1;
            float rvb[QB][R];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = (q & 3) + 8 * (q >> 2);
                const uint32_t soff = (uint32_t)(co_a + k) * ch_step;
                if constexpr (MODE == 1) {
                    if (q % QB == 0) {
#pragma unroll
                        for (int qq = 0; qq < QB; ++qq) {
                            const uint32_t sq = (uint32_t)(co_a + ((q + qq) & 3) + 8 * ((q + qq) >> 2)) * ch_step;
#pragma unroll
                            for (int rr = 0; rr < R; ++rr)
                                rvb[qq][rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)voff_rr[rr], (int)sq, 0));
                        }
                    }
                }
                // the channel's bias / scale / shift.  F16: the workgroup's table (one 8-byte LDS read per channel pair, lanes of a half
                // read one address).  Otherwise: wave-uniform addresses (two channels per q: lane halves h = 0, 1), i.e. scalar loads --
                // no per-lane loads (and no vector-memory wait) in the store phase.
                // A channel behind the last one gets 0 / 0 / 0: its lanes compute 0 and leave the output's bound alone.
                const bool dead_a = cpart && co0 + k >= Cout, dead_b = cpart && co0 + k + 4 >= Cout;            // uniform
                float bs_q = 0.f, sc_q = 1.f, sh_q = 0.f, mul_q, add_q;
                if constexpr (F16) {
                    const f32x2v pq = ptab[wco * 32 + k + 4 * h];
                    mul_q = pq[0]; add_q = pq[1];
                } else {
                    const int ci = wco * 32 + k + 4 * h;
                    bs_q = ptab3[ci]; sc_q = ptab3[CO + ci]; sh_q = ptab3[2 * CO + ci];
                    mul_q = sc_q; add_q = sh_q;
                }
                const bool dead_lane = h ? dead_b : dead_a;
                uint32_t off[R];                     // VGPR part of the offset, the SGPR part is sof
                const uint32_t sof = MODE == 2 ? 0u : soff;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    if constexpr (MODE == 2) off[rr] = dead_lane ? 0xFFFFFFFFu : voff_rr[rr] + soff; else off[rr] = voff_rr[rr];
                }
                float vv[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    float v;
                    if constexpr (F16) v = actf(__builtin_fmaf(acc[rr][q], mul_q, add_q));
                    else { v = acc[rr][q] + bs_q; v = actf(v * sc_q + sh_q); }
                    if constexpr (MASKED) {         // (NCHW: the byte offset is a quarter of the float's; the SGPR part as well)
                        if (out_mask) __builtin_amdgcn_raw_buffer_store_b8(v > 0.f ? (uint8_t)1 : (uint8_t)0, rmask, (int)(off[rr] >> 2), (int)(sof >> 2), 0);
                    }
                    if constexpr (MODE == 1) v = (v + rvb[q % QB][rr]) * res_scale;
                    if (!(SSTEM_SPLIT_ABLATE & 64)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro, (int)off[rr], (int)sof, 0);
                    asm("v_max_f32 %0, %0, |%1|" : "+v"(vm) : "v"(v));            // fmaxf(vm, fabsf(v)) without the canonicalising copy
                    vv[rr] = v;
                }
                if constexpr (POOLS && MODE == 0) {
                    // the 2 x 2 pooling that follows this layer in the reference's networks (model_interp.py:60-70 AvgPool2d, model_fusionnet.py /
                    // model_unet.py MaxPool2d), stored by the launch that holds the values: rows 2j, 2j + 1 are two of the lane's own values,
                    // the column neighbour is the next lane's (a quad permute); even lanes store.  The stand-alone kernel's arithmetic
                    // (misc_kernels.hip pool2x2_forward: first maximum in row-major order, NaN wins; (((a + b) + c) + d) * 0.25): the same bits.
                    if (pool_kind) {
#pragma unroll
                        for (int j = 0; j < R / 2; ++j) {
                            const float ax = vv[2 * j], bx = vv[2 * j + 1];
                            const float ay = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ax), 0xB1, 0xF, 0xF, false));
                            const float by = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bx), 0xB1, 0xF, 0xF, false));
                            float pv;
                            if (pool_kind == 1) {
                                pv = ax;
                                if (ay > pv || ay != ay) pv = ay;
                                if (bx > pv || bx != bx) pv = bx;
                                if (by > pv || by != by) pv = by;
                            } else {
                                pv = (((ax + ay) + bx) + by) * 0.25f;
                            }
                            // (odd lanes: an offset behind the resource's end -- set AFTER the row's offset is added, 2^32 - 1 + row wraps into range)
                            const uint32_t poff = (r & 1) ? 0xFFFFFFFFu : pool_voff + (uint32_t)j * pool_row;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, pv), rpool, (int)poff, (int)(soff >> 2), 0);
                        }
                    }
                }
            }
            vmax = vm;
        };
        typedef std::integral_constant<int, 0> M0;
        typedef std::integral_constant<int, 1> M1;
        typedef std::integral_constant<int, 2> M2;
        auto store_act = [&](auto mode_tag) __attribute__((always_inline)) {
            if (act == 1) store_all([](float v) { return v > 0.f ? v : 0.f; }, mode_tag);
            else if (act == 2) store_all([slope](float v) { return v > 0.f ? v : v * slope; }, mode_tag);
            else store_all([](float v) { return v; }, mode_tag);
        };
        if (!CT && out_blocked) store_act(M2());
        else if (residual) store_act(M1());
        else store_act(M0());
        return;
    }
    descale_acc();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int co = cb_e * CO + wco * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (co >= Cout) continue;
        if (ksplit > 1) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int y = Y0 + RS * (wr * R + rr) + yl;
                if (y < H && x < W) slab[(((int64_t)ks * N + n) * Cout + co) * plane + (int64_t)y * W + x] = acc[rr][q];
            }
            continue;
        }
        const float bs = bias ? bias[co] : 0.f;
        const float sc = scale ? scale[co] : 1.f;
        const float sh = shift ? shift[co] : 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int y = Y0 + RS * (wr * R + rr) + yl;
            if (y < H && x < W) {
                int64_t o = ((int64_t)n * Cout + co) * plane + (int64_t)y * W + x;
                if constexpr (CT) {
                    const int par = co / ct_c;
                    o = (((int64_t)n * ct_c + (co - par * ct_c)) * 2 * H + 2 * y + (par >> 1)) * 2 * W + 2 * x + (par & 1);
                }
                float v = acc[rr][q] + bs;
                v = act_s(v * sc + sh, act, slope);
                if constexpr (MASKED) { if (out_mask) out_mask[o] = v > 0.f ? 1 : 0; }
                if (residual) v = (v + residual[o]) * res_scale;
                out[(!CT && out_blocked) ? (int64_t)n * o_stride + (int64_t)y * o_row + ((int64_t)(x >> 6) * Cout + co) * 64 + (x & 63)
                                         : o + (int64_t)n * (o_stride - o_img)] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    };
    if constexpr (DEEP) {
        // ---- the tile-walking stream: step s = (tile s / cpk, chunk s % cpk); staging set and LDS buffer of a step: s & 1
        const int ntile_y = (H + TROWS - 1) / TROWS;
        const int ty0 = by * walk;
        const int T = min(walk, ntile_y - ty0);
        const int S = T * cpk;                                   // (no K slices with DEEP: cpk == nchunks, c_first == 0)
        int cs = 0, ys = Y0;                                     // chunk and tile row of step s,
        int c1 = cpk > 1 ? 1 : 0, y1 = cpk > 1 ? Y0 : Y0 + TROWS;          // ... of step s + 1
        auto advance = [&](int& c, int& y) { if (c + 1 < cpk) ++c; else { c = 0; y += TROWS; } };
        int c2 = c1, y2 = y1;                                    // ... of step s + 2
        advance(c2, y2);
        load_a(a0, 0, 0, cb);
        {
            const uint32_t v0 = tile_voff(X0, ys);
            issue_in_v(cs, in_n, v0, S0());
            if (S > 1) issue_in_v(c1, in_n, tile_voff(X0, y1), S1());
#pragma unroll
            for (int j = 0; j < 4; ++j) commit_px_v(0, j, cs, v0, S0());
        }
        __syncthreads();
        auto one = [&](int s, auto par_tag, auto opp_tag) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            // weight piece 0 against both input pieces; the weight fragments are requested in front of the tile's loads (in-order returns)
            if (!(SSTEM_SPLIT_ABLATE & 8)) load_a(a1, cs, 1, cb);
            if (s + 2 < S && !(SSTEM_SPLIT_ABLATE & 4)) issue_in_v(c2, in_n, tile_voff(X0, y2), par_tag);       // this step's set was stored a step ago
            // step s + 1's tile (loaded a step ago) is split and stored between the MFMAs of BOTH weight pieces -- three of the lane's
            // four pixels under piece 0's 18 R MFMAs, one under piece 1's 9 R: with two MFMA rows per wave, piece 1's 18 MFMAs alone
            // could hide a quarter of the commit's ~200 vector instructions (ablation: the commit cost 0.32 of 0.83 ms on 8 x 32 -> 32 at 1024^2)
            const uint32_t v1 = tile_voff(X0, y1);
            mfmas(T0(), a0, PAR, PAR ^ 1, c1, v1, opp_tag, CommitPlan<(R + 2) * 2 >= 6 ? 2 : 0, 3, 0>());
            // weight piece 1 (stale values are stored behind the last step: nobody reads them)
            if (s + 1 < S && !(SSTEM_SPLIT_ABLATE & 8)) load_a(a0, c1, 0, cb);
            mfmas(T1(), (SSTEM_SPLIT_ABLATE & 8) ? a0 : a1, PAR, PAR ^ 1, c1, v1, opp_tag, CommitPlan<1, 1, 3>());
            __syncthreads();
            cs = c1; ys = y1; c1 = c2; y1 = y2;
            advance(c2, y2);
        };
        // steps go in pairs (the staging set and the LDS buffer of a step are its parity: compile-time in each half of the pair); a tile's
        // store phase follows the step that was its last chunk, in either half (any number of chunks per tile, odd ones included)
        auto finish_tile = [&](int ytile) __attribute__((always_inline)) {
            Y0 = ytile;                                          // the tile is complete: store it, start the next one from zero
            epilogue();
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[rr][q] = 0.f;
        };
#pragma unroll 1
        for (int s = 0; s < S; s += 2) {
            {
                const bool last = cs == cpk - 1;
                const int ytile = ys;
                one(s, S0(), S1());
                if (last) finish_tile(ytile);
            }
            if (s + 1 < S) {
                const bool last = cs == cpk - 1;
                const int ytile = ys;
                one(s + 1, S1(), S0());
                if (last) finish_tile(ytile);
            }
        }
        if (out_amax)
            amax_word_update(out_amax, vmax, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), reinterpret_cast<float*>(lds));
        return;
    } else if constexpr (P == 2) {
        // TAIL: as below -- the loop's last chunk has requested the tap-row chunk's tile and first fragments and stored the tile
        const int c_loop_end = (TAIL && c_end == nchunks) ? c_end - 1 : c_end;
        for (int c = c_first; c < c_loop_end; ++c) { step(T0(), c, a0, a1); step(T1(), c, a1, a0); }
        if constexpr (TAIL) {
            if (c_end == nchunks) {
                const int c = nchunks - 1;
                const int buf = (c - c_first) & 1;
                load_a(a1, c, 1, cb);
                mfmas_tail(T0(), a0, buf);
                mfmas_tail(T1(), a1, buf);
                __syncthreads();                              // the bound's reduction below reuses the tile images' first bytes
            }
        }
    } else {
        // TAIL: the tap-row chunk (the last one, in the last K slice) runs behind the loop, out of code of its own; the loop's last chunk
        // has requested its tile and weights and stored the tile (`more`), as for any other chunk
        const int c_loop_end = (TAIL && c_end == nchunks) ? c_end - 1 : c_end;
        for (int c = c_first; c < c_loop_end; c += 2) {
            step(T0(), c, a0, a1); step(T1(), c, a1, a0); step(T2(), c, a0, a1);
            if (c + 1 < c_loop_end) { step(T0(), c + 1, a1, a0); step(T1(), c + 1, a0, a1); step(T2(), c + 1, a1, a0); }
        }
        if constexpr (TAIL) {
            if (c_end == nchunks) {
                const int c = nchunks - 1;
                const int buf = (c - c_first) & 1;
                if (((c - c_first) & 1) == 0) {               // an even number of chunks went before: the fragments are in a0 already
                } else {
#pragma unroll
                    for (int t = 0; t < 9; ++t) a0[t] = a1[t];
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);
                load_a(a1, c, 1, cb);
                mfmas_tail(T0(), a0, buf);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                load_a(a0, c, 2, cb);
                mfmas_tail(T1(), a1, buf);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                mfmas_tail(T2(), a0, buf);
            }
        }
    }

    epilogue();
    // the output's bound: one atomic per workgroup (the tile images are dead: the K loop ended with a barrier)
    if (out_amax && ksplit == 1)
        amax_word_update(out_amax, vmax, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), reinterpret_cast<float*>(lds));
}

// sum of the K slices in ascending order + the fused epilogue
__global__ __launch_bounds__(256) void conv3x3_split_splitk_epilogue(
    const float* __restrict__ slab, const float* __restrict__ bias, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int64_t total, int64_t plane, int Cout, int ksplit,
    int act, float slope, const float* __restrict__ residual, float res_scale, uint8_t* __restrict__ out_mask,
    float* __restrict__ out_amax = nullptr)
{
    float vmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
        const int co = (int)((i / plane) % Cout);
        v += bias ? bias[co] : 0.f;
        v = act_s(v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f), act, slope);
        if (out_mask) out_mask[i] = v > 0.f ? 1 : 0;
        if (residual) v = (v + residual[i]) * res_scale;
        out[i] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    __shared__ float red[4];
    if (out_amax) amax_word_update(out_amax, vmax, blockIdx.x, red);
}

inline int grid_1d_s(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

inline int split_cop(int Cout) { return Cout <= 32 ? 32 : (Cout + 63) / 64 * 64; }      // padded output channels of the packed weights

inline int64_t packed_split_elems(int Cin, int Cout, int P)
{
    const int nchunks = (Cin + SKC - 1) / SKC;
    return (int64_t)nchunks * P * 9 * split_cop(Cout) * SKC;
}

// Output channels per workgroup and K slices: a pure function of the problem size (launcher and workspace query).  64 channels per
// workgroup (two waves share every input fragment) unless that leaves the grid below SSTEM_SPLIT_CO32_BELOW workgroups: then 32
// (twice the workgroups, every wave its own channel block; default 0 = never: on the 2-sample layers of the fusion step it measured
// 3-8 % slower than the 64-channel blocks, gpurun_out r5i); K slices of whole 16-channel chunks while the grid is below 512.
struct SplitGeom { int CO, ncb, ksplit; };
inline bool split_wt16(int W)
{
    static const bool off = [] { const char* e = getenv("SSTEM_SPLIT_WT16"); return e && atoi(e) == 0; }();      // developer knob (A/B runs)
    return !off && W <= 16 && W % 4 == 0;
}
inline SplitGeom split_geom(int N, int Cin, int H, int W, int Cout)
{
    static const int co32_below = [] { const char* e = getenv("SSTEM_SPLIT_CO32_BELOW"); return e ? atoi(e) : 0; }();
    static const int min_cpk = [] { const char* e = getenv("SSTEM_SPLIT_MIN_CPK"); return e ? atoi(e) : 2; }();
    static const bool ks_off = [] { const char* e = getenv("SSTEM_CONV_KSPLIT"); return e && atoi(e) == 0; }();
    static const int ks_below = [] { const char* e = getenv("SSTEM_SPLIT_KSPLIT_BELOW"); return e ? atoi(e) : 512; }();      // (sweep: profiles/r05/n_*)
    const bool w16 = split_wt16(W);                               // 16 x 16 tiles on maps up to 16 pixels wide
    const int tw = w16 ? 16 : STW, th = w16 ? 16 : STH;
    const int64_t tiles = (int64_t)((W + tw - 1) / tw) * ((H + th - 1) / th) * N;
    const int nchunks = (Cin + SKC - 1) / SKC;
    SplitGeom g;
    g.CO = (Cout <= 32 || tiles * ((Cout + 63) / 64) < co32_below) ? 32 : 64;
    g.ncb = (Cout + g.CO - 1) / g.CO;
    int ks = 1;
    if (!ks_off)
        while (tiles * g.ncb * ks < ks_below && ks < 8 && nchunks % (ks * 2) == 0 && nchunks / (ks * 2) >= min_cpk) ks *= 2;
    g.ksplit = ks;
    return g;
}

}  // namespace

bool conv3x3_split_supported(int N, int Cin, int H, int W, int Cout) { return conv3x3_bf16_supported(N, Cin, H, W, Cout); }
// fp16 pieces: the 16-byte staging addresses eight channel planes through one buffer resource (below 2^31 bytes); planes beyond that
// take the dword staging, which wants the whole image below 2^31 bytes
bool conv3x3_split_f16_supported(int N, int Cin, int H, int W, int Cout)
{
    const int64_t plane_bytes = (int64_t)H * W * 4;
    return conv3x3_bf16_supported(N, Cin, H, W, Cout) && (plane_bytes * 8 < ((int64_t)1 << 31) || (int64_t)Cin * plane_bytes < ((int64_t)1 << 31));
}

// fp16 pieces: a 16-byte header in front of the packed image + the 1024-slot amax word of the weights behind it
constexpr int F16_HDR_ELEMS = 8, F16_TAIL_FLOATS = AMAX_SLOTS;
int64_t conv3x3_split_packed_floats(int Cin, int Cout, int pieces, int f16)
{
    return packed_split_elems(Cin, Cout, pieces) / 2 + (f16 ? F16_HDR_ELEMS / 2 + F16_TAIL_FLOATS : 0);
}

hipError_t launch_amax(const float* x, int64_t n, float* word, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int g = grid_1d_s((n + 3) / 4, 256);                 // one atomic per workgroup, one slot each
    if (g > AMAX_SLOTS) g = AMAX_SLOTS;
    hipLaunchKernelGGL(amax_kernel, dim3(g), dim3(256), 0, s, x, n, word);
    return hipGetLastError();
}

// K slices for small grids: a workgroup's chunk step is P (P + 1) / 2 times as long as the bf16 kernel's, so a slice may be as short as
// one chunk
int conv3x3_split_ksplit(int N, int Cin, int H, int W, int Cout) { return split_geom(N, Cin, H, W, Cout).ksplit; }

int64_t conv3x3_split_forward_workspace_floats(int N, int Cin, int H, int W, int Cout, int pieces, int f16)
{
    const int ks = conv3x3_split_ksplit(N, Cin, H, W, Cout);
    return conv3x3_split_packed_floats(Cin, Cout, pieces, f16) + (ks > 1 ? (int64_t)ks * N * Cout * H * W : 0);
}

hipError_t launch_pack_weights_3x3_split_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, int pieces, hipStream_t s)
{
    const int CO_f = split_cop(Cout), CO_t = split_cop(Cin);
    const int nchunks_f = (Cin + SKC - 1) / SKC, nchunks_t = (Cout + SKC - 1) / SKC;
    const int64_t n_f = wp_f ? packed_split_elems(Cin, Cout, pieces) : 0, n_t = wp_t ? packed_split_elems(Cout, Cin, pieces) : 0;
    if (pieces == 3)
        hipLaunchKernelGGL(pack_weights_3x3_split_both<3>, dim3(grid_1d_s(n_f + n_t, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wp_f),
                           reinterpret_cast<__bf16*>(wp_t), Cin, Cout, CO_f, nchunks_f, n_f, CO_t, nchunks_t, n_t, 0);
    else
        hipLaunchKernelGGL(pack_weights_3x3_split_both<2>, dim3(grid_1d_s(n_f + n_t, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wp_f),
                           reinterpret_cast<__bf16*>(wp_t), Cin, Cout, CO_f, nchunks_f, n_f, CO_t, nchunks_t, n_t, 0);
    return hipGetLastError();
}

int64_t pack_group_entry_split(int Cin, int Cout, int pieces, int64_t* out)
{
    const int CO_f = split_cop(Cout), CO_t = split_cop(Cin);
    const int nchunks_f = (Cin + SKC - 1) / SKC, nchunks_t = (Cout + SKC - 1) / SKC;
    out[3] = Cin; out[4] = Cout;
    out[5] = CO_f; out[6] = nchunks_f; out[7] = (Cout + CO_f - 1) / CO_f; out[8] = packed_split_elems(Cin, Cout, pieces);
    out[9] = CO_t; out[10] = nchunks_t; out[11] = (Cin + CO_t - 1) / CO_t; out[12] = packed_split_elems(Cout, Cin, pieces);
    return (out[8] + out[12] + 255) / 256;
}

hipError_t launch_pack_weights_3x3_split_group(const int64_t* table, int n_entries, int64_t total_blocks, int pieces, hipStream_t s)
{
    if (n_entries <= 0 || total_blocks <= 0) return hipSuccess;
    if (total_blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (pieces == 3) hipLaunchKernelGGL(pack_weights_3x3_split_group<3>, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_entries);
    else hipLaunchKernelGGL(pack_weights_3x3_split_group<2>, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_entries);
    return hipGetLastError();
}

// fp16 pieces of a recorded layer, both orientations: [header][image][amax word of the weights] in each workspace (the word of the
// first one given is the one measured into)
hipError_t launch_pack_weights_3x3_split_f16_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s)
{
    if (!wp_f && !wp_t) return hipSuccess;
    const int CO_f = split_cop(Cout), CO_t = split_cop(Cin);
    const int nchunks_f = (Cin + SKC - 1) / SKC, nchunks_t = (Cout + SKC - 1) / SKC;
    const int64_t n_f = wp_f ? packed_split_elems(Cin, Cout, 2) : 0, n_t = wp_t ? packed_split_elems(Cout, Cin, 2) : 0;
    float* word = wp_f ? wp_f + (F16_HDR_ELEMS + n_f) / 2 : wp_t + (F16_HDR_ELEMS + n_t) / 2;
    hipError_t e = hipMemsetAsync(word, 0, AMAX_SLOTS * sizeof(float), s);
    if (e != hipSuccess) return e;
    e = launch_amax(w, (int64_t)Cin * Cout * 9, word, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_weights_3x3_split_f16_both, dim3(grid_1d_s(n_f + n_t, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wp_f),
                       reinterpret_cast<__bf16*>(wp_t), word, Cin, Cout, CO_f, nchunks_f, n_f, CO_t, nchunks_t, n_t);
    return hipGetLastError();
}

int64_t pack_group_entry_split_f16(int Cin, int Cout, int64_t* out)
{
    (void)pack_group_entry_split(Cin, Cout, 2, out);
    out[14] = ((int64_t)Cin * Cout * 9 + AMAX_GROUP_ELEMS - 1) / AMAX_GROUP_ELEMS;      // blocks of the bound launch
    return ((out[8] + out[12]) / 2 + 255) / 256;                                        // one thread per weight slot (both pieces)
}

hipError_t launch_pack_weights_3x3_split_f16_group(const int64_t* table, int n_entries, int64_t total_blocks, int64_t amax_blocks,
                                                   float* bounds, hipStream_t s)
{
    if (n_entries <= 0 || total_blocks <= 0) return hipSuccess;
    if (total_blocks > 0x7fffffffLL || amax_blocks <= 0 || amax_blocks > 0x7fffffffLL || !bounds) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(bounds, 0, (size_t)n_entries * sizeof(float), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(amax_weights_group, dim3((unsigned)amax_blocks), dim3(256), 0, s, table, n_entries);
    hipLaunchKernelGGL(pack_weights_3x3_split_f16_group, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_entries);
    return hipGetLastError();
}

hipError_t launch_conv3x3_split_mfma(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                     float* out, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout,
                                     int act, float slope, int w_transposed_flipped, int pieces, hipStream_t s, const ConvExtra& ex)
{
    if (pieces != 2 && pieces != 3) return hipErrorInvalidValue;
    if (!conv3x3_split_supported(N, Cin, H, W, Cout) || ex.bn_part) return hipErrorInvalidValue;
    const bool f16 = ex.f16 != 0;
    if (f16 && !conv3x3_split_f16_supported(N, Cin, H, W, Cout)) return hipErrorInvalidValue;
    if (f16 && (pieces != 2 || !ex.in_amax)) return hipErrorInvalidValue;
    if (f16 && (ex.in_mask || ex.out_mask) && (ex.out_blocked || ex.pool_out || ex.out_img_stride)) return hipErrorInvalidValue;   // masks: plain NCHW stores
    if ((ex.out_blocked == 1 && ex.residual) || (ex.out_blocked && ex.out_mask)) return hipErrorInvalidValue;
    const SplitGeom geo = split_geom(N, Cin, H, W, Cout);
    const int CO = geo.CO, ncb = geo.ncb, nchunks = (Cin + SKC - 1) / SKC, COP = split_cop(Cout);
    const int64_t welems = packed_split_elems(Cin, Cout, pieces) + (f16 ? F16_HDR_ELEMS + 2 * F16_TAIL_FLOATS : 0);
    __bf16* wp = reinterpret_cast<__bf16*>(workspace);
    const bool prepacked = (w_transposed_flipped & 2) != 0;
    w_transposed_flipped &= 1;
    hipError_t e = hipSuccess;
    if (f16) {     // [header 16 B][packed image][amax word of the weights]
        const int64_t pelems = packed_split_elems(Cin, Cout, 2);
        float* w_word = workspace + (F16_HDR_ELEMS + pelems) / 2;
        if (!prepacked) {
            e = hipMemsetAsync(w_word, 0, AMAX_SLOTS * sizeof(float), s);
            if (e != hipSuccess) return e;
            e = launch_amax(w, (int64_t)Cin * Cout * 9, w_word, s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(pack_weights_3x3_split_f16, dim3(grid_1d_s(pelems, 256)), dim3(256), 0, s, w, wp, w_word, Cin, Cout, COP, nchunks,
                               pelems, w_transposed_flipped);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    } else if (!prepacked) {
        if (pieces == 3)
            hipLaunchKernelGGL(pack_weights_3x3_split_both<3>, dim3(grid_1d_s(welems, 256)), dim3(256), 0, s, w, wp, (__bf16*)nullptr, Cin, Cout,
                               COP, nchunks, welems, 0, 0, (int64_t)0, w_transposed_flipped);
        else
            hipLaunchKernelGGL(pack_weights_3x3_split_both<2>, dim3(grid_1d_s(welems, 256)), dim3(256), 0, s, w, wp, (__bf16*)nullptr, Cin, Cout,
                               COP, nchunks, welems, 0, 0, (int64_t)0, w_transposed_flipped);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const bool ct = ex.out_blocked == 2;                     // the sub-pixel form of a ConvTranspose2d(k3, s2, p1, op1): see the kernel
    if (ct && (!f16 || Cout % 128 != 0 || Cin % SKC != 0 || W % 4 != 0)) return hipErrorInvalidValue;
    const int64_t out_img = ex.out_img_stride;               // floats between the images of `out` (0: back to back)
    // a pooled copy: whole tiles only (H % 8 == 0, W % 32 == 0), the fp16 id, plain NCHW store without residual, never split over K
    if (ex.pool_out && (!f16 || ex.out_blocked || ex.residual || H % 8 != 0 || W % 32 != 0 || ex.pool_kind < 1 || ex.pool_kind > 2 ||
                        (int64_t)Cout * H * W * 4 >= ((int64_t)1 << 32)))      // (whole tiles of one image below 4 GiB: the buffer-resource store path)
        return hipErrorInvalidValue;
    if (!out && !ex.pool_out) return hipErrorInvalidValue;   // no full-resolution output: only next to a pooled copy
    int ksplit = (ex.out_blocked || out_img || ex.pool_out) ? 1 : geo.ksplit;      // a blocked / shuffled / strided / pooled store is the launch's own
    const int64_t out_elems = (int64_t)N * Cout * H * W;
    if (ksplit > 1 && workspace_floats < welems / 2 + (int64_t)ksplit * out_elems) ksplit = 1;
    float* slab = workspace + welems / 2;
    static const bool novec = [] { const char* e = getenv("SSTEM_BF16_NOVEC"); return e && atoi(e) != 0; }();
    // 16-byte staging; fp16 pieces: eight channel planes stay below 2^31 bytes (the staging loads' buffer resource, lanes outside at 2^31)
    const bool vec = !novec && W % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (!f16 || (int64_t)H * W * 32 < ((int64_t)1 << 31));
    const bool w16 = vec && split_wt16(W);
    const int tw = w16 ? 16 : STW, th = w16 ? 16 : STH;
    const dim3 grid((W + tw - 1) / tw, (H + th - 1) / th, (unsigned)(N * ncb * ksplit));
    if (!vec && (int64_t)Cin * H * W * 4 >= (int64_t)S_OOB) return hipErrorInvalidValue;
    static const int remap_knob = [] { const char* e = getenv("SSTEM_XCD_REMAP"); return e ? atoi(e) : 1; }();
    const int remap = (remap_knob && (int64_t)grid.x * grid.y * grid.z < ((int64_t)1 << 31)) ? 1 : 0;
    const bool masked = ex.in_mask != nullptr || ex.out_mask != nullptr;
    uint8_t* kernel_out_mask = ksplit > 1 ? nullptr : ex.out_mask;          // a launch split over K leaves the mask to its slice-sum launch
    float* kernel_out_amax = ksplit > 1 ? nullptr : ex.out_amax;            // ... and the output's bound as well
    int lds_bytes = 2 * pieces * SIN_BYTES + 256 * 16 + 64 * 12;      // piece images, parking slots, the per-channel table of the store phase
#if SSTEM_SPLIT_ABLATE
    if (const char* pad = getenv("SSTEM_SPLIT_LDS_PAD")) lds_bytes += atoi(pad);          // occupancy experiments
#endif
    const bool tail = split_tail_chunk(Cin, pieces, f16);      // the packing's own rule
    const float* w_bound = f16 ? reinterpret_cast<const float*>(wp) : nullptr;
    const __bf16* wimg = f16 ? wp + F16_HDR_ELEMS : wp;
#define SSTEM_SPLIT_FWD_T(A, B, PP, V, M, T, TL)                                                                                  \
    do {                                                                                                                          \
        static bool done[64] = {};                                                                                                \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_split_mfma<A, B, PP, V, M, T, TL>), lds_bytes, done);           \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL((conv3x3_split_mfma<A, B, PP, V, M, T, TL>), grid, dim3(256), lds_bytes, s, in, wimg, bias, scale, shift, out, N, Cin, \
                           H, W, Cout, nchunks, ncb, act, slope, ksplit, slab, remap, ex.residual, ex.res_scale, COP, ex.in_mask,  \
                           kernel_out_mask, nullptr, nullptr, kernel_out_amax, ex.out_blocked, 1, out_img);                       \
    } while (0)
#define SSTEM_SPLIT_F16_TM(A, B, V, T, TL, M)                                                                                     \
    do {                                                                                                                          \
        static bool done[64] = {};                                                                                                \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_split_mfma<A, B, 2, V, M, T, TL, true>), lds_bytes, done);      \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL((conv3x3_split_mfma<A, B, 2, V, M, T, TL, true>), grid, dim3(256), lds_bytes, s, in, wimg, bias, scale, shift, \
                           out, N, Cin, H, W, Cout, nchunks, ncb, act, slope, ksplit, slab, remap, ex.residual, ex.res_scale, COP,  \
                           ex.in_mask, kernel_out_mask, ex.in_amax, w_bound, kernel_out_amax, ex.out_blocked, 1, out_img, ex.pool_out, ex.pool_kind); \
    } while (0)
    // (masked fp16 instances -- recorded launches, round 5 -- exist for the 16-byte staging path; the dword path keeps bf16 pieces)
#define SSTEM_SPLIT_F16_T(A, B, V, T, TL)                                                                                         \
    do { if (masked) { if constexpr (V) SSTEM_SPLIT_F16_TM(A, B, V, T, TL, true); else return hipErrorInvalidValue; }             \
         else SSTEM_SPLIT_F16_TM(A, B, V, T, TL, false); } while (0)
#define SSTEM_SPLIT_F16(A, B, V, T)                                                                                               \
    do { if (tail) SSTEM_SPLIT_F16_T(A, B, V, T, true); else SSTEM_SPLIT_F16_T(A, B, V, T, false); } while (0)
    // the tile-walking stream (DEEP): `walk` tiles down the image per workgroup, the grid's y extent shrinks accordingly
#define SSTEM_SPLIT_F16_DEEP(A, B)                                                                                                \
    do {                                                                                                                          \
        static bool done[64] = {};                                                                                                \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_split_mfma<A, B, 2, true, false, 32, false, true, true>), lds_bytes, done); \
        if (e != hipSuccess) return e;                                                                                            \
        const dim3 gridw(grid.x, (grid.y + walk - 1) / walk, grid.z);                                                             \
        hipLaunchKernelGGL((conv3x3_split_mfma<A, B, 2, true, false, 32, false, true, true>), gridw, dim3(256), lds_bytes, s, in, wimg, bias, scale, \
                           shift, out, N, Cin, H, W, Cout, nchunks, ncb, act, slope, 1, slab, remap, ex.residual, ex.res_scale, COP, \
                           nullptr, nullptr, ex.in_amax, w_bound, kernel_out_amax, ex.out_blocked, walk, out_img, ex.pool_out, ex.pool_kind); \
    } while (0)
    // which launches walk: fp16 pieces, 16-byte staging, 32-wide tiles, no tap-row chunk, no K slices, and enough workgroups left to
    // fill the chip several times over (SSTEM_SPLIT_WALK: 0 = never, n = tiles per workgroup)
    // (read at every launch -- a getenv costs 0.1 us --: tests and A/B runs flip them inside one process; SSTEM_SPLIT_WALK_MIN_WGS: the
    // smallest grid a walking launch may be left with)
    const char* env_walk = getenv("SSTEM_SPLIT_WALK");
    const char* env_walk_min = getenv("SSTEM_SPLIT_WALK_MIN_WGS");
    const int walk_knob = env_walk ? atoi(env_walk) : 8;
    const int64_t walk_min_wgs = env_walk_min ? atoi(env_walk_min) : 2048;
    int walk = 0;
    // (the 64-channel-block instance does not walk: its second staging set spills -- 256 VGPRs + 59 -- and it is power-bound: measured 3 % slower)
    if (f16 && vec && !w16 && !tail && ksplit == 1 && walk_knob > 0 && CO == 32 && !masked) {
        walk = walk_knob;
        while (walk > 1 && (int64_t)grid.x * ((grid.y + walk - 1) / walk) * grid.z < walk_min_wgs) walk >>= 1;
        if (walk < 2) walk = 0;
    }
#define SSTEM_SPLIT_FWD(A, B, PP, V, M, T)                                                                                        \
    do {                                                                                                                          \
        if constexpr (PP == 3) { if (tail) { SSTEM_SPLIT_FWD_T(A, B, PP, V, M, T, true); break; } }                               \
        SSTEM_SPLIT_FWD_T(A, B, PP, V, M, T, false);                                                                              \
    } while (0)
#define SSTEM_SPLIT_PV(A, B, PP, V)                                                                                               \
    do {                                                                                                                          \
        if constexpr (V) {                                                                                                        \
            if (w16) { if (masked) SSTEM_SPLIT_FWD(A, B, PP, true, true, 16); else SSTEM_SPLIT_FWD(A, B, PP, true, false, 16); break; } \
        }                                                                                                                         \
        if (masked) SSTEM_SPLIT_FWD(A, B, PP, V, true, 32); else SSTEM_SPLIT_FWD(A, B, PP, V, false, 32);                        \
    } while (0)
#define SSTEM_SPLIT_SHAPE(A, B)                                                                          \
    do {                                                                                                 \
        if (pieces == 3) { if (vec) SSTEM_SPLIT_PV(A, B, 3, true); else SSTEM_SPLIT_PV(A, B, 3, false); } \
        else { if (vec) SSTEM_SPLIT_PV(A, B, 2, true); else SSTEM_SPLIT_PV(A, B, 2, false); }           \
    } while (0)
#define SSTEM_SPLIT_F16_CT()                                                                                                       \
    do {                                                                                                                          \
        static bool done[64] = {};                                                                                                \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_split_mfma<2, 2, 2, true, false, 32, false, true, false, true>), lds_bytes, done); \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL((conv3x3_split_mfma<2, 2, 2, true, false, 32, false, true, false, true>), grid, dim3(256), lds_bytes, s, in, wimg, bias, \
                           scale, shift, out, N, Cin, H, W, Cout, nchunks, ncb, act, slope, 1, slab, remap, ex.residual, ex.res_scale, COP, \
                           nullptr, nullptr, ex.in_amax, w_bound, kernel_out_amax, 2, 1, out_img);                                \
    } while (0)
    // ... and its tile-walking form (4 of 9 taps: 12 MFMAs per MFMA row, chunk and piece -- steps as short as the 32-channel block's)
#define SSTEM_SPLIT_F16_CT_DEEP()                                                                                                  \
    do {                                                                                                                          \
        static bool done[64] = {};                                                                                                \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_split_mfma<2, 2, 2, true, false, 32, false, true, true, true>), lds_bytes, done); \
        if (e != hipSuccess) return e;                                                                                            \
        const dim3 gridw(grid.x, (grid.y + ctwalk - 1) / ctwalk, grid.z);                                                         \
        hipLaunchKernelGGL((conv3x3_split_mfma<2, 2, 2, true, false, 32, false, true, true, true>), gridw, dim3(256), lds_bytes, s, in, wimg, bias, \
                           scale, shift, out, N, Cin, H, W, Cout, nchunks, ncb, act, slope, 1, slab, remap, ex.residual, ex.res_scale, COP, \
                           nullptr, nullptr, ex.in_amax, w_bound, kernel_out_amax, 2, ctwalk, out_img);                           \
    } while (0)
    if (ct) {
        if (!vec || w16 || tail || CO != 64) return hipErrorInvalidValue;
        const char* env_ctw = getenv("SSTEM_SPLIT_WALK_CT");             // tiles a sub-pixel ConvTranspose workgroup walks (0 = per-tile kernel)
        int ctwalk = env_ctw ? atoi(env_ctw) : 8;
        while (ctwalk > 1 && (int64_t)grid.x * ((grid.y + ctwalk - 1) / ctwalk) * grid.z < 2048) ctwalk >>= 1;
        if (ctwalk >= 2) SSTEM_SPLIT_F16_CT_DEEP(); else SSTEM_SPLIT_F16_CT();
        return hipGetLastError();
    }
#undef SSTEM_SPLIT_F16_CT_DEEP
#undef SSTEM_SPLIT_F16_CT
#if SSTEM_SPLIT_DEV      // developer builds (minutes of compile time less): only the 16-byte-staging fp16 instances of 32-wide tiles
    if (!f16 || w16 || !vec) return hipErrorInvalidValue;
    if (walk) SSTEM_SPLIT_F16_DEEP(1, 4);
    else if (CO == 64) SSTEM_SPLIT_F16(2, 2, true, 32); else SSTEM_SPLIT_F16(1, 4, true, 32);
#else
    if (walk) SSTEM_SPLIT_F16_DEEP(1, 4);
    else if (f16) {
        if (CO == 64) { if (w16) SSTEM_SPLIT_F16(2, 2, true, 16); else if (vec) SSTEM_SPLIT_F16(2, 2, true, 32); else SSTEM_SPLIT_F16(2, 2, false, 32); }
        else { if (w16) SSTEM_SPLIT_F16(1, 4, true, 16); else if (vec) SSTEM_SPLIT_F16(1, 4, true, 32); else SSTEM_SPLIT_F16(1, 4, false, 32); }
    } else if (CO == 64) SSTEM_SPLIT_SHAPE(2, 2); else SSTEM_SPLIT_SHAPE(1, 4);
#endif
#undef SSTEM_SPLIT_F16_DEEP
#undef SSTEM_SPLIT_F16
#undef SSTEM_SPLIT_F16_T
#undef SSTEM_SPLIT_F16_TM
#undef SSTEM_SPLIT_SHAPE
#undef SSTEM_SPLIT_PV
#undef SSTEM_SPLIT_FWD
#undef SSTEM_SPLIT_FWD_T
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    const int eg = grid_1d_s(out_elems, 256);
    hipLaunchKernelGGL(conv3x3_split_splitk_epilogue, dim3(eg), dim3(256), 0, s, slab, bias, scale, shift, out,
                       out_elems, (int64_t)H * W, Cout, ksplit, act, slope, ex.residual, ex.res_scale, ex.out_mask, ex.out_amax);
    return hipGetLastError();
}

}  // namespace sstem
