// 3x3 convolution blocks with bf16 operands on the gfx950 matrix cores (BASELINE config 5: "bf16 activations with fp32
// sepconv accumulate").  Opt-in algorithm id SSTEM_CONV_MFMA_BF16: the tensors in HBM stay fp32 (so BatchNorm, the
// up-sampling, the sepconv op and the optimiser see what they see on the fp32 path); activations and weights are rounded to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged, products are exact, sums are fp32
// (v_mfma_f32_32x32x16_bf16: 16x the MAC rate of the fp32 MFMA used by conv_kernels.hip).
//
// Same layers as conv_kernels.hip (Conv2d 3x3 s1 p1 [+ folded BatchNorm affine] [+ ReLU | LeakyReLU], and with
// transposed + flipped weights the data gradient / the zero-insert ConvTranspose): model_interp.py:121-143,
// networks.py:179-186, model_unet.py:11-48, model_fusionnet.py:12-43.
//
// conv3x3_bf16_mfma<WCO, WR>: D[co][pixel] += W[co][k] * In[k][pixel], one MFMA = 32 co x 32 pixels x 16 input channels of one tap.
//   workgroup = 4 waves = WCO (output-channel blocks of 32) x WR (row groups); output tile = 8 rows x 32 columns x 32*WCO channels;
//   K walks chunks of 16 input channels.  Operand layouts follow the instruction: lane (r = lane & 31, h = lane >> 5) supplies
//   8 consecutive k of row r (A: output channel, B: pixel), so
//     * the input tile sits in LDS channel-last, [10 rows][34 columns][16 channels] bf16 = 32 B per pixel: a B fragment is one
//       ds_read_b128 at (pixel * 32 + h * 16), 64 lanes covering 1 KiB contiguously (conflict-free), with the tap as an immediate
//       offset; an input row's fragment is read once per kx and used by the (up to) three output rows it contributes to;
//     * the weights are pre-packed [co block][chunk][tap][co][16 channels] bf16 and never touch LDS: each wave owns its 32 output
//       channels, so its nine A fragments per chunk are nine global_load_dwordx4 (1 KiB contiguous per wave), prefetched one
//       chunk ahead into registers.
//   Staging: one wave-instruction loads one channel of 64 consecutive tile pixels (buffer load: per-lane pixel offset computed
//   once, the channel as a scalar offset, padding lanes point past the buffer and read 0); eight channels are packed into one
//   ds_write_b128.  Loads of chunk c+1 are in flight during the MFMAs of chunk c.
//   Epilogue in registers as on the fp32 path (+ bias, * scale + shift, activation), or raw split-K partial sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_kernels.h"

namespace sstem {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BKC = 16;                       // input channels per K chunk
constexpr int BTH = 8, BTW = 32;              // output tile (rows x columns)
constexpr int BIN_R = BTH + 2, BIN_PW = BTW + 2;
constexpr int BIN_PX = BIN_R * BIN_PW;        // 340 tile pixels
constexpr int BIN_BYTES = BIN_PX * 32;        // 10880 B per buffer
constexpr uint32_t OOB = 0x80000000u;         // per-lane offset of a padding lane: beyond any buffer this kernel accepts

__device__ __forceinline__ float act_bf(float v, int act, float slope)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

// W[co][ci][3][3] (or W[ci][co][3][3] read with flipped taps) -> Wp[cb][chunk][tap][CO][16] bf16, zero-padded in co and ci
__global__ void pack_weights_3x3_bf16(const float* __restrict__ w, __bf16* __restrict__ wp, int Cin, int Cout, int CO,
                                      int nchunks, int ncb, int transposed_flipped)
{
    const int64_t total = (int64_t)ncb * nchunks * 9 * CO * BKC;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cl = idx % BKC;
        int64_t r = idx / BKC;
        const int col = r % CO; r /= CO;
        const int tap = r % 9; r /= 9;
        const int chunk = r % nchunks;
        const int cb = r / nchunks;
        const int ci = chunk * BKC + cl, co = cb * CO + col;
        float v = 0.f;
        if (ci < Cin && co < Cout)
            v = transposed_flipped ? w[((int64_t)ci * Cout + co) * 9 + (8 - tap)] : w[((int64_t)co * Cin + ci) * 9 + tap];
        wp[idx] = (__bf16)v;
    }
}

template <int WCO, int WR>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_mfma(
    const float* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int act, float slope, int ksplit, float* __restrict__ slab)
{
    static_assert(WCO * WR == 4, "four waves");
    constexpr int CO = 32 * WCO, R = BTH / WR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BIN_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int wco = wave % WCO, wr = wave / WCO;
    const int X0 = blockIdx.x * BTW, Y0 = blockIdx.y * BTH;
    const int ks = blockIdx.z % ksplit;
    const int zb = blockIdx.z / ksplit;
    const int n = zb / ncb, cb = zb % ncb;
    const int cpk = nchunks / ksplit;
    const int c_first = ks * cpk, c_end = c_first + cpk;
    const int64_t plane = (int64_t)H * W;
    const uint32_t plane4 = (uint32_t)plane * 4u;

    // staging: 12 wave-items (2 channel halves x 6 groups of 64 tile pixels), 3 per wave; the half is uniform per item
    const rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)n * Cin * plane), 0,
                                                         (int)((uint32_t)Cin * plane4), 0x00020000);
    uint32_t voff[3];
    int lds_off[3];                      // byte offset of this lane's 16-B slot in the tile, or -1
    int half_of[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int wi = wave * 3 + k;
        const int half = wi / 6;
        const int px = (wi % 6) * 64 + lane;
        const int row = px / BIN_PW, col = px - row * BIN_PW;
        const int y = Y0 - 1 + row, x = X0 - 1 + col;
        const bool inside = px < BIN_PX && y >= 0 && y < H && x >= 0 && x < W;
        voff[k] = inside ? (uint32_t)(y * W + x) * 4u : OOB;
        lds_off[k] = px < BIN_PX ? px * 32 + half * 16 : -1;
        half_of[k] = half;
    }

    float stg[3][8];
    auto issue_in = [&](int chunk) {
        const int cl_lim = Cin - chunk * BKC;                    // channels left from this chunk on (uniform)
        const uint32_t sbase = (uint32_t)(chunk * BKC) * plane4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = half_of[k] * 8 + i;
                float v = 0.f;
                if (c < cl_lim)                                  // uniform: the last chunk of a ragged channel count
                    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, (int)voff[k], (int)(sbase + (uint32_t)c * plane4), 0));
                stg[k][i] = v;
            }
        }
    };
    auto commit_in = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bf16x8 pk;
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = (__bf16)stg[k][i];
            if (lds_off[k] >= 0) *reinterpret_cast<bf16x8*>(lds + buf * BIN_BYTES + lds_off[k]) = pk;
        }
    };

    // weights of this wave's 32 output channels: fragment (chunk, tap) = 16 B per lane at [tap][co = wco*32 + r][h*8 ..]
    const __bf16* wp_lane = wp + ((int64_t)cb * nchunks * 9 * CO + wco * 32 + r) * BKC + h * 8;
    auto load_a = [&](bf16x8 (&a)[9], int chunk) {
        const __bf16* p = wp_lane + (int64_t)chunk * 9 * CO * BKC;
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = *reinterpret_cast<const bf16x8*>(p + t * CO * BKC);
    };

    f32x16 acc[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[rr][q] = 0.f;

    const int b_lane = ((wr * R) * BIN_PW + r) * 32 + h * 16;
    auto mfmas = [&](const bf16x8 (&a)[9], int buf) {
        const unsigned char* bp = lds + buf * BIN_BYTES + b_lane;
#pragma unroll
        for (int ro = 0; ro < R + 2; ++ro) {                     // input row of this wave's row group
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(bp + (ro * BIN_PW + kx) * 32);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int rr = ro - ky;
                    if (rr >= 0 && rr < R)
                        acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky * 3 + kx], b, acc[rr], 0, 0, 0);
                }
            }
        }
    };

    bf16x8 a0[9], a1[9];
    issue_in(c_first);
    load_a(a0, c_first);
    commit_in(0);
    __syncthreads();

    auto body = [&](int c, const bf16x8 (&acur)[9], bf16x8 (&anxt)[9]) {
        const bool more = (c + 1 < c_end);
        const int buf = (c - c_first) & 1;
        if (more) { issue_in(c + 1); load_a(anxt, c + 1); }
        mfmas(acur, buf);
        if (more) commit_in(buf ^ 1);
        __syncthreads();
    };
    for (int c = c_first; c < c_end; c += 2) {
        body(c, a0, a1);
        if (c + 1 < c_end) body(c + 1, a1, a0);
    }

    // ---- epilogue: acc[rr][q] = out[co = cb*CO + wco*32 + (q&3) + 8*(q>>2) + 4*h][y = Y0 + wr*R + rr][x = X0 + r]
    const int x = X0 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int co = cb * CO + wco * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (co >= Cout) continue;
        if (ksplit > 1) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int y = Y0 + wr * R + rr;
                if (y < H && x < W) slab[(((int64_t)ks * N + n) * Cout + co) * plane + (int64_t)y * W + x] = acc[rr][q];
            }
            continue;
        }
        const float bs = bias ? bias[co] : 0.f;
        const float sc = scale ? scale[co] : 1.f;
        const float sh = shift ? shift[co] : 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int y = Y0 + wr * R + rr;
            if (y < H && x < W) {
                float v = acc[rr][q] + bs;
                v = v * sc + sh;
                out[((int64_t)n * Cout + co) * plane + (int64_t)y * W + x] = act_bf(v, act, slope);
            }
        }
    }
}

// Sum of the K slices in ascending order + the fused epilogue (same arithmetic as conv3x3_splitk_epilogue of the fp32 path).
__global__ __launch_bounds__(256) void conv3x3_bf16_splitk_epilogue(
    const float* __restrict__ slab, const float* __restrict__ bias, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int64_t total, int64_t plane, int Cout, int ksplit,
    int act, float slope)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
        const int co = (int)((i / plane) % Cout);
        v += bias ? bias[co] : 0.f;
        v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
        out[i] = act_bf(v, act, slope);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static inline int grid_1d_bf(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

int conv3x3_bf16_co_block(int Cout) { return Cout <= 32 ? 32 : 64; }

static inline int64_t packed_bf16_elems(int Cin, int Cout)
{
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + BKC - 1) / BKC;
    return (int64_t)ncb * nchunks * 9 * CO * BKC;
}

int64_t conv3x3_bf16_packed_floats(int Cin, int Cout) { return packed_bf16_elems(Cin, Cout) / 2; }

bool conv3x3_bf16_supported(int N, int Cin, int H, int W, int Cout)
{
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO;
    // 32-bit byte offsets inside one image of the input (buffer addressing, padding lanes at 2^31)
    return Cin > 0 && (int64_t)Cin * H * W * 4 < (int64_t)OOB && (int64_t)N * ncb * 8 <= 65535;
}

// K slices for small grids (the rule of conv3x3_ksplit, on 16-channel chunks)
int conv3x3_bf16_ksplit(int N, int Cin, int H, int W, int Cout)
{
    static const bool off = [] { const char* e = getenv("SSTEM_CONV_KSPLIT"); return e && atoi(e) == 0; }();
    if (off) return 1;
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + BKC - 1) / BKC;
    const int64_t wgs = (int64_t)((W + BTW - 1) / BTW) * ((H + BTH - 1) / BTH) * N * ncb;
    int ks = 1;
    while (wgs * ks < 512 && ks < 8 && nchunks % (ks * 2) == 0 && nchunks / (ks * 2) >= 2) ks *= 2;
    return ks;
}

int64_t conv3x3_bf16_forward_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const int ks = conv3x3_bf16_ksplit(N, Cin, H, W, Cout);
    return packed_bf16_elems(Cin, Cout) / 2 + (ks > 1 ? (int64_t)ks * N * Cout * H * W : 0);
}

hipError_t launch_conv3x3_bf16_mfma(const float* in, const float* w, const float* bias, const float* scale,
                                    const float* shift, float* out, float* workspace, int64_t workspace_floats, int N,
                                    int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                                    hipStream_t s)
{
    if (!conv3x3_bf16_supported(N, Cin, H, W, Cout)) return hipErrorInvalidValue;
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + BKC - 1) / BKC;
    const int64_t welems = packed_bf16_elems(Cin, Cout);
    __bf16* wp = reinterpret_cast<__bf16*>(workspace);
    hipLaunchKernelGGL(pack_weights_3x3_bf16, dim3(grid_1d_bf(welems, 256)), dim3(256), 0, s, w, wp, Cin, Cout, CO, nchunks,
                       ncb, w_transposed_flipped);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int ksplit = conv3x3_bf16_ksplit(N, Cin, H, W, Cout);
    const int64_t out_elems = (int64_t)N * Cout * H * W;
    if (ksplit > 1 && workspace_floats < welems / 2 + (int64_t)ksplit * out_elems) ksplit = 1;
    float* slab = workspace + welems / 2;
    const dim3 grid((W + BTW - 1) / BTW, (H + BTH - 1) / BTH, (unsigned)(N * ncb * ksplit));
    if (CO == 64)
        hipLaunchKernelGGL((conv3x3_bf16_mfma<2, 2>), grid, dim3(256), 0, s, in, wp, bias, scale, shift, out, N, Cin, H, W, Cout,
                           nchunks, ncb, act, slope, ksplit, slab);
    else
        hipLaunchKernelGGL((conv3x3_bf16_mfma<1, 4>), grid, dim3(256), 0, s, in, wp, bias, scale, shift, out, N, Cin, H, W, Cout,
                           nchunks, ncb, act, slope, ksplit, slab);
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    hipLaunchKernelGGL(conv3x3_bf16_splitk_epilogue, dim3(grid_1d_bf(out_elems, 256)), dim3(256), 0, s, slab, bias, scale,
                       shift, out, out_elems, (int64_t)H * W, Cout, ksplit, act, slope);
    return hipGetLastError();
}

}  // namespace sstem
