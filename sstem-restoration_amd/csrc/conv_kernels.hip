// Dense convolution blocks of the correction / fusion / kernel-prediction networks for MI355X.
//
// Replaces what the reference delegates to cuDNN through torch.nn (no fused ops exist there):
//   Conv2d(3x3, s1, p1) [+ BatchNorm2d(eval) affine] + ReLU | LeakyReLU(0.2)
//     sff_scripts_interp/model/model_interp.py:121-143, sp_scripts_train/networks.py:179-186,
//     sff_scripts_fusion/model/model_unet.py:11-48, model_fusionnet.py:12-43
//   ConvTranspose2d(3x3, s2, p1, op1)      model_unet.py:32,70, model_fusionnet.py:21-27
//   Conv2d(1x1)                            networks.py:238
//
// conv3x3_mfma: implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: one fmaf per product,
// k-ordered), D[co][pixel] += W[co][k] * In[k][pixel], k = (ci, ky, kx).
//   workgroup = 4 waves; output tile = 8 rows x 32 cols x (32*COT) output channels;
//   wave w owns rows 2w, 2w+1: COT x 2 accumulator tiles of 32x32 (16 VGPRs each);
//   K is walked in chunks of 8 input channels (72 k-values = 36 MFMA k-steps): the two k of one
//   MFMA are (ci, ky, kx) and (ci+4, ky, kx), so both lane halves address LDS with the same
//   immediate offset from a per-lane base;
//   LDS (double-buffered): input tile [8][10][34] + packed weights [72][32*COT]; global->register
//   loads of chunk c+1 are issued before the MFMAs of chunk c and written to LDS after them.
//   Epilogue fused in registers: + bias, * scale + shift (folded BatchNorm), ReLU / LeakyReLU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "conv_kernels.h"

namespace sstem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 8;                 // input channels per K chunk
constexpr int KK = KC * 9;            // 72 k-values per chunk
constexpr int TH = 8, TW = 32;        // output tile (rows x cols)
constexpr int IN_R = TH + 2;          // 10 input rows
constexpr int IN_PW = 34;             // 32 + 2 halo columns
constexpr int IN_TILE = KC * IN_R * IN_PW;   // 2720 floats

// "wave-uniform 64-bit base (SGPR pair) + one 32-bit per-lane byte offset" stores: the saddr form, no per-lane 64-bit addresses
typedef __attribute__((address_space(1))) float gfloat_t;
template <typename T>
__device__ __forceinline__ void pin_uniform_ptr(T*& p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ void store_lane(float* ubase, uint32_t lane_byte_off, float v)
{
    *reinterpret_cast<gfloat_t*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off) = v;
}

// (ConvExtra: conv_kernels.h)
__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

// ---- weight packing: W[co][ci][3][3] -> Wp[cb][chunk][k'][CO], k' = (cl%4)*9+ky*3+kx + 36*(cl/4)
// (zero-padded in co and ci).  For the transposed use (dgrad / zero-insert ConvTranspose) the
// source is W[ci][co][3][3] read with the taps flipped.
__global__ void pack_weights_3x3(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                 int CO, int nchunks, int ncb, int transposed_flipped)
{
    const int64_t total = (int64_t)ncb * nchunks * KK * CO;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = idx % CO;
        int64_t r = idx / CO;
        const int kp = r % KK; r /= KK;
        const int chunk = r % nchunks;
        const int cb = r / nchunks;
        const int half = kp / 36, rem = kp % 36;
        const int cl = rem / 9 + 4 * half, tap = rem % 9;
        const int ci = chunk * KC + cl, co = cb * CO + col;
        float v = 0.f;
        if (ci < Cin && co < Cout) {
            if (!transposed_flipped) v = w[((int64_t)co * Cin + ci) * 9 + tap];
            else v = w[((int64_t)ci * Cout + co) * 9 + (8 - tap)];
        }
        wp[idx] = v;
    }
}

// Both packings of one layer's weights in ONE launch (training: the forward packing and the transposed + flipped one its data
// gradient needs -- two launches per layer and step before): indices [0, n_fwd) are the forward layout, the rest the transposed one
// of the (Cout -> Cin) problem.  Same element function as pack_weights_3x3.
__global__ void pack_weights_3x3_both(const float* __restrict__ w, float* __restrict__ wp_f, float* __restrict__ wp_t, int Cin,
                                      int Cout, int CO_f, int nchunks_f, int ncb_f, int64_t n_fwd, int CO_t, int nchunks_t,
                                      int ncb_t, int64_t n_t)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const bool t = i >= n_fwd;
        const int64_t idx = t ? i - n_fwd : i;
        const int CO = t ? CO_t : CO_f, nchunks = t ? nchunks_t : nchunks_f;
        const int cin = t ? Cout : Cin, cout = t ? Cin : Cout;          // sizes of the convolution this packing serves
        const int col = idx % CO;
        int64_t r = idx / CO;
        const int kp = r % KK; r /= KK;
        const int chunk = r % nchunks;
        const int cb = r / nchunks;
        const int half = kp / 36, rem = kp % 36;
        const int cl = rem / 9 + 4 * half, tap = rem % 9;
        const int ci = chunk * KC + cl, co = cb * CO + col;
        float v = 0.f;
        if (ci < cin && co < cout) v = t ? w[((int64_t)ci * cout + co) * 9 + (8 - tap)] : w[((int64_t)co * cin + ci) * 9 + tap];
        (t ? wp_t : wp_f)[idx] = v;
    }
}

// Both packings of MANY layers in one launch (training: after the optimiser step every layer's weights have changed; one pack
// launch per layer and step was 19 launches of the 2-sample fusion step and 46 of the IFNet step).  table[e] = 16 int64: w, wp_f,
// wp_t, Cin, Cout, CO_f, nchunks_f, ncb_f, n_fwd, CO_t, nchunks_t, ncb_t, n_t, first 256-thread block of the entry (ascending), 0, 0
// -- the layout numbers come from pack_group_entry below, i.e. from the same functions the per-layer launch uses.
__global__ __launch_bounds__(256) void pack_weights_3x3_group(const int64_t* __restrict__ table, int n_entries)
{
    int lo = 0, hi = n_entries - 1;                       // last entry whose first block is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 16 + 13] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* en = table + (int64_t)lo * 16;
    const float* w = reinterpret_cast<const float*>(en[0]);
    float* wp_f = reinterpret_cast<float*>(en[1]);
    float* wp_t = reinterpret_cast<float*>(en[2]);
    const int Cin = (int)en[3], Cout = (int)en[4];
    const int64_t n_fwd = en[8], n_t = en[12];
    const int64_t i = ((int64_t)blockIdx.x - en[13]) * 256 + threadIdx.x;
    if (i >= n_fwd + n_t) return;
    const bool t = i >= n_fwd;
    const int64_t idx = t ? i - n_fwd : i;
    const int CO = (int)(t ? en[9] : en[5]), nchunks = (int)(t ? en[10] : en[6]);
    const int cin = t ? Cout : Cin, cout = t ? Cin : Cout;
    const int col = idx % CO;
    int64_t r = idx / CO;
    const int kp = r % KK; r /= KK;
    const int chunk = r % nchunks;
    const int cb = r / nchunks;
    const int half = kp / 36, rem = kp % 36;
    const int cl = rem / 9 + 4 * half, tap = rem % 9;
    const int ci = chunk * KC + cl, co = cb * CO + col;
    float v = 0.f;
    if (ci < cin && co < cout) v = t ? w[((int64_t)ci * cout + co) * 9 + (8 - tap)] : w[((int64_t)co * cin + ci) * 9 + tap];
    (t ? wp_t : wp_f)[idx] = v;
}

template <int COT, int WT = 32, int RPW = 2>
__global__ __launch_bounds__(256, COT == 1 ? 4 : 2) void conv3x3_mfma(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int act, float slope,
    int ksplit, float* __restrict__ slab, int xcd_remap, ConvExtra ex)
{
    // ex.residual (nullable, [N,Cout,H,W]): out = (act(affine(conv + bias)) + residual) * ex.res_scale -- the additive skips of
    //   the reference's blocks (model_fusionnet.py:57-61 `conv_1 + conv_2`, :129-138 `(deconv + down) / 2`) in the store.
    // ex.bn_part (nullable): train-mode BatchNorm statistics ride along -- every workgroup writes, per output channel of its
    //   tile, (count, mean, M2) of v = conv + bias over the tile's pixels inside the image (two passes over the values it holds
    //   in registers: no cancellation) to bn_part[(co * ex.bn_tiles + tile) * 3 ..]; bn_fwd_apply merges them (Chan) in double.
    // ksplit > 1 (small grids, see conv3x3_ksplit): blockIdx.z also carries a K slice; every slice walks
    // nchunks / ksplit input-channel chunks and writes its RAW partial sums to slab[ks][n][co][y][x];
    // conv3x3_splitk_epilogue adds the slices in a fixed order and applies bias / scale / shift / activation.
    // WT = 16 (maps up to 16 pixels wide: the deepest levels of the U-Nets at 256x256 inputs): the 32 pixel columns of an MFMA row
    // are TWO image rows of 16 -- a 16x16 map is one whole tile instead of a tile whose right half is padding (50 % of the MFMAs).
    constexpr int CO = 32 * COT;
    constexpr int W_TILE = KK * CO;
    // RPW = 1 (small grids): one MFMA row per wave, 4-row tiles -- twice the workgroups of the 8-row tile, so that layers with
    // 256..511 eight-row tiles fill the chip without a split over K (no slab, no slice-sum launch) and the deeper ones split less.
    constexpr int RM = 32 / WT;                       // image rows per MFMA row
    constexpr int TH = 4 * RPW * RM, TW = WT;         // output tile (image rows x columns): shadows the file-level 8 x 32
    constexpr int IN_R = TH + 2, IN_PW = WT + 2;      // input tile with its halo
    constexpr int IN_TILE = KC * IN_R * IN_PW;        // 2720 (WT 32) / 2592 (WT 16) floats
    constexpr int BUF = IN_TILE + W_TILE;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int jr = j / WT, jc = j % WT;               // image row (within the MFMA row) and column of this lane's pixel
    // XCD-aware tile order (measured on the bf16 kernel, conv_bf16_kernels.hip): workgroups go to the 8 XCDs round-robin in launch
    // order and each XCD has its own L2, so with the plain order a tile's left and right neighbours -- which share the cache lines
    // of its halo columns -- always run on other XCDs.  Re-mapped, XCD k owns a contiguous run of the order (channel block fastest:
    // the blocks of one pixel tile read the same input tile; then x, y, K slice, image).
    int bx = blockIdx.x, by = blockIdx.y, ks = blockIdx.z % ksplit, n = (blockIdx.z / ksplit) / ncb, cb = (blockIdx.z / ksplit) % ncb;
    if (xcd_remap) {
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const uint32_t k = lin & 7u, q = total >> 3, rem = total & 7u;
        uint32_t t = k * q + (k < rem ? k : rem) + (lin >> 3);                  // XCD k: q (+1 for the first rem) consecutive tiles
        cb = (int)(t % (uint32_t)ncb); t /= (uint32_t)ncb;
        bx = (int)(t % gx); t /= gx;
        by = (int)(t % gy); t /= gy;
        ks = (int)(t % (uint32_t)ksplit); n = (int)(t / (uint32_t)ksplit);
    }
    const int X0 = bx * TW, Y0 = by * TH;
    const int cpk = nchunks / ksplit;                 // chunks per K slice (nchunks % ksplit == 0, launcher)
    const int c_first = ks * cpk, c_end = c_first + cpk;
    const int64_t plane = (int64_t)H * W;

    // ---- staging maps (fixed per thread): element e = tid + 256*k of the [8][10][34] input tile reads
    // chunk_base[in_off[k]] where chunk_base = in + (n*Cin + chunk*8)*plane is wave-uniform and in_off[k] is a
    // 32-bit byte offset (8*plane*4 < 4 GiB); in_cl[k] < 0 marks zero fill (padding / outside the image).
    constexpr int IN_PER_T = (IN_TILE + 255) / 256;     // 11
    constexpr int W_V4 = W_TILE / 4;                    // float4 count: 576 (COT=1) / 1152 (COT=2)
    constexpr int W_PER_T = (W_V4 + 255) / 256;         // 3 / 5
    uint32_t in_off[IN_PER_T];
    int in_cl[IN_PER_T];
#pragma unroll
    for (int k = 0; k < IN_PER_T; ++k) {
        const int e = tid + 256 * k;
        const int cl = e / (IN_R * IN_PW);
        const int rem = e - cl * (IN_R * IN_PW);
        const int r = rem / IN_PW, cc = rem - r * IN_PW;
        const int y = Y0 - 1 + r, x = X0 - 1 + cc;
        const bool ok = (e < IN_TILE) && y >= 0 && y < H && x >= 0 && x < W;
        in_off[k] = ok ? ((uint32_t)cl * (uint32_t)plane + (uint32_t)(y * W + x)) * 4u : 0u;
        in_cl[k] = ok ? cl : -1;
    }
    const float* in_n = in + (int64_t)n * Cin * plane;
    const float* wp_cb = wp + (int64_t)cb * nchunks * W_TILE;

    float in_r[IN_PER_T];
    f32x4 w_r[W_PER_T];
    auto stage_load = [&](int chunk) {
        const float* cbase = in_n + (int64_t)chunk * KC * plane;          // uniform
        const int cl_lim = Cin - chunk * KC;                               // channels left in this chunk
#pragma unroll
        for (int k = 0; k < IN_PER_T; ++k) {
            float v = 0.f;
            if (in_cl[k] >= 0 && in_cl[k] < cl_lim)
                v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cbase) + in_off[k]);
            in_r[k] = v;
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(wp_cb + (int64_t)chunk * W_TILE);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            w_r[k] = (e < W_V4) ? src[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](int buf) {
        float* b = lds + buf * BUF;
#pragma unroll
        for (int k = 0; k < IN_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < IN_TILE) b[e] = in_r[k];
        }
        f32x4* wdst = reinterpret_cast<f32x4*>(b + IN_TILE);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < W_V4) wdst[e] = w_r[k];
        }
    };

    f32x16 acc[COT][RPW];
#pragma unroll
    for (int t = 0; t < COT; ++t)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][rr][q] = 0.f;

    stage_load(c_first);
    stage_store(0);
    __syncthreads();

    // per-lane LDS bases: B (input) = half*4 channels + wave rows + column j; A (weights) = half*36 rows + i
    const int b_base = h * (4 * IN_R * IN_PW) + ((RPW * wave) * RM + jr) * IN_PW + jc;
    const int a_base = IN_TILE + h * (36 * CO) + j;

    for (int c = c_first; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) stage_load(c + 1);
        const float* buf = lds + ((c - c_first) & 1) * BUF;
        const float* bp = buf + b_base;
        const float* ap = buf + a_base;
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int cl = s / 9, ky = (s % 9) / 3, kx = s % 3;
            float a[COT], b[RPW];
#pragma unroll
            for (int t = 0; t < COT; ++t) a[t] = ap[s * CO + t * 32];
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) b[rr] = bp[(cl * IN_R + ky + rr * RM) * IN_PW + kx];
#pragma unroll
            for (int t = 0; t < COT; ++t)
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
                    acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[rr], acc[t][rr], 0, 0, 0);
        }
        if (more) stage_store((c + 1 - c_first) & 1);
        __syncthreads();
    }

    // ---- epilogue: acc[t][rr][q] = out[co = cb*CO + t*32 + (q&3) + 8*(q>>2) + 4*h][y = Y0 + (2*wave + rr)*RM + jr][x = X0 + jc]
    const int x = X0 + jc;
    const int yl = Y0 + (RPW * wave) * RM + jr;       // this lane's image row for rr = 0; rr = 1 is RM rows below
    if (ex.bn_part && ksplit == 1) {
        // batch-statistics partials of v = acc + bias for this tile (see ConvExtra): lanes outside the image do not count
        float* red = lds;                                   // [4 waves][64] floats, the main loop's buffers are dead (barrier above)
        const int rows_in = min(TH, H - Y0), cols_in = min(TW, W - X0);
        const float cnt = (float)(rows_in * cols_in);
        const int tile = (n * gridDim.y + by) * gridDim.x + bx;
        const bool xin = x < W;
#pragma unroll
        for (int t = 0; t < COT; ++t) {
            const int co0 = cb * CO + t * 32;
            float mean_q[16];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                float part[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    const float bsv = (bias && co < Cout) ? bias[co] : 0.f;
                    float sacc = 0.f;
#pragma unroll
                    for (int rr = 0; rr < RPW; ++rr) {
                        const bool in_img = xin && (yl + rr * RM) < H;
                        const float v = acc[t][rr][q] + bsv;
                        const float d = pass == 0 ? v : (v - mean_q[q]) * (v - mean_q[q]);
                        sacc += in_img ? d : 0.f;
                    }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) sacc += __shfl_xor(sacc, o, 64);     // over the 32 columns of this lane half
                    part[q] = sacc;
                }
                // lanes j == q hold the sum of "their" q after this select chain: one LDS word per (wave, h, q)
                float mine = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) mine = (j == q) ? part[q] : mine;
                if (j < 16) red[wave * 64 + h * 16 + j + 32 * 0] = mine;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float tot = ((red[0 * 64 + h * 16 + q] + red[1 * 64 + h * 16 + q]) + red[2 * 64 + h * 16 + q]) + red[3 * 64 + h * 16 + q];
                    if (pass == 0) mean_q[q] = tot / cnt;
                    else part[q] = tot;
                }
                __syncthreads();
                if (pass == 1 && wave == 0 && j < 16) {
                    float m2 = 0.f, mn = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { m2 = (j == q) ? part[q] : m2; mn = (j == q) ? mean_q[q] : mn; }
                    const int co = co0 + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (co < Cout) {
                        float* dst = ex.bn_part + ((int64_t)co * ex.bn_tiles + tile) * 3;
                        dst[0] = cnt; dst[1] = mn; dst[2] = m2;
                    }
                }
            }
        }
    }
    // Tiles wholly inside the image take the lean path (same arithmetic, same bits): one per-lane byte offset, uniform bases per
    // (q, row), per-channel constants loaded up front, the activation resolved once per workgroup.  The generic path below
    // spends ~35 instructions per stored element (bounds tests, a switch on the activation, 64-bit address arithmetic).
    const bool whole = Y0 + TH <= H && X0 + TW <= W && (int64_t)Cout * plane * 4 < ((int64_t)1 << 32);
    if (whole) {
        const bool cpart = cb * CO + CO > Cout;                                        // uniform: partial channel block
        const uint32_t plane4 = (uint32_t)plane * 4u;
        const uint32_t lane_off = (uint32_t)(4 * h) * plane4 + (uint32_t)(yl * W + x) * 4u;
#pragma unroll
        for (int t = 0; t < COT; ++t) {
            const int co0 = cb * CO + t * 32;
            if (co0 >= Cout) break;                                                    // uniform
            if (ksplit > 1) {
                float* base = slab + (((int64_t)ks * N + n) * Cout + co0) * plane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool live = !(cpart && co0 + (q & 3) + 8 * (q >> 2) + 4 * h >= Cout);  // per lane
                    float* chp = base + (int64_t)((q & 3) + 8 * (q >> 2)) * plane;
#pragma unroll
                    for (int rr = 0; rr < RPW; ++rr) {
                        float* rp = chp + rr * RM * W;
                        pin_uniform_ptr(rp);                                                      // outside the divergent store
                        if (live) store_lane(rp, lane_off, acc[t][rr][q]);
                    }
                }
                continue;
            }
            float* base = out + ((int64_t)n * Cout + co0) * plane;
            const float* rbase = ex.residual ? ex.residual + ((int64_t)n * Cout + co0) * plane : nullptr;
            // all 48 per-channel constants requested at once (ONE wait; the registers of the main loop are dead here): loading them four
            // channels at a time exposed the load latency four times per wave and cost 25 % on a 64-channel layer
            float bs[16], sc[16], sh[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (cpart && co >= Cout) co = Cout - 1;                                    // never stored: any valid index
                bs[q] = bias ? bias[co] : 0.f;
                sc[q] = scale ? scale[co] : 1.f;
                sh[q] = shift ? shift[co] : 0.f;
            }
            // settle them HERE, once: left to the compiler every predicated store block below got its own vmcnt(0) -- which on gfx9 also
            // waits for the stores issued before it, so the 64 stores of a wave went out one at a time
            __builtin_amdgcn_s_waitcnt(0x0F70);                                 // vmcnt(0)
            auto store_all = [&](auto actf) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool live = !(cpart && co0 + (q & 3) + 8 * (q >> 2) + 4 * h >= Cout);          // per lane
                    float* chp = base + (int64_t)((q & 3) + 8 * (q >> 2)) * plane;
                    const float* rchp = rbase ? rbase + (int64_t)((q & 3) + 8 * (q >> 2)) * plane : nullptr;
                    float rv[RPW] = {};
                    if (rbase) {                                                                  // uniform
#pragma unroll
                        for (int rr = 0; rr < RPW; ++rr) {
                            const float* rp = rchp + rr * RM * W;
                            pin_uniform_ptr(rp);
                            if (live) rv[rr] = *reinterpret_cast<const gfloat_t*>(reinterpret_cast<uint64_t>(rp) + lane_off);
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < RPW; ++rr) {
                        float v = acc[t][rr][q] + bs[q];
                        v = v * sc[q] + sh[q];
                        v = actf(v);
                        if (rbase) v = (v + rv[rr]) * ex.res_scale;
                        float* rp = chp + rr * RM * W;
                        pin_uniform_ptr(rp);                                                      // outside the divergent store
                        if (live) store_lane(rp, lane_off, v);
                    }
                }
            };
            if (act == 1) store_all([](float v) { return v > 0.f ? v : 0.f; });
            else if (act == 2) store_all([slope](float v) { return v > 0.f ? v : v * slope; });
            else store_all([](float v) { return v; });
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < COT; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = cb * CO + t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (co >= Cout) continue;
            if (ksplit > 1) {                          // raw partial sums of this K slice
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int y = yl + rr * RM;
                    if (y < H && x < W)
                        slab[(((int64_t)ks * N + n) * Cout + co) * plane + (int64_t)y * W + x] = acc[t][rr][q];
                }
                continue;
            }
            const float bs = bias ? bias[co] : 0.f;
            const float sc = scale ? scale[co] : 1.f;
            const float sh = shift ? shift[co] : 0.f;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int y = yl + rr * RM;
                if (y < H && x < W) {
                    float v = acc[t][rr][q] + bs;
                    v = v * sc + sh;
                    v = apply_act(v, act, slope);
                    const int64_t o = ((int64_t)n * Cout + co) * plane + (int64_t)y * W + x;
                    if (ex.residual) v = (v + ex.residual[o]) * ex.res_scale;
                    out[o] = v;
                }
            }
        }
    }
}

// Sum of the K slices (ascending slice index: fixed order, bitwise reproducible) + the fused epilogue of conv3x3_mfma.
__global__ __launch_bounds__(256) void conv3x3_splitk_epilogue(
    const float* __restrict__ slab, const float* __restrict__ bias, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int64_t total, int64_t plane, int Cout, int ksplit,
    int act, float slope, const float* __restrict__ residual, float res_scale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
        const int co = (int)((i / plane) % Cout);
        v += bias ? bias[co] : 0.f;
        v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
        v = apply_act(v, act, slope);
        if (residual) v = (v + residual[i]) * res_scale;
        out[i] = v;
    }
}

// The same sum for a convolution that feeds a train-mode BatchNorm: out = sum of slices + bias, and the batch-statistics
// partials ride along.  Workgroup = one (sample, channel, piece) chunk of up to SPLITK_BN_CHUNK consecutive floats (the chunking of
// norm_kernels.hip); it writes (count, mean, M2) of its chunk -- mean first, then M2 around it from the values it has just
// written (its own stores: same thread, same addresses) -- to bn_part[(co * (N * pieces) + n * pieces + piece) * 3 ..].
constexpr int SPLITK_BN_CHUNK = 16384;
__global__ __launch_bounds__(256) void conv3x3_splitk_epilogue_bn(
    const float* __restrict__ slab, const float* __restrict__ bias, float* __restrict__ out, int64_t total, int64_t plane,
    int Cout, int ksplit, int pieces, float* __restrict__ bn_part)
{
    __shared__ float sh[256];
    const int co = blockIdx.y, n = blockIdx.x / pieces, piece = blockIdx.x % pieces;
    const int64_t start = (int64_t)piece * SPLITK_BN_CHUNK;
    const int64_t len = plane - start < SPLITK_BN_CHUNK ? plane - start : SPLITK_BN_CHUNK;
    const int64_t base = ((int64_t)n * Cout + co) * plane + start;
    const float bs = bias ? bias[co] : 0.f;
    auto block_sum = [&](float v) -> float {
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        const float r = sh[0];
        __syncthreads();
        return r;
    };
    float s1 = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 256) {
        float v = slab[base + i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + base + i];
        v += bs;
        out[base + i] = v;
        s1 += v;
    }
    const float mean = block_sum(s1) / (float)len;
    float s2 = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 256) { const float d = out[base + i] - mean; s2 += d * d; }
    const float m2 = block_sum(s2);
    if (threadIdx.x == 0) {
        float* dst = bn_part + ((int64_t)co * gridDim.x + blockIdx.x) * 3;
        dst[0] = (float)len; dst[1] = mean; dst[2] = m2;
    }
}

// ---- direct kernels (any kernel size / the cross-check of the MFMA path) ----------------------
__global__ __launch_bounds__(256) void conv2d_direct(
    const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int KH, int KW, int PH, int PW, int act, float slope)
{
    const int64_t plane = (int64_t)H * W;
    const int64_t total = (int64_t)N * Cout * plane;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        int64_t r = idx / W;
        const int y = r % H; r /= H;
        const int co = r % Cout;
        const int n = r / Cout;
        float acc = 0.f;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* ip = in + ((int64_t)n * Cin + ci) * plane;
            const float* wq = w + ((int64_t)co * Cin + ci) * KH * KW;
            for (int ky = 0; ky < KH; ++ky) {
                const int yy = y + ky - PH;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < KW; ++kx) {
                    const int xx = x + kx - PW;
                    if (xx < 0 || xx >= W) continue;
                    acc = fmaf(ip[(int64_t)yy * W + xx], wq[ky * KW + kx], acc);
                }
            }
        }
        float v = acc + (bias ? bias[co] : 0.f);
        v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
        out[idx] = apply_act(v, act, slope);
    }
}

// ConvTranspose2d(k=3, s=2, p=1, output_padding=1): out[n,co,Y,X] = sum_ci sum_{ky,kx}
// in[n,ci,y,x] * W[ci,co,ky,kx] with Y = 2y - 1 + ky, X = 2x - 1 + kx.  Output is 2H x 2W.
__global__ __launch_bounds__(256) void convT3x3s2_direct(
    const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int act, float slope)
{
    const int OH = 2 * H, OW = 2 * W;
    const int64_t total = (int64_t)N * Cout * OH * OW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int X = idx % OW;
        int64_t r = idx / OW;
        const int Y = r % OH; r /= OH;
        const int co = r % Cout;
        const int n = r / Cout;
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = Y + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int y = ty >> 1;
            if (y >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = X + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int x = tx >> 1;
                if (x >= W) continue;
                for (int ci = 0; ci < Cin; ++ci)
                    acc = fmaf(in[(((int64_t)n * Cin + ci) * H + y) * W + x],
                               w[(((int64_t)ci * Cout + co) * 3 + ky) * 3 + kx], acc);
            }
        }
        float v = acc + (bias ? bias[co] : 0.f);
        v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
        out[idx] = apply_act(v, act, slope);
    }
}

// ---- weight gradients (first version: one workgroup per (co, ci) pair, block reduction) --------
// gW[co,ci,ky,kx] = sum_{n,y,x} g[n,co,y,x] * in[n,ci,y+ky-PH,x+kx-PW]          (Conv2d)
// gW[ci,co,ky,kx] = sum_{n,y,x} in[n,ci,y,x] * g[n,co,2y-1+ky,2x-1+kx]          (ConvTranspose 3x3 s2)
template <int MAXTAPS>
__device__ __forceinline__ void block_reduce_store(float (&part)[MAXTAPS], int taps, float* dst, int accumulate = 0)
{
    __shared__ float red[4][MAXTAPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) {
        float v = part[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < taps) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        dst[threadIdx.x] = accumulate ? dst[threadIdx.x] + v : v;
    }
}

// 1x1 weight gradient: gw[co][ci] = sum over batch and pixels of g[n][co][p] * in[n][ci][p] -- a plain dot product per
// (co, ci) pair (the OutConv of the SP U-Nets: 64 -> 1 at full resolution; the general kernel below took 2.4 ms per call
// there, with a division per pixel and 256 threads per pair).  One 1024-thread workgroup per pair, 16-byte loads when the
// plane allows, fixed-order reduction (lane partials -> wave shuffle tree -> 16 wave sums added in order).
__global__ __launch_bounds__(1024) void conv1x1_wgrad_direct(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ gw, int N, int Cin, int64_t plane, int Cout,
    int accumulate)
{
    __shared__ float wsum[16];
    const int co = blockIdx.x / Cin, ci = blockIdx.x % Cin;
    float acc = 0.f;
    const bool vec = (plane % 4) == 0;
    for (int n = 0; n < N; ++n) {
        const float* gp = g + ((int64_t)n * Cout + co) * plane;
        const float* ip = in + ((int64_t)n * Cin + ci) * plane;
        if (vec) {
            const float4* g4 = reinterpret_cast<const float4*>(gp);
            const float4* i4 = reinterpret_cast<const float4*>(ip);
            for (int64_t q = threadIdx.x; q < plane / 4; q += 1024) {
                const float4 a = g4[q], b = i4[q];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        } else {
            for (int64_t q = threadIdx.x; q < plane; q += 1024) acc = fmaf(gp[q], ip[q], acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wsum[w];
        float* dst = gw + (int64_t)co * Cin + ci;
        *dst = accumulate ? *dst + t : t;
    }
}

__global__ __launch_bounds__(256) void conv2d_wgrad_direct(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ gw,
    int N, int Cin, int H, int W, int Cout, int KH, int KW, int PH, int PW, int accumulate)
{
    const int co = blockIdx.x / Cin, ci = blockIdx.x % Cin;
    const int64_t plane = (int64_t)H * W;
    float part[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) part[t] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* gp = g + ((int64_t)n * Cout + co) * plane;
        const float* ip = in + ((int64_t)n * Cin + ci) * plane;
        for (int64_t p = threadIdx.x; p < plane; p += 256) {
            const int y = p / W, x = p - (int64_t)y * W;
            const float gv = gp[p];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                if (ky >= KH) break;
                const int yy = y + ky - PH;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    if (kx >= KW) break;
                    const int xx = x + kx - PW;
                    if (xx < 0 || xx >= W) continue;
                    part[ky * 5 + kx] = fmaf(gv, ip[(int64_t)yy * W + xx], part[ky * 5 + kx]);
                }
            }
        }
    }
    // compact 5x5 slots to KHxKW
    float outp[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) outp[t] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
            if (ky < KH && kx < KW) outp[ky * KW + kx] = part[ky * 5 + kx];
    block_reduce_store<25>(outp, KH * KW, gw + ((int64_t)co * Cin + ci) * KH * KW, accumulate);
}

__global__ __launch_bounds__(256) void convT3x3s2_wgrad_direct(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ gw,
    int N, int Cin, int H, int W, int Cout)
{
    const int ci = blockIdx.x / Cout, co = blockIdx.x % Cout;
    const int OH = 2 * H, OW = 2 * W;
    const int64_t plane = (int64_t)H * W, oplane = (int64_t)OH * OW;
    float part[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) part[t] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* gp = g + ((int64_t)n * Cout + co) * oplane;
        const float* ip = in + ((int64_t)n * Cin + ci) * plane;
        for (int64_t p = threadIdx.x; p < plane; p += 256) {
            const int y = p / W, x = p - (int64_t)y * W;
            const float iv = ip[p];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int Y = 2 * y - 1 + ky;
                if (Y < 0 || Y >= OH) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int X = 2 * x - 1 + kx;
                    if (X < 0 || X >= OW) continue;
                    part[ky * 3 + kx] = fmaf(iv, gp[(int64_t)Y * OW + X], part[ky * 3 + kx]);
                }
            }
        }
    }
    block_reduce_store<9>(part, 9, gw + ((int64_t)ci * Cout + co) * 9);
}

// data gradient of ConvTranspose 3x3 s2 p1 op1 = a stride-2 3x3 convolution of g:
// gin[n,ci,y,x] = sum_co sum_{ky,kx} g[n,co,2y-1+ky,2x-1+kx] * W[ci,co,ky,kx]
__global__ __launch_bounds__(256) void convT3x3s2_dgrad_direct(
    const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gin,
    int N, int Cin, int H, int W, int Cout)
{
    const int OH = 2 * H, OW = 2 * W;
    const int64_t total = (int64_t)N * Cin * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        int64_t r = idx / W;
        const int y = r % H; r /= H;
        const int ci = r % Cin;
        const int n = r / Cin;
        float acc = 0.f;
        for (int co = 0; co < Cout; ++co) {
            const float* gp = g + ((int64_t)n * Cout + co) * OH * OW;
            const float* wq = w + ((int64_t)ci * Cout + co) * 9;
            for (int ky = 0; ky < 3; ++ky) {
                const int Y = 2 * y - 1 + ky;
                if (Y < 0 || Y >= OH) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int X = 2 * x - 1 + kx;
                    if (X < 0 || X >= OW) continue;
                    acc = fmaf(gp[(int64_t)Y * OW + X], wq[ky * 3 + kx], acc);
                }
            }
        }
        gin[idx] = acc;
    }
}

// ---- 3x3 weight gradient on the matrix cores ---------------------------------------------------
// gW[co][ci][tap] = sum over pixels of g[co][p] * in[ci][p + tap shift]: a GEMM with M = co, N = ci (one
// 32x32 accumulator tile per tap), K = pixels.  Workgroup = 4 waves = 64 co x 64 ci (wave = one 32x32
// quadrant, 9 accumulator tiles = 144 VGPRs); it walks its share of the 2-row x 32-col pixel tiles
// (split-K over `ksplit` workgroups per (co,ci) block), staging g [64][64(+1)] and in [64][4x34(+1)]
// in LDS (odd pitches: the column reads of 32 channels are conflict-free), and finally stores its
// partial sums as whole 128-B rows into slab[ks][co][ci / 64][tap][ci % 64] (wgrad_slab_index); a second kernel adds the slabs in a
// fixed order (bitwise reproducible, no float atomics) and permutes to [co][ci][3][3].
constexpr int WT_R = 2;                        // pixel-tile rows
constexpr int G_P = WT_R * TW + 1;             // 65
constexpr int I_P = (WT_R + 2) * IN_PW + 1;    // 137

// WCO x WCI waves per workgroup, each wave one 32(co) x 32(ci) quadrant: 2x2 for wide layers, 1x1 / 2x1 /
// 1x2 when a channel count is <= 32 (no padded quadrants, 4x smaller slabs).
template <int WCO, int WCI>
__global__ __launch_bounds__(64 * WCO * WCI, 2) void conv3x3_wgrad_mfma(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ slab,
    int N, int Cin, int H, int W, int Cout, int CinP, int CoutP, int ksplit, int tiles_x, int tiles_y,
    float* __restrict__ bias_slab)
{
    // bias_slab (nullable): the bias gradient gb[co] = sum over pixels of g[co][p] rides along -- the g tiles are in LDS
    // anyway.  The workgroups of the first ci block add up their tiles (thread = one channel x one of BPARTS pixel ranges)
    // and write bias_slab[(ks * BPARTS + part)][co]; conv3x3_wgrad_reduce adds the rows in a fixed order.  Replaces one
    // torch reduction launch (a full re-read of g) per layer.
    constexpr int WG_CO = 32 * WCO, WG_CI = 32 * WCI, THREADS = 64 * WCO * WCI;
    constexpr int BPARTS = THREADS / WG_CO, BPIX = (WT_R * TW) / BPARTS;
    __shared__ float g_t[WG_CO * G_P];
    __shared__ float i_t[WG_CI * I_P];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int wi = wave / WCI, wj = wave % WCI;
    const int nib = CinP / WG_CI;
    const int blk = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
    const int cb = blk / nib, ib = blk % nib;
    const int64_t plane = (int64_t)H * W;
    const int ntiles = N * tiles_y * tiles_x;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const float* ap = g_t + (wi * 32 + j) * G_P + h;
    const float* bp = i_t + (wj * 32 + j) * I_P + h;
    const bool do_bias = (bias_slab != nullptr) && (ib == 0);      // workgroup-uniform
    const int bch = tid % WG_CO, bpart = tid / WG_CO;
    float bsum = 0.f;

    constexpr int NW = WCO * WCI;
    constexpr int G_IT = WG_CO / NW, I_IT = WG_CI * (WT_R + 2) / NW;
    // ---- staging of g [co][2 rows x 32 cols] and in [ci][4 rows x 34 cols].
    // One wave-instruction = one whole channel of the g tile (lane = row*32 + col) or one (channel, row) segment of the input
    // tile (lanes 0..33 = columns): channel and row are wave-uniform, so every address is a uniform base plus a per-lane
    // offset computed once per tile -- no per-element index arithmetic; loads are unconditional (clamped to a valid address).
    // (The first version looped over elements with a division, a branch, a dependent load and a store each: the compiler
    // waited vmcnt(0) before every store -- 50 serialised latencies per tile, MFMA pipe 39-43 % busy, 43-49 % of the wave-
    // cycles parked: profiles/r01/t_mfma_utilisation.txt.)
    // PREF (the 4-wave shape): the loads of tile t+1 are issued right after the barrier that opens tile t's MFMA phase and
    // stay in flight during it; only at the top of the next trip are they masked (the validity tests are recomputed from
    // the tile index there -- a select next to the load would make the wave wait for it) and stored to LDS.
    constexpr bool PREF = (NW == 4);
    auto geometry = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        const int tx = tile % tiles_x;
        const int r0 = tile / tiles_x;
        n = r0 / tiles_y; X0 = tx * TW; Y0 = (r0 % tiles_y) * WT_R;
    };
    // Input tile of one channel = (WT_R+2) x 34 = 136 floats, addressed flat: e = j*64 + lane, j = 0..2 (70 % of the lanes
    // carry data; four 34-lane row segments would be 53 % and a third more registers in flight).  Row and column of a lane's
    // three elements are fixed for the kernel; the LDS destination is lane-linear (i_t[c][e]).
    constexpr int I_E = (WT_R + 2) * IN_PW;                 // 136
    constexpr int I_J = (I_E + 63) / 64;                    // 3 wave-instructions per channel
    constexpr int IP_IT = PREF ? (WG_CI / NW) * I_J : 1;    // 48 registers in flight for the 4-wave shape
    int er[I_J], ec[I_J];
#pragma unroll
    for (int j = 0; j < I_J; ++j) { const int e = j * 64 + lane; er[j] = e / IN_PW; ec[j] = e - er[j] * IN_PW; }
    float gv[G_IT], ivp[IP_IT];
    auto lane_offsets = [&](int X0, int Y0, uint32_t (&off)[I_J], bool (&ok)[I_J]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < I_J; ++j) {
            const int yi = Y0 - 1 + er[j], xi = X0 - 1 + ec[j];
            ok[j] = (j * 64 + lane < I_E) && yi >= 0 && yi < H && xi >= 0 && xi < W;
            off[j] = ok[j] ? (uint32_t)(yi * W + xi) * 4u : 0u;
        }
    };
    auto issue = [&](int tile) __attribute__((always_inline)) {         // loads only
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
        const uint32_t poff = (yy < H && xx < W) ? (uint32_t)(yy * W + xx) * 4u : 0u;
#pragma unroll
        for (int k = 0; k < G_IT; ++k) {
            const int co = cb * WG_CO + wave + NW * k;
            const float* base = g + ((int64_t)n * Cout + (co < Cout ? co : 0)) * plane;
            gv[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + poff);
        }
        if constexpr (PREF) {
            uint32_t off[I_J]; bool ok[I_J];
            lane_offsets(X0, Y0, off, ok);
#pragma unroll
            for (int k = 0; k < WG_CI / NW; ++k) {
                const int ci = ib * WG_CI + wave + NW * k;                // uniform
                const float* base = in + ((int64_t)n * Cin + (ci < Cin ? ci : 0)) * plane;
#pragma unroll
                for (int j = 0; j < I_J; ++j)
                    ivp[k * I_J + j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off[j]);
            }
        }
    };
    auto commit = [&](int tile) __attribute__((always_inline)) {        // masks + LDS stores of what issue(tile) loaded
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
        const bool pix_ok = yy < H && xx < W;
#pragma unroll
        for (int k = 0; k < G_IT; ++k) {
            const int c = wave + NW * k;
            g_t[c * G_P + lane] = (pix_ok && cb * WG_CO + c < Cout) ? gv[k] : 0.f;
        }
        if constexpr (PREF) {
            uint32_t off[I_J]; bool ok[I_J];
            lane_offsets(X0, Y0, off, ok);
#pragma unroll
            for (int k = 0; k < WG_CI / NW; ++k) {
                const int c = wave + NW * k;
                const bool ch_ok = ib * WG_CI + c < Cin;
#pragma unroll
                for (int j = 0; j < I_J; ++j)
                    if (j * 64 + lane < I_E) i_t[c * I_P + j * 64 + lane] = (ch_ok && ok[j]) ? ivp[k * I_J + j] : 0.f;
            }
        }
    };
    if constexpr (PREF) {
        if (ks < ntiles) issue(ks);
    }
    for (int tile = ks; tile < ntiles; tile += ksplit) {
        if constexpr (PREF) {
            commit(tile);
        } else {
        const int tx = tile % tiles_x;
        const int r0 = tile / tiles_x;
        const int ty = r0 % tiles_y, n = r0 / tiles_y;
        const int X0 = tx * TW, Y0 = ty * WT_R;
        constexpr int I_B = (NW == 4) ? (I_IT < 64 ? I_IT : 64) : 32;   // segments in flight per pass (one pass for the 4-wave shape)
        static_assert(I_IT % I_B == 0, "whole passes");
        {
            float gv[G_IT];
            const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
            const bool pix_ok = yy < H && xx < W;
            const uint32_t poff = pix_ok ? (uint32_t)(yy * W + xx) * 4u : 0u;
#pragma unroll
            for (int k = 0; k < G_IT; ++k) {
                const int c = wave + NW * k;                              // uniform
                const int co = cb * WG_CO + c;
                const float* base = g + ((int64_t)n * Cout + (co < Cout ? co : 0)) * plane;
                const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + poff);
                gv[k] = (pix_ok && co < Cout) ? v : 0.f;
            }
            const int xi = X0 - 1 + lane;
            const bool col_ok = lane < IN_PW && xi >= 0 && xi < W;
            const uint32_t xoff = col_ok ? (uint32_t)xi * 4u : 0u;
#pragma unroll 1
            for (int k0 = 0; k0 < I_IT; k0 += I_B) {
                float iv[I_B];
#pragma unroll
                for (int k = 0; k < I_B; ++k) {
                    const int pr = wave + NW * (k0 + k);                  // uniform: (channel, row) pair
                    const int c = pr / (WT_R + 2), r = pr % (WT_R + 2);
                    const int ci = ib * WG_CI + c, yi = Y0 - 1 + r;
                    const bool row_ok = ci < Cin && yi >= 0 && yi < H;
                    const float* base = in + ((int64_t)n * Cin + (ci < Cin ? ci : 0)) * plane + (int64_t)(row_ok ? yi : 0) * W;
                    const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + xoff);
                    iv[k] = (row_ok && col_ok) ? v : 0.f;
                }
                if (k0 == 0) {
#pragma unroll
                    for (int k = 0; k < G_IT; ++k) g_t[(wave + NW * k) * G_P + lane] = gv[k];
                }
                if (lane < IN_PW) {
#pragma unroll
                    for (int k = 0; k < I_B; ++k) {
                        const int pr = wave + NW * (k0 + k);
                        i_t[(pr / (WT_R + 2)) * I_P + (pr % (WT_R + 2)) * IN_PW + lane] = iv[k];
                    }
                }
            }
        }
        }
        __syncthreads();
        if constexpr (PREF) {
            if (tile + ksplit < ntiles) issue(tile + ksplit);           // in flight during this tile's MFMAs
        }
        if (do_bias) {      // rows of g_t are G_P = 65 dwords apart: consecutive channels hit consecutive banks
            const float* gp = g_t + bch * G_P + bpart * BPIX;
#pragma unroll
            for (int q = 0; q < BPIX; ++q) bsum += gp[q];
        }
#pragma unroll 4
        for (int s = 0; s < WT_R * TW / 2; ++s) {
            const int p = 2 * s, r = p / TW, c = p % TW;
            const float a = ap[p];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float b = bp[(r + t / 3) * IN_PW + c + t % 3];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- partial sums -> slab (wgrad_slab_index; row = co, lane column = ci: 128-B contiguous stores)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = cb * WG_CO + wi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const int ci = ib * WG_CI + wj * 32 + j;
            slab[wgrad_slab_index(ks, t, co, ci, CoutP, CinP)] = acc[t][q];
        }
    }
    if (do_bias) bias_slab[((int64_t)ks * BPARTS + bpart) * CoutP + cb * WG_CO + bch] = bsum;
}

// ---- second-generation weight gradient: K split over the waves of a workgroup -------------------------------------------------
// The kernel above gives every wave one 32x32 (co x ci) quadrant and splits the pixels over WORKGROUPS only: a layer with few
// (co, ci) blocks needs hundreds of workgroups -- i.e. partial slabs -- to fill the chip (9 x 64 x 64 floats each: the reduce
// launch read 75-150 MB per layer), a channel count <= 32 wastes whole quadrants of MFMAs (6 -> 32: 95 %), and at the per-GPU
// batch of a data-parallel step (2 samples) most layers ran on 64-128 workgroups of a 256-CU chip.  Here a workgroup has 8 waves
// = WCO x WCI quadrants x WK pixel groups: the WK waves of a quadrant walk different rows of the same (WK * RPW) x 32 pixel tile
// and their accumulators are added inside the workgroup (through LDS, in wave order: fixed) before ONE slab leaves it.
//   (2,2,2,2)  64 x 64 blocks, 4-row tiles:  twice the waves per slab
//   (1,2,4,1) / (2,1,4,1)  one side <= 32 channels: no padded quadrants, four waves per slab
//   (1,1,8,1)  both sides <= 32: eight waves per slab
// Staging as in the 4-wave kernel above: one wave-instruction = 64 consecutive floats of one channel's flat g tile / input tile,
// loads of tile t+1 issued behind the barrier that opens tile t's MFMA phase, masked and stored to LDS at the top of the next trip.
template <int WCO, int WCI, int WK, int RPW, int WT = 32>
__global__ __launch_bounds__(64 * WCO * WCI * WK, 1) void conv3x3_wgrad_mfma_v2(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ slab,
    int N, int Cin, int H, int W, int Cout, int CinP, int CoutP, int ksplit, int tiles_x, int tiles_y,
    float* __restrict__ bias_slab)
{
    constexpr int NQ = WCO * WCI, NW = NQ * WK, THREADS = 64 * NW;
    constexpr int WG_CO = 32 * WCO, WG_CI = 32 * WCI;
    // WT = 16 (maps up to 16 pixels wide): the 32 pixels of an MFMA row are two image rows of 16, as in conv3x3_mfma<.., 16>
    constexpr int RM = 32 / WT;                        // image rows per MFMA row
    constexpr int TR = WK * RPW * RM;                  // pixel-tile rows (image rows)
    constexpr int TW = WT, IN_PW = WT + 2;             // shadow the file-level 32 / 34
    constexpr int G_E = TR * TW, G_P = G_E + 1;        // g tile of one channel (flat rows x WT), odd pitch
    constexpr int I_E = (TR + 2) * IN_PW, I_P = I_E | 1;   // input tile of one channel ((TR+2) x (WT+2)), odd pitch
    constexpr int G_J = (G_E + 63) / 64, I_J = (I_E + 63) / 64;      // wave-instructions per channel
    constexpr int G_IT = WG_CO * G_J / NW, I_IT = WG_CI * I_J / NW;  // per wave and tile
    static_assert((WG_CO * G_J) % NW == 0 && (WG_CI * I_J) % NW == 0 && THREADS % WG_CO == 0, "staging shares");
    constexpr int STAGE_FLOATS = WG_CO * G_P + WG_CI * I_P;
    constexpr int RED_FLOATS = NW * 1024;              // one 32x32 accumulator tile per wave
    constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds2[];
    float* g_t = lds2;
    float* i_t = lds2 + WG_CO * G_P;
    (void)LDS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int wk = wave / NQ, wq = wave % NQ, wi = wq / WCI, wj = wq % WCI;
    const int nib = CinP / WG_CI;
    const int blk = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
    const int cb = blk / nib, ib = blk % nib;
    const int64_t plane = (int64_t)H * W;
    const int ntiles = N * tiles_y * tiles_x;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const bool do_bias = (bias_slab != nullptr) && (ib == 0);
    constexpr int BPARTS = THREADS / WG_CO, BPIX = G_E / BPARTS;
    static_assert(G_E % BPARTS == 0, "bias parts");
    const int bch = tid % WG_CO, bpart = tid / WG_CO;
    float bsum = 0.f;

    auto geometry = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        const int tx = tile % tiles_x;
        const int r0 = tile / tiles_x;
        n = r0 / tiles_y; X0 = tx * TW; Y0 = (r0 % tiles_y) * TR;
    };
    // flat positions of this lane's elements (fixed for the kernel): g element e = q*64 + lane -> (row e/32, col e%32);
    // input element e = q*64 + lane -> (row e/34, col e%34)
    int ger[G_J], gec[G_J], ier[I_J], iec[I_J];
#pragma unroll
    for (int q = 0; q < G_J; ++q) { const int e = q * 64 + lane; ger[q] = e / TW; gec[q] = e - ger[q] * TW; }
#pragma unroll
    for (int q = 0; q < I_J; ++q) { const int e = q * 64 + lane; ier[q] = e / IN_PW; iec[q] = e - ier[q] * IN_PW; }

    float gv[G_IT], iv[I_IT];
    auto issue = [&](int tile) __attribute__((always_inline)) {          // loads only (clamped addresses, unconditional)
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        uint32_t goff[G_J], ioff[I_J];
#pragma unroll
        for (int q = 0; q < G_J; ++q) {
            const int yy = Y0 + ger[q], xx = X0 + gec[q];
            goff[q] = (q * 64 + lane < G_E && yy < H && xx < W) ? (uint32_t)(yy * W + xx) * 4u : 0u;
        }
#pragma unroll
        for (int q = 0; q < I_J; ++q) {
            const int yi = Y0 - 1 + ier[q], xi = X0 - 1 + iec[q];
            ioff[q] = (q * 64 + lane < I_E && yi >= 0 && yi < H && xi >= 0 && xi < W) ? (uint32_t)(yi * W + xi) * 4u : 0u;
        }
#pragma unroll
        for (int k = 0; k < G_IT; ++k) {
            const int idx = wave + NW * k;                                // uniform: (channel, instruction) pair
            const int c = idx / G_J, q = idx % G_J;
            const int co = cb * WG_CO + c;
            const float* base = g + ((int64_t)n * Cout + (co < Cout ? co : 0)) * plane;
            gv[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + goff[q % G_J]);
        }
#pragma unroll
        for (int k = 0; k < I_IT; ++k) {
            const int idx = wave + NW * k;
            const int c = idx / I_J, q = idx % I_J;
            const int ci = ib * WG_CI + c;
            const float* base = in + ((int64_t)n * Cin + (ci < Cin ? ci : 0)) * plane;
            iv[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + ioff[q % I_J]);
        }
    };
    auto commit = [&](int tile) __attribute__((always_inline)) {         // masks (recomputed from the tile index) + LDS stores
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        bool gok[G_J], iok[I_J];
#pragma unroll
        for (int q = 0; q < G_J; ++q) gok[q] = (Y0 + ger[q]) < H && (X0 + gec[q]) < W;
#pragma unroll
        for (int q = 0; q < I_J; ++q) {
            const int yi = Y0 - 1 + ier[q], xi = X0 - 1 + iec[q];
            iok[q] = yi >= 0 && yi < H && xi >= 0 && xi < W;
        }
#pragma unroll
        for (int k = 0; k < G_IT; ++k) {
            const int idx = wave + NW * k;
            const int c = idx / G_J, q = idx % G_J;
            const int e = q * 64 + lane;
            if (e < G_E) g_t[c * G_P + e] = (gok[q % G_J] && cb * WG_CO + c < Cout) ? gv[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < I_IT; ++k) {
            const int idx = wave + NW * k;
            const int c = idx / I_J, q = idx % I_J;
            const int e = q * 64 + lane;
            if (e < I_E) i_t[c * I_P + e] = (iok[q % I_J] && ib * WG_CI + c < Cin) ? iv[k] : 0.f;
        }
    };

    // this wave's operands: A = g of its co quadrant at its own rows, B = input of its ci quadrant
    const float* ap = g_t + (wi * 32 + j) * G_P + (wk * RPW) * 32 + h;                 // flat: MFMA row (wk*RPW + r), pixel p
    const float* bp = i_t + (wj * 32 + j) * I_P + (wk * RPW * RM) * IN_PW + h;

    if (ks < ntiles) issue(ks);
    for (int tile = ks; tile < ntiles; tile += ksplit) {
        commit(tile);
        __syncthreads();
        if (tile + ksplit < ntiles) issue(tile + ksplit);                 // in flight during this tile's MFMAs
        if (do_bias) {
            const float* gp = g_t + bch * G_P + bpart * BPIX;
#pragma unroll
            for (int q = 0; q < BPIX; ++q) bsum += gp[q];
        }
#pragma unroll 4
        for (int s = 0; s < RPW * 16; ++s) {
            const int p = 2 * s, r = p / TW, c = p % TW;          // image row (within the wave's rows) and column of pixel p
            const float a = ap[p];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float b = bp[(r + t / 3) * IN_PW + c + t % 3];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- add the WK partial accumulators of every quadrant (wave order: fixed) and store ONE slab: per tap, every wave parks its
    // tile in LDS (element (row = co, col = ci) at [wave][row * 32 + col]), then the workgroup's threads add and store 128-B rows
    float* red = lds2;
    constexpr int PER_T = NQ * 1024 / THREADS;         // output elements per thread and tap
#pragma unroll                                          // (fully: a run-time tap index would put the accumulators in scratch)
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave * 1024 + ((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + j] = acc[t][q];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            const int o = tid + THREADS * e;                     // (quadrant, row, col)
            const int qd = o >> 10, rc = o & 1023;
            float v = red[qd * 1024 + rc];                       // wk = 0
#pragma unroll
            for (int k = 1; k < WK; ++k) v += red[(k * NQ + qd) * 1024 + rc];
            const int co = cb * WG_CO + (qd / WCI) * 32 + (rc >> 5);
            const int ci = ib * WG_CI + (qd % WCI) * 32 + (rc & 31);
            slab[wgrad_slab_index(ks, t, co, ci, CoutP, CinP)] = v;
        }
        __syncthreads();
    }
    if (do_bias) bias_slab[((int64_t)ks * BPARTS + bpart) * CoutP + cb * WG_CO + bch] = bsum;
}

// Fixed-order sum of the K slices.  One workgroup = one output channel co x 64 input channels x all nine taps:
//   reads   slab[k][co][ci0 / 64][t][0..63] for every slice k: one run of 2,304 bytes per slice (wgrad_slab_index), the slices spread
//           over `groups` = blockDim.x / 64 slice groups (thread (e, kg) adds slices kg, kg + groups, ... of element e for each tap);
//   writes  gw[co][ci0 .. ci0+63][0..8]: 576 consecutive floats, one coalesced run (a first version gave every (tap, co) row its own
//           workgroup: each thread then stored ONE float 36 bytes from its neighbour's, nine such scattered passes per output run --
//           50-74 us per layer on the IFNet's 9 MB weight tensors, 10 % of its training step).
// The partial sums are combined in group order: same bits on every run.  accumulate != 0: gw / gb are the parameters' .grad
// buffers and the sums are ADDED to what they hold (one read-modify-write per element, stream order: deterministic).
// The last CoutP / 64 workgroups add up the bias rows the same way.
constexpr int RED_KG = 16;           // most slice groups per workgroup
// the body of one reduce workgroup (shared by the per-layer kernel and the grouped launch): TPW taps per workgroup -- 9 (one coalesced
// 576-float run), or 3 when there are few (co, ci) blocks; blk0 = the workgroup's index among the job's wblocks + bias blocks;
// ngroups = slice groups (threads beyond 64 * ngroups only keep the barriers company)
template <int TPW>
__device__ __forceinline__ void conv3x3_wgrad_reduce_body(const float* __restrict__ slab, float* __restrict__ gw, int Cin, int Cout, int CinP,
                                                          int CoutP, int ksplit, const float* __restrict__ bias_slab, float* __restrict__ gb,
                                                          int bias_rows, int wblocks, int accumulate, int blk0, int ngroups, float* part_raw)
{
    // TPW = 3: the thin layers have hundreds of slabs but only 64-128 (co, 64 ci) blocks -- 65 workgroups read 78 MB in 51 us; three
    // workgroups per block (taps 0-2, 3-5, 6-8: runs of 3 floats every 9) triple the loads in flight.
    constexpr int TSPLIT = 9 / TPW;
    float (*part)[TPW][64] = reinterpret_cast<float (*)[TPW][64]>(part_raw);
    const int e = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const bool on = kg < ngroups;
    const int nthreads = 64 * ngroups;
    if (blk0 >= wblocks) {
        const int co = (blk0 - wblocks) * 64 + e;
        float s = 0.f;
        if (on && co < CoutP) {
#pragma unroll 4
            for (int r = kg; r < bias_rows; r += ngroups) s += bias_slab[(int64_t)r * CoutP + co];
        }
        if (on) part[kg][0][e] = s;
        __syncthreads();
        if (kg == 0 && co < Cout) {
            float v = part[0][0][e];
            for (int k = 1; k < ngroups; ++k) v += part[k][0][e];
            gb[co] = accumulate ? gb[co] + v : v;
        }
        return;
    }
    const int cblocks = (CinP + 63) / 64;
    constexpr int64_t tap_stride = 64;                            // between taps inside a (co, 64 ci) block: wgrad_slab_index
    const int64_t slice = wgrad_slab_floats(CoutP, CinP);
    for (int64_t blk = blk0; blk < (int64_t)CoutP * cblocks * TSPLIT; blk += wblocks) {
        const int t0 = (int)(blk % TSPLIT) * TPW;
        const int64_t cbk = blk / TSPLIT;
        const int co = (int)(cbk / cblocks), ci0 = (int)(cbk % cblocks) * 64;
        const int ci = ci0 + e;
        float s[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) s[t] = 0.f;
        if (on && ci < CinP) {
            const float* p = slab + wgrad_slab_index(0, t0, co, ci, CoutP, CinP);
#pragma unroll 2
            for (int k = kg; k < ksplit; k += ngroups) {
                const float* q = p + (int64_t)k * slice;
#pragma unroll
                for (int t = 0; t < TPW; ++t) s[t] += q[t * tap_stride];
            }
        }
        if (on) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) part[kg][t][e] = s[t];
        }
        __syncthreads();
        if (on && co < Cout) {
            for (int o = threadIdx.x; o < 64 * TPW; o += nthreads) {          // (local ci, tap) in memory order
                const int cl = o / TPW, t = o - cl * TPW;
                if (ci0 + cl < Cin) {
                    float v = part[0][t][cl];
                    for (int k = 1; k < ngroups; ++k) v += part[k][t][cl];
                    float* dst = gw + ((int64_t)co * Cin + ci0 + cl) * 9 + t0 + t;
                    *dst = accumulate ? *dst + v : v;
                }
            }
        }
        __syncthreads();
    }
}

template <int TPW>                   // taps per workgroup
__global__ __launch_bounds__(64 * RED_KG) void conv3x3_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ gw,
                                                                    int Cin, int Cout, int CinP, int CoutP, int ksplit,
                                                                    const float* __restrict__ bias_slab, float* __restrict__ gb,
                                                                    int bias_rows, int wblocks, int accumulate)
{
    __shared__ float part[RED_KG * TPW * 64];
    conv3x3_wgrad_reduce_body<TPW>(slab, gw, Cin, Cout, CinP, CoutP, ksplit, bias_slab, gb, bias_rows, wblocks, accumulate, (int)blockIdx.x,
                                   (int)(blockDim.x >> 6), part);
}

// every deferred reduce job of a backward pass in one launch: a workgroup finds its job by its block range and runs that job's body
// with that job's number of slice groups (the per-layer launch's workgroup shape: the same sums in the same order)
__global__ __launch_bounds__(64 * RED_KG) void wgrad_reduce_group(const WgradReduceGroup g)
{
    __shared__ float part[RED_KG * 9 * 64];
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.job[j + 1].block0) ++j;
    const WgradReduceJob& q = g.job[j];
    const int blk0 = (int)blockIdx.x - q.block0;
    if (q.kind == 0)
        conv3x3_wgrad_reduce_body<9>(q.slab, q.gw, q.Cin, q.Cout, q.CinP, q.CoutP, q.ksplit, q.bias_slab, q.gb, q.bias_rows, q.wblocks, q.accumulate,
                                     blk0, q.ngroups, part);
    else if (q.kind == 1)
        conv3x3_wgrad_reduce_body<3>(q.slab, q.gw, q.Cin, q.Cout, q.CinP, q.CoutP, q.ksplit, q.bias_slab, q.gb, q.bias_rows, q.wblocks, q.accumulate,
                                     blk0, q.ngroups, part);
    else
        convT_wgrad_reduce_body(q.slab, q.gw, q.Cin, q.Cout, q.CinP, q.CoutP, q.ksplit, q.bias_slab, q.gb, q.bias_rows, q.wblocks, q.accumulate,
                                blk0, q.ngroups, reinterpret_cast<float (*)[64]>(part));
}

// ---- host launchers ------------------------------------------------------------------------
static inline int grid_1d(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

int conv3x3_co_block(int Cout)
{
    static const int forced = [] { const char* e = getenv("SSTEM_CONV_CO"); return e ? atoi(e) : 0; }();
    if (forced == 32 || forced == 64) return forced;   // developer knob for A/B runs
    (void)Cout;
    return 32;   // measured on MI355X: 32-channel blocks (40 KB LDS, 4 workgroups per CU) beat 64 on every layer shape
}

int64_t conv3x3_workspace_floats(int Cin, int Cout)
{
    const int CO = conv3x3_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    return (int64_t)ncb * nchunks * KK * CO;
}

// K slices for small grids.  A workgroup owns an 8x32-pixel x 32/64-channel tile and walks all of K; when the tile count
// is below two workgroups per CU (deep layers at small batch: 64 workgroups at N = 2, 512 channels, 16x16 -- measured
// 4.5x less efficient per sample than the same layer at N = 16) K is cut into 2, 4 or 8 slices.  Pure function of the
// problem size; slices divide the chunk count evenly and keep at least two chunks each (double-buffered pipeline).
// Tile geometry of conv3x3_mfma, a pure function of the problem size:
//   tile width 32, or 16 on maps up to 16 pixels wide (two image rows per MFMA row; SSTEM_CONV_NARROW=0: off);
//   two MFMA rows per wave (8-row tiles), or one (4-row tiles) when the 8-row tiling gives fewer than 512 workgroups
//   (SSTEM_CONV_RPW1=0: off);  then K slices: when the tile count is still below two workgroups per CU (deep layers at small batch)
//   K is cut into 2, 4 or 8 slices of whole input-channel chunks, at least two chunks each (SSTEM_CONV_KSPLIT=0: off).
int conv3x3_co_block(int Cout);
struct ConvGeom { int tw, th, rpw, ksplit, tiles_x, tiles_y; };
static ConvGeom conv_geom(int N, int Cin, int H, int W, int Cout)
{
    static const bool narrow = [] { const char* e = getenv("SSTEM_CONV_NARROW"); return !(e && atoi(e) == 0); }();
    static const bool rpw1 = [] { const char* e = getenv("SSTEM_CONV_RPW1"); return !(e && atoi(e) == 0); }();
    static const bool ks_off = [] { const char* e = getenv("SSTEM_CONV_KSPLIT"); return e && atoi(e) == 0; }();
    static const int wg_target = [] { const char* e = getenv("SSTEM_CONV_WG_TARGET"); return e && atoi(e) > 0 ? atoi(e) : 512; }();   // developer knob
    const int CO = conv3x3_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    ConvGeom g;
    g.tw = (narrow && W <= 16 && CO == 32) ? 16 : 32;
    const int rm = 32 / g.tw;
    g.rpw = 2;
    g.tiles_x = (W + g.tw - 1) / g.tw;
    auto wgs_for = [&](int rpw) { const int th = 4 * rpw * rm; return (int64_t)g.tiles_x * ((H + th - 1) / th) * N * ncb; };
    if (rpw1 && CO == 32 && wgs_for(2) < wg_target) g.rpw = 1;
    g.th = 4 * g.rpw * rm;
    g.tiles_y = (H + g.th - 1) / g.th;
    const int64_t wgs = wgs_for(g.rpw);
    int ks = 1;
    if (!ks_off)
        while (wgs * ks < wg_target && ks < 8 && nchunks % (ks * 2) == 0 && nchunks / (ks * 2) >= 2) ks *= 2;
    g.ksplit = ks;
    return g;
}

int conv3x3_ksplit(int N, int Cin, int H, int W, int Cout) { return conv_geom(N, Cin, H, W, Cout).ksplit; }

int64_t conv3x3_forward_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const int ks = conv3x3_ksplit(N, Cin, H, W, Cout);
    return conv3x3_workspace_floats(Cin, Cout) + (ks > 1 ? (int64_t)ks * N * Cout * H * W : 0);
}

// number of (count, mean, M2) partials per channel a launch with ex.bn_part writes (the caller sizes bn_part = Cout * this * 3)
int64_t conv3x3_bn_partials(int N, int Cin, int H, int W, int Cout)
{
    // A launch split over K leaves the statistics to the BatchNorm's own pass: measured at batch 2 (profiles/r02), the slice-sum
    // kernel that also produced them (conv3x3_splitk_epilogue_bn, one workgroup per (sample, channel, piece) with two block
    // reductions) took 25.8 us per layer against 6.3 + 6.2 us for the plain slice sum + the BatchNorm partial pass on these
    // small tensors.  SSTEM_SPLITK_BN=1 brings it back (A/B runs).
    static const bool splitk_bn = [] { const char* e = getenv("SSTEM_SPLITK_BN"); return e && atoi(e) != 0; }();
    const ConvGeom g = conv_geom(N, Cin, H, W, Cout);
    if (g.ksplit > 1)
        return splitk_bn ? (int64_t)N * (((int64_t)H * W + SPLITK_BN_CHUNK - 1) / SPLITK_BN_CHUNK) : 0;
    return (int64_t)N * g.tiles_x * g.tiles_y;
}

hipError_t launch_conv3x3_mfma(const float* in, const float* w, const float* bias, const float* scale,
                               const float* shift, float* out, float* workspace, int64_t workspace_floats, int N,
                               int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                               hipStream_t s, const ConvExtra& ex_in)
{
    // w_transposed_flipped bit 1 (SSTEM_CONV_WEIGHT_PREPACKED): the head of the workspace already holds this call's packed
    // weights (an earlier call with the same weights, orientation and sizes wrote them): no pack launch
    const bool prepacked = (w_transposed_flipped & 2) != 0;
    w_transposed_flipped &= 1;
    const int CO = conv3x3_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    const int64_t wtotal = (int64_t)ncb * nchunks * KK * CO;
    hipError_t e = hipSuccess;
    if (!prepacked) {
        hipLaunchKernelGGL(pack_weights_3x3, dim3(grid_1d(wtotal, 256)), dim3(256), 0, s, w, workspace, Cin,
                           Cout, CO, nchunks, ncb, w_transposed_flipped);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // split K only when the caller's workspace has room for the slices (sstem_conv3x3_forward_workspace_floats)
    const ConvGeom gm = conv_geom(N, Cin, H, W, Cout);
    int ksplit = gm.ksplit;
    const int64_t out_elems = (int64_t)N * Cout * H * W;
    if (ksplit > 1 && workspace_floats < wtotal + (int64_t)ksplit * out_elems) {
        if (ex_in.bn_part) return hipErrorInvalidValue;      // the partial layout follows conv3x3_ksplit: the full workspace is required
        ksplit = 1;
    }
    float* slab = workspace + wtotal;
    if ((int64_t)N * ncb * ksplit > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)gm.tiles_x, (unsigned)gm.tiles_y, (unsigned)(N * ncb * ksplit));
    const size_t lds_bytes = 2 * (size_t)(IN_TILE + KK * CO) * sizeof(float);      // (the 16-wide tile needs 2592 of the 2720 input floats)
    static const int remap_knob = [] { const char* e = getenv("SSTEM_XCD_REMAP"); return e ? atoi(e) : 1; }();     // developer knob (A/B runs)
    const int remap = (remap_knob && (int64_t)grid.x * grid.y * grid.z < ((int64_t)1 << 31)) ? 1 : 0;   // 32-bit linear tile ids in the kernel
    ConvExtra ex = ex_in;
    ex.bn_tiles = (int)(N * grid.x * grid.y);
    if (ex.bn_part && (scale || shift || act != 0 || ex.residual)) return hipErrorInvalidValue;   // statistics of the raw conv + bias only
    if (CO == 64) {
        auto k = conv3x3_mfma<2>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, s, in, workspace, bias, scale, shift, out, N, Cin, H, W,
                           Cout, nchunks, ncb, act, slope, ksplit, slab, remap, ex);
    } else if (gm.tw == 16 || gm.rpw == 1) {
#define SSTEM_CONV_VARIANT(WT_, RPW_)                                                                                               \
    {                                                                                                                               \
        auto k = conv3x3_mfma<1, WT_, RPW_>;                                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);     \
        if (e != hipSuccess) return e;                                                                                              \
        hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, s, in, workspace, bias, scale, shift, out, N, Cin, H, W, Cout, nchunks,   \
                           ncb, act, slope, ksplit, slab, remap, ex);                                                               \
    }
        if (gm.tw == 16 && gm.rpw == 1) SSTEM_CONV_VARIANT(16, 1)
        else if (gm.tw == 16) SSTEM_CONV_VARIANT(16, 2)
        else SSTEM_CONV_VARIANT(32, 1)
#undef SSTEM_CONV_VARIANT
    } else {
        auto k = conv3x3_mfma<1>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, s, in, workspace, bias, scale, shift, out, N, Cin, H, W,
                           Cout, nchunks, ncb, act, slope, ksplit, slab, remap, ex);
    }
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    if (ex.bn_part) {
        const int pieces = (int)(((int64_t)H * W + SPLITK_BN_CHUNK - 1) / SPLITK_BN_CHUNK);
        if (Cout > 65535) return hipErrorInvalidValue;
        hipLaunchKernelGGL(conv3x3_splitk_epilogue_bn, dim3((unsigned)(N * pieces), (unsigned)Cout), dim3(256), 0, s, slab, bias, out,
                           out_elems, (int64_t)H * W, Cout, ksplit, pieces, ex.bn_part);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(conv3x3_splitk_epilogue, dim3(grid_1d(out_elems, 256)), dim3(256), 0, s, slab, bias, scale, shift,
                       out, out_elems, (int64_t)H * W, Cout, ksplit, act, slope, ex.residual, ex.res_scale);
    return hipGetLastError();
}

hipError_t launch_pack_weights_3x3_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s)
{
    const int CO_f = conv3x3_co_block(Cout), CO_t = conv3x3_co_block(Cin);
    const int ncb_f = (Cout + CO_f - 1) / CO_f, nchunks_f = (Cin + KC - 1) / KC;
    const int ncb_t = (Cin + CO_t - 1) / CO_t, nchunks_t = (Cout + KC - 1) / KC;
    const int64_t n_f = wp_f ? (int64_t)ncb_f * nchunks_f * KK * CO_f : 0, n_t = wp_t ? (int64_t)ncb_t * nchunks_t * KK * CO_t : 0;
    hipLaunchKernelGGL(pack_weights_3x3_both, dim3(grid_1d(n_f + n_t, 256)), dim3(256), 0, s, w, wp_f, wp_t, Cin, Cout, CO_f,
                       nchunks_f, ncb_f, n_f, CO_t, nchunks_t, ncb_t, n_t);
    return hipGetLastError();
}

// layout numbers of one entry of the group-pack table (out[3..12]); returns the 256-thread blocks the entry needs
int64_t pack_group_entry(int Cin, int Cout, int64_t* out)
{
    const int CO_f = conv3x3_co_block(Cout), CO_t = conv3x3_co_block(Cin);
    const int ncb_f = (Cout + CO_f - 1) / CO_f, nchunks_f = (Cin + KC - 1) / KC;
    const int ncb_t = (Cin + CO_t - 1) / CO_t, nchunks_t = (Cout + KC - 1) / KC;
    out[3] = Cin; out[4] = Cout;
    out[5] = CO_f; out[6] = nchunks_f; out[7] = ncb_f; out[8] = (int64_t)ncb_f * nchunks_f * KK * CO_f;
    out[9] = CO_t; out[10] = nchunks_t; out[11] = ncb_t; out[12] = (int64_t)ncb_t * nchunks_t * KK * CO_t;
    return (out[8] + out[12] + 255) / 256;
}

hipError_t launch_pack_weights_3x3_group(const int64_t* table, int n_entries, int64_t total_blocks, hipStream_t s)
{
    if (n_entries <= 0 || total_blocks <= 0) return hipSuccess;
    if (total_blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_weights_3x3_group, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_entries);
    return hipGetLastError();
}

// 1 x 1 convolutions with a handful of output channels (the OutConv of the SP U-Nets, networks.py:238: 64 -> 1 at full resolution;
// the SFF nets' last layers are 3 x 3): a stream over the input planes, four pixels per thread by 16-byte loads, the channel sum in the
// direct kernel's order (ci ascending, one fma per product) -- the same bits, 2.6 -> 4.5 TB/s.
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_stream(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ out, int N, int Cin, int64_t plane, int act, float slope)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t quads = plane >> 2;
    const int64_t total = (int64_t)N * quads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / quads, q = i - n * quads;
        const f4* ip = reinterpret_cast<const f4*>(in + (int64_t)n * Cin * plane) + q;
        f4 acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ci = 0; ci < Cin; ++ci) {
            const f4 v = ip[(int64_t)ci * quads];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float wv = w[co * Cin + ci];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(v[e], wv, acc[co][e]);
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            f4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[co][e] + (bias ? bias[co] : 0.f);
                v = v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
                r[e] = apply_act(v, act, slope);
            }
            reinterpret_cast<f4*>(out + ((int64_t)n * COUT + co) * plane)[q] = r;
        }
    }
}

// 3 x 3 convolutions with a handful of OUTPUT channels (the last layers of the SFF nets: 32 -> 2 flow, 32 -> 1 restored section at
// full resolution; the IFNets' first block 6 -> 6): on the MFMA kernels the 32-channel output block is 3-6 % occupied and the launch
// takes 0.57 ms at 8 x 1024^2 where the input stream takes 0.2.  Here: a stream over the input planes, every thread 4 columns x 2 rows of
// all COUT channels (4 tile rows x (16 + 4 + 4) bytes loaded per input channel, the halos from the neighbours' cache lines), exact fp32
// fma in (ci, ky, kx) order, weights by scalar loads, one 16-byte store per row and channel; the launch's largest stored magnitude
// goes to `out_amax` (an amax word, nullable) like the split kernels' -- the next layer of an fp16 chain needs no measuring pass.
template <int COUT, int TXL = 64, int UNR = 2>          // TXL = 64: one wave per row group (the halo exchange below)
__global__ __launch_bounds__(256) void conv3x3_stream_small(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ out, int Cin, int H, int W, int act, float slope,
                                                            float* __restrict__ out_amax)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tx = threadIdx.x % TXL, ty = threadIdx.x / TXL;                // TXL x (256 / TXL) threads: a tile of 4 TXL columns x 512 / TXL rows
    const int x0 = (blockIdx.x * TXL + tx) * 4, y0 = (blockIdx.y * (256 / TXL) + ty) * 2;
    const int n = blockIdx.z;
    const int64_t plane = (int64_t)H * W;
    const bool live = x0 < W && y0 < H;                                      // W % 4 == 0: a live thread's four columns are inside
    float vmax = 0.f;
    if (live) {
        // one buffer resource over the image's Cin planes (below 2^31 bytes: the launcher), the plane in the SGPR offset, the lane's row
        // starts in four VGPRs: rows outside the image and the halo columns of the first / last thread of a row carry offset 2^31 and
        // read zeros through the range check -- no predicates, no 64-bit address arithmetic in the loop
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)n * Cin * plane), 0,
                                                                             (int)((uint32_t)Cin * (uint32_t)plane * 4u), 0x00020000);
        const uint32_t OOB = 0x80000000u;
        uint32_t vm[4], ve[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 - 1 + r;
            const bool ok = y >= 0 && y < H;
            vm[r] = ok ? (uint32_t)(y * W + x0) * 4u : OOB;
            ve[r] = OOB;                                          // the tile's edge threads: column x0 - 1 (first) / x0 + 4 (last)
            if (tx == 0 && ok && x0 > 0) ve[r] = vm[r] - 4u;
            if (tx == TXL - 1 && ok && x0 + 4 < W) ve[r] = vm[r] + 16u;
        }
        const uint32_t plane4 = (uint32_t)plane * 4u;
        f4 acc[COUT][2];
#pragma unroll
        for (int co = 0; co < COUT; ++co) { acc[co][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[co][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll UNR
        for (int ci = 0; ci < Cin; ++ci) {
            float v[4][6];                                                   // tile rows y0 - 1 .. y0 + 2, columns x0 - 1 .. x0 + 4
            const int so = (int)((uint32_t)ci * plane4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f4 m = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vm[r], so, 0));
                // the halo columns come from the neighbouring lanes' vectors (one wave = 64 threads along x = one row group of the tile);
                // only the tile's first and last thread read theirs from memory, with ONE load per row: every other lane's offset is out
                // of range.  (Two dword halo loads per row and lane were a third of the kernel's time: 0.32 -> 0.21 ms without them.)
                const float edge = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, (int)ve[r], so, 0));
                const float lft = __shfl_up(m[3], 1), rgt = __shfl_down(m[0], 1);
                // the last live lane's right neighbour is not in this branch (W % 256 != 0): what a shuffle returns for an inactive
                // source lane is not ours to rely on, so the zero padding is selected explicitly (round-4 advisor finding)
                v[r][0] = tx == 0 ? edge : lft;
                v[r][5] = tx == TXL - 1 ? edge : (x0 + 4 < W ? rgt : 0.f);
                v[r][1] = m[0]; v[r][2] = m[1]; v[r][3] = m[2]; v[r][4] = m[3];
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float* wp = w + ((int64_t)co * Cin + ci) * 9;          // uniform: scalar loads
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wv = wp[ky * 3 + kx];
#pragma unroll
                        for (int o = 0; o < 2; ++o)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[co][o][e] = fmaf(v[o + ky][e + kx], wv, acc[co][o][e]);
                    }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float bs = bias ? bias[co] : 0.f, sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (y0 + o >= H) continue;
                f4 r4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = apply_act((acc[co][o][e] + bs) * sc + sh, act, slope);
                    r4[e] = t;
                    vmax = fmaxf(vmax, fabsf(t));
                }
                *reinterpret_cast<f4*>(out + ((int64_t)n * COUT + co) * plane + (int64_t)(y0 + o) * W + x0) = r4;
            }
        }
    }
    if (out_amax) {             // one atomic per workgroup into slot (workgroup & 1023) of the word (conv_split_kernels.hip, amax_word_update)
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const uint32_t slot = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 1023u;
            atomicMax(reinterpret_cast<unsigned int*>(out_amax) + slot, __builtin_bit_cast(uint32_t, m));
        }
    }
}

// ---- the IFNet's first convolution straight from the uint8 frames (round 5; SURVEY 8(f) f3 as worded) ---------------------------
// The reference reads two 8-bit grayscale PNGs, divides by 255 in float32, replicates each plane x3 and concatenates them into the
// [1,6,H,W] network input (sff_scripts_interp/inference_singleImage.py:55-66); the first layer of the IFNet is Conv2d(6 -> 6) + ReLU
// (model_interp.py:121-127).  Here that layer reads the two uint8 planes themselves: virtual channel ci is plane ci / 3, value
// float32(byte) / float32(255) (numpy's arithmetic, as sstem_gray_u8_to_f32), the products summed in conv3x3_stream_small's order
// (ci, ky, kx ascending, fp32 fma) -- the bits of that kernel on the materialised fp32 input.  The launch also leaves the two fp32
// planes behind (`planes`, nullable, [2][N][H][W]: frame-major, so each frame's planes are one contiguous [N,1,H,W] tensor): the fused
// apply at the other end of the network stages its tiles from them.  One uint8 plane
// per frame crosses into the network instead of six fp32 channels.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_first_u8(const uint8_t* __restrict__ frames, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, float* __restrict__ planes, int H, int W, int act, float slope,
                                                        float* __restrict__ out_amax)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tx = threadIdx.x % 64, ty = threadIdx.x / 64;
    const int x0 = (blockIdx.x * 64 + tx) * 4, y0 = (blockIdx.y * 4 + ty) * 2;
    const int n = blockIdx.z;
    const int64_t plane = (int64_t)H * W;
    const bool live = x0 < W && y0 < H;
    float vmax = 0.f;
    if (live) {
        // both planes of the image behind one resource of 2 plane bytes: rows outside the image and the halo columns at the image's
        // edge carry offset 2^31 and read zeros (zero padding: byte 0 is 0.0)
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frames + (int64_t)n * 2 * plane), 0,
                                                                             (int)(2u * (uint32_t)plane), 0x00020000);
        const uint32_t OOB = 0x80000000u;
        f4 acc[COUT][2];
#pragma unroll
        for (int co = 0; co < COUT; ++co) { acc[co][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[co][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[4][6];                       // tile rows y0 - 1 .. y0 + 2, columns x0 - 1 .. x0 + 4 of plane p, as float32(byte) / 255
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 - 1 + r;
                const bool ok = y >= 0 && y < H;
                const uint32_t base = ok ? (uint32_t)(y * W + x0) : OOB;
                const uint32_t m = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (int)base, (int)((uint32_t)p * (uint32_t)plane), 0);
                const uint32_t lo = (ok && x0 > 0) ? base - 1u : OOB, hi = (ok && x0 + 4 < W) ? base + 4u : OOB;
                const uint32_t l = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rin, (int)lo, (int)((uint32_t)p * (uint32_t)plane), 0);
                const uint32_t h = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rin, (int)hi, (int)((uint32_t)p * (uint32_t)plane), 0);
                v[r][0] = __fdiv_rn((float)(l & 0xffu), 255.0f);
                v[r][5] = __fdiv_rn((float)(h & 0xffu), 255.0f);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[r][1 + e] = __fdiv_rn((float)((m >> (8 * e)) & 0xffu), 255.0f);
            }
            if (planes) {
#pragma unroll
                for (int o = 0; o < 2; ++o)
                    if (y0 + o < H)
                        *reinterpret_cast<f4*>(planes + ((int64_t)p * gridDim.z + n) * plane + (int64_t)(y0 + o) * W + x0) =
                            (f4){v[1 + o][1], v[1 + o][2], v[1 + o][3], v[1 + o][4]};
            }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const int ci = 3 * p + cc;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float* wp = w + ((int64_t)co * 6 + ci) * 9;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = wp[ky * 3 + kx];
#pragma unroll
                            for (int o = 0; o < 2; ++o)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[co][o][e] = fmaf(v[o + ky][e + kx], wv, acc[co][o][e]);
                        }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float bs = bias ? bias[co] : 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (y0 + o >= H) continue;
                f4 r4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = apply_act((acc[co][o][e] + bs) * 1.f + 0.f, act, slope);      // (the fp32 kernel's epilogue with scale 1, shift 0)
                    r4[e] = t;
                    vmax = fmaxf(vmax, fabsf(t));
                }
                *reinterpret_cast<f4*>(out + ((int64_t)n * COUT + co) * plane + (int64_t)(y0 + o) * W + x0) = r4;
            }
        }
    }
    if (out_amax) {
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const uint32_t slot = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 1023u;
            atomicMax(reinterpret_cast<unsigned int*>(out_amax) + slot, __builtin_bit_cast(uint32_t, m));
        }
    }
}

bool conv3x3_first_u8_supported(int N, int H, int W, int Cout)
{
    return N > 0 && N <= 65535 && H > 0 && W > 0 && W % 4 == 0 && (H + 7) / 8 <= 65535 && (int64_t)2 * H * W < ((int64_t)1 << 31) && Cout == 6;
}

hipError_t launch_conv3x3_first_u8(const uint8_t* frames, const float* w, const float* bias, float* out, float* planes, int N, int H, int W,
                                   int Cout, int act, float slope, float* out_amax, hipStream_t s)
{
    if (!conv3x3_first_u8_supported(N, H, W, Cout) || (reinterpret_cast<uintptr_t>(frames) & 3) != 0 ||
        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(planes)) & 15) != 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)((H + 7) / 8), (unsigned)N);
    hipLaunchKernelGGL(conv3x3_first_u8<6>, grid, dim3(256), 0, s, frames, w, bias, out, planes, H, W, act, slope, out_amax);
    return hipGetLastError();
}

bool conv3x3_stream_small_supported(int N, int Cin, int H, int W, int Cout)
{
    return N > 0 && N <= 65535 && Cin > 0 && H > 0 && W > 0 && W % 4 == 0 && (H + 7) / 8 <= 65535 && (int64_t)Cin * H * W * 4 < ((int64_t)1 << 31) &&
           (Cout == 1 || Cout == 2 || Cout == 3 || Cout == 4 || Cout == 6 || Cout == 8);
}

hipError_t launch_conv3x3_stream_small(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                       float* out, int N, int Cin, int H, int W, int Cout, int act, float slope, float* out_amax, hipStream_t s)
{
    if (!conv3x3_stream_small_supported(N, Cin, H, W, Cout) ||
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) return hipErrorInvalidValue;
    // 256 columns x 8 rows per workgroup (64 x 4 threads), two input channels in flight per thread: 64-wide against 32-wide tiles and
    // an unroll of 4 against 2 measured within 5 % of each other once the loads went through the buffer resource (60-64 VGPRs for one
    // or two output channels; with predicated pointer loads 110-215 and two to four waves per SIMD: 0.39 -> 0.33 ms for 8 x 32 -> 1 at 1024^2)
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)((H + 7) / 8), (unsigned)N);
#define SSTEM_STREAM_SMALL(C) \
    hipLaunchKernelGGL((conv3x3_stream_small<C, 64, 2>), grid, dim3(256), 0, s, in, w, bias, scale, shift, out, Cin, H, W, act, slope, out_amax)
    switch (Cout) {
    case 1: SSTEM_STREAM_SMALL(1); break;
    case 2: SSTEM_STREAM_SMALL(2); break;
    case 3: SSTEM_STREAM_SMALL(3); break;
    case 4: SSTEM_STREAM_SMALL(4); break;
    case 6: SSTEM_STREAM_SMALL(6); break;
    default: SSTEM_STREAM_SMALL(8); break;
    }
#undef SSTEM_STREAM_SMALL
    return hipGetLastError();
}

hipError_t launch_conv2d_direct(const float* in, const float* w, const float* bias, const float* scale,
                                const float* shift, float* out, int N, int Cin, int H, int W, int Cout,
                                int KH, int KW, int PH, int PW, int act, float slope, hipStream_t s)
{
    const int64_t plane = (int64_t)H * W;
    if (KH == 1 && KW == 1 && PH == 0 && PW == 0 && Cout <= 2 && plane % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const int blocks = grid_1d((int64_t)N * (plane >> 2), 256);
        if (Cout == 1) hipLaunchKernelGGL(conv1x1_stream<1>, dim3(blocks), dim3(256), 0, s, in, w, bias, scale, shift, out, N, Cin, plane, act, slope);
        else hipLaunchKernelGGL(conv1x1_stream<2>, dim3(blocks), dim3(256), 0, s, in, w, bias, scale, shift, out, N, Cin, plane, act, slope);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(conv2d_direct, dim3(grid_1d((int64_t)N * Cout * H * W, 256)), dim3(256), 0, s, in, w,
                       bias, scale, shift, out, N, Cin, H, W, Cout, KH, KW, PH, PW, act, slope);
    return hipGetLastError();
}

hipError_t launch_convT3x3s2_direct(const float* in, const float* w, const float* bias, const float* scale,
                                    const float* shift, float* out, int N, int Cin, int H, int W, int Cout,
                                    int act, float slope, hipStream_t s)
{
    hipLaunchKernelGGL(convT3x3s2_direct, dim3(grid_1d((int64_t)N * Cout * 4 * H * W, 256)), dim3(256), 0, s,
                       in, w, bias, scale, shift, out, N, Cin, H, W, Cout, act, slope);
    return hipGetLastError();
}

hipError_t launch_conv2d_wgrad_direct(const float* in, const float* g, float* gw, int N, int Cin, int H, int W,
                                      int Cout, int KH, int KW, int PH, int PW, hipStream_t s, int accumulate)
{
    if (KH == 1 && KW == 1 && PH == 0 && PW == 0) {
        hipLaunchKernelGGL(conv1x1_wgrad_direct, dim3((unsigned)(Cout * Cin)), dim3(1024), 0, s, in, g, gw, N, Cin,
                           (int64_t)H * W, Cout, accumulate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(conv2d_wgrad_direct, dim3((unsigned)(Cout * Cin)), dim3(256), 0, s, in, g, gw, N, Cin, H, W,
                       Cout, KH, KW, PH, PW, accumulate);
    return hipGetLastError();
}

hipError_t launch_convT3x3s2_wgrad_direct(const float* in, const float* g, float* gw, int N, int Cin, int H,
                                          int W, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(convT3x3s2_wgrad_direct, dim3((unsigned)(Cout * Cin)), dim3(256), 0, s, in, g, gw, N, Cin,
                       H, W, Cout);
    return hipGetLastError();
}

hipError_t launch_convT3x3s2_dgrad_direct(const float* g, const float* w, float* gin, int N, int Cin, int H,
                                          int W, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(convT3x3s2_dgrad_direct, dim3(grid_1d((int64_t)N * Cin * H * W, 256)), dim3(256), 0, s, g,
                       w, gin, N, Cin, H, W, Cout);
    return hipGetLastError();
}

// fixed-order sum of the weight slabs (and of bias_rows rows of bias partial sums when gb is given), shared with the bf16 kernel
hipError_t launch_conv3x3_wgrad_reduce(const float* slabs, float* gw, int Cin, int Cout, int CinP, int CoutP, int ksplit,
                                       const float* bias_slab, float* gb, int bias_rows, hipStream_t s, int accumulate)
{
    const int64_t cblk = (int64_t)CoutP * ((CinP + 63) / 64);
    const bool tsplit = cblk < 512;                          // few (co, ci) blocks: three workgroups per block (3 taps each; nine with one
                                                             // tap each measured 5 % slower on the 64-block layers: 0.074 vs 0.070 ms per call)
    int64_t rblocks = tsplit ? 3 * cblk : cblk;
    if (rblocks > 256 * 64) rblocks = 256 * 64;             // grid-stride beyond that
    const int bblocks = gb ? (CoutP + 63) / 64 : 0;         // extra blocks of the same launch add up the bias rows
    const int groups = ksplit >= 64 ? 16 : (ksplit >= 24 ? 8 : 4);      // slice groups per workgroup (a pure function of the slab count)
    if (accumulate & 2) {          // deferred: one grouped launch at the end of the backward pass (conv_kernels.h)
        wgrad_defer(WgradReduceJob{slabs, gw, bias_slab, gb, Cin, Cout, CinP, CoutP, ksplit, bias_rows, (int)rblocks, bblocks, groups,
                                   tsplit ? 1 : 0, accumulate & 1, 0});
        return hipSuccess;
    }
    if (tsplit)
        hipLaunchKernelGGL(conv3x3_wgrad_reduce<3>, dim3((unsigned)(rblocks + bblocks)), dim3(64 * groups), 0, s, slabs,
                           gw, Cin, Cout, CinP, CoutP, ksplit, bias_slab, gb, bias_rows, (int)rblocks, accumulate);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_reduce<9>, dim3((unsigned)(rblocks + bblocks)), dim3(64 * groups), 0, s, slabs,
                           gw, Cin, Cout, CinP, CoutP, ksplit, bias_slab, gb, bias_rows, (int)rblocks, accumulate);
    return hipGetLastError();
}

// ---- deferred reduce jobs: one list per process (one process per GPU; the autograd engine runs backward nodes and final callbacks on
// threads of its own, so the list is not thread-local), guarded by a mutex
static std::mutex g_wgrad_mu;
static std::vector<WgradReduceJob> g_wgrad_jobs;
void wgrad_defer(const WgradReduceJob& j) { std::lock_guard<std::mutex> l(g_wgrad_mu); g_wgrad_jobs.push_back(j); }
int wgrad_deferred_count() { std::lock_guard<std::mutex> l(g_wgrad_mu); return (int)g_wgrad_jobs.size(); }
void wgrad_deferred_drop() { std::lock_guard<std::mutex> l(g_wgrad_mu); g_wgrad_jobs.clear(); }
hipError_t wgrad_deferred_flush(hipStream_t s)
{
    std::vector<WgradReduceJob> jobs;
    { std::lock_guard<std::mutex> l(g_wgrad_mu); jobs.swap(g_wgrad_jobs); }
    for (size_t i0 = 0; i0 < jobs.size(); i0 += WGRAD_GROUP_MAX_JOBS) {
        WgradReduceGroup g;
        g.n = (int)std::min<size_t>(WGRAD_GROUP_MAX_JOBS, jobs.size() - i0);
        g.pad = 0;
        int64_t blocks = 0;
        for (int k = 0; k < g.n; ++k) {
            g.job[k] = jobs[i0 + k];
            g.job[k].block0 = (int)blocks;
            blocks += g.job[k].wblocks + g.job[k].bblocks;
        }
        if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(wgrad_reduce_group, dim3((unsigned)blocks), dim3(64 * RED_KG), 0, s, g);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

struct WgradPlan { int wco, wci, wk, rpw, wt, CinP, CoutP, ksplit, tx, ty, bparts; bool v2; };

static WgradPlan wgrad_plan(int N, int Cin, int H, int W, int Cout)
{
    // Candidates: the first-generation 4-wave 2x2 kernel (2-row tiles, two workgroups per CU) and the 8-wave kernels with K split
    // over the waves -- (2,2,2,2) / (2,2,2,1): 64 x 64 blocks on 4- / 2-row tiles; (1,2,4,1) / (2,1,4,1): 32 x 64 / 64 x 32 blocks;
    // (1,1,8,1): 32 x 32 blocks, eight waves per slab.  For each, and for each number of slabs, a small cost model:
    //   matrix time    = rounds x tiles per workgroup x (tile pixels x block co x block ci x 18 flop) / (70 % of a CU's fp32 MFMA rate)
    //   fixed cost     = 12 us per round of workgroups (first-tile latency, in-workgroup reduction, slab store)
    //   slab traffic   = slabs x 9 x CoutP x CinP x 4 B written and read back by the reduce launch, at 2.5 TB/s, + 4 us
    //   operand reads  = every co block re-reads the input tiles (with their halo rows), every ci block the gradient tiles, at 3 TB/s
    // and the cheapest wins.  What the model encodes is what the traces showed (profiles/r02): the deep layers of a 2-sample step
    // ran on 64 workgroups with 2 tiles each (52 us for 1.2 GFLOP), the thin layers paid 68-134 us of reduce for 256 slabs, and a
    // channel count <= 32 wasted whole quadrants.  Pure function of the problem size: the workspace query sees the same plan.
    // SSTEM_WGRAD_V2=0: the first-generation kernels only (A/B runs).
    static const int v2_knob = [] { const char* e = getenv("SSTEM_WGRAD_V2"); return e ? atoi(e) : 1; }();
    static const int small = [] { const char* e = getenv("SSTEM_WGRAD_SMALL"); return e ? atoi(e) : -1; }();
    static const int target_knob = [] { const char* e = getenv("SSTEM_WGRAD_TARGET"); return e && atoi(e) > 0 ? atoi(e) : 512; }();   // developer knob; measured on the
    // fusion step (profiles/r02): 1024 -> 512 halves the slab traffic of the reduce launch, batch 16 step 24.17 -> 23.68 ms, batch 2 unchanged
    static const int min_tiles = [] { const char* e = getenv("SSTEM_WGRAD_MIN_TILES"); return e && atoi(e) > 0 ? atoi(e) : 4; }();   // developer knob
    WgradPlan p;
    p.wt = 32;
    p.tx = (W + TW - 1) / TW;
    const int64_t tiles_2x32 = (int64_t)N * p.tx * ((H + WT_R - 1) / WT_R);
    if (const char* f = getenv("SSTEM_WGRAD_FORCE")) {       // developer knob (tools/sweep_wgrad.py): "wco,wci,wk,rpw,slabs", read at every call
        int a, b, c, d, k;
        if (sscanf(f, "%d,%d,%d,%d,%d", &a, &b, &c, &d, &k) == 5 && (a == 1 || a == 2) && (b == 1 || b == 2) && k >= 1 &&
            ((c == 1 && d == 2 && a == 2 && b == 2) || (a * b * c == 8 && (d == 1 || (d == 2 && a == 2 && b == 2 && c == 2))))) {
            p.wco = a; p.wci = b; p.wk = c; p.rpw = d; p.v2 = c > 1;
            const int bco = 32 * a, bci = 32 * b, tr = c * d;
            p.CinP = (Cin + bci - 1) / bci * bci; p.CoutP = (Cout + bco - 1) / bco * bco;
            p.ty = (H + tr - 1) / tr;
            const int64_t ntiles = (int64_t)N * p.tx * p.ty;
            p.ksplit = (int)(k < ntiles ? k : ntiles);
            p.bparts = p.v2 ? 512 / bco : 256 / bco;
            return p;
        }
    }
    if (v2_knob) {
        // Rules read off a sweep of every configuration x slab count on the layers of the SFF fusion step at 2 and 16 samples
        // (tools/sweep_wgrad.py, profiles/r02/e_wgrad_sweep.txt; a first version chose by a cost model and was 10-30 % off the
        // best measured point on most layers):
        //   a side of <= 32 channels      -> the 8-wave kernel without padded quadrants, one slab per workgroup, ~256 workgroups
        //   both sides >= 64 channels     -> the first-generation 2x2 kernel, ~512 workgroups (ties with (2,1,4,1) everywhere measured)
        //   ... unless its 2-row tiles are too few to give 256 workgroups (32x32 maps at small batch): 32x32 blocks, 8 waves per slab
        //   maps up to 16 pixels wide -> the 8-wave kernels with two image rows per MFMA row (SSTEM_CONV_NARROW=0: off)
        static const bool narrow_knob = [] { const char* e = getenv("SSTEM_CONV_NARROW"); return !(e && atoi(e) == 0); }();
        const bool narrow = narrow_knob && W <= 16;
        auto set = [&](int wco, int wci, int wk, int rpw, bool v2, int wg_target, int min_tiles_per_wg) {
            const int wt = (v2 && narrow) ? 16 : 32;
            const int bco = 32 * wco, bci = 32 * wci, tr = wk * rpw * (32 / wt);
            p.wco = wco; p.wci = wci; p.wk = wk; p.rpw = rpw; p.v2 = v2; p.wt = wt;
            p.tx = (W + wt - 1) / wt;
            p.CinP = (Cin + bci - 1) / bci * bci; p.CoutP = (Cout + bco - 1) / bco * bco;
            p.ty = (H + tr - 1) / tr;
            p.bparts = v2 ? 512 / bco : 256 / bco;
            const int64_t ntiles = (int64_t)N * p.tx * p.ty;
            const int64_t blocks = (int64_t)(p.CinP / bci) * (p.CoutP / bco);
            int64_t k = (wg_target + blocks - 1) / blocks;
            if (k > ntiles / min_tiles_per_wg) k = ntiles / min_tiles_per_wg;
            if (k < 1) k = 1;
            p.ksplit = (int)k;
            return blocks * k;
        };
        if (Cout <= 32 && Cin <= 32) { set(1, 1, 8, 1, true, 256, 2); return p; }
        if (Cout <= 32) { set(1, 2, 4, 1, true, 256, 2); return p; }
        if (Cin <= 32) { set(2, 1, 4, 1, true, 256, 1); return p; }
        if (narrow) { set(2, 2, 2, 2, true, 256, 1); return p; }
        if (set(2, 2, 1, 2, false, 512, 2) >= 256) return p;
        set(1, 1, 8, 1, true, 256, 1);
        return p;
    }
    p.v2 = false; p.wk = 1; p.rpw = WT_R;
    // first generation.  Workgroup shape measured on MI355X after the staging rewrite (tools/bench_wgrad.py, one box, two
    // repetitions): the one-wave 32x32 workgroup wins when BOTH channel counts are <= 32 and there are many pixel tiles, loses
    // when only one side is small and at small batch.  SSTEM_WGRAD_SMALL=0 / 1 forces never / whenever a side is <= 32 (A/B runs).
    const bool both_small = Cout <= 32 && Cin <= 32 && tiles_2x32 >= 4096;
    p.wco = (small == 1 ? Cout <= 32 : (small == -1 && both_small)) ? 1 : 2;
    p.wci = (small == 1 ? Cin <= 32 : (small == -1 && both_small)) ? 1 : 2;
    const int bco = 32 * p.wco, bci = 32 * p.wci;
    p.CinP = (Cin + bci - 1) / bci * bci;
    p.CoutP = (Cout + bco - 1) / bco * bco;
    p.ty = (H + WT_R - 1) / WT_R;
    p.bparts = (64 * p.wco * p.wci) / bco;
    const int64_t ntiles = (int64_t)N * p.tx * p.ty;
    const int blocks = (p.CinP / bci) * (p.CoutP / bco);
    // enough workgroups to fill the chip twice (smaller workgroups -> more of them), but keep at least
    // 4 pixel tiles per workgroup so the partial-slab traffic stays below the useful work
    const int target = target_knob * 4 / (p.wco * p.wci);
    int64_t k = (target + blocks - 1) / blocks;
    if (k > ntiles / min_tiles) k = ntiles / min_tiles;
    if (k < 1) k = 1;
    p.ksplit = (int)k;
    return p;
}

// weight slabs, then the bias partial sums: ksplit x (threads / channels per workgroup) rows of CoutP
int64_t conv3x3_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const WgradPlan p = wgrad_plan(N, Cin, H, W, Cout);
    return (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) + (int64_t)p.ksplit * 16 * p.CoutP;
}

template <int WCO, int WCI, int WK, int RPW, int WT = 32>
static hipError_t launch_wgrad_v2(const float* in, const float* g, float* workspace, int N, int Cin, int H, int W, int Cout,
                                  const WgradPlan& p, float* bias_slab, hipStream_t s)
{
    constexpr int NW = WCO * WCI * WK, TR = WK * RPW * (32 / WT);
    constexpr int stage = 32 * WCO * (TR * WT + 1) + 32 * WCI * (((TR + 2) * (WT + 2)) | 1);
    constexpr int red = NW * 1024;
    constexpr size_t lds_bytes = (size_t)(stage > red ? stage : red) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = conv3x3_wgrad_mfma_v2<WCO, WCI, WK, RPW, WT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    const int blocks = (p.CinP / (32 * WCI)) * (p.CoutP / (32 * WCO));
    hipLaunchKernelGGL(k, dim3((unsigned)(blocks * p.ksplit)), dim3(64 * NW), lds_bytes, s, in, g, workspace, N, Cin, H, W, Cout,
                       p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wgrad_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                     int H, int W, int Cout, hipStream_t s, int accumulate)
{
    const WgradPlan p = wgrad_plan(N, Cin, H, W, Cout);
    static const bool plan_debug = [] { const char* e = getenv("SSTEM_WGRAD_PLAN_DEBUG"); return e && atoi(e) != 0; }();   // developer knob
    if (plan_debug)
        fprintf(stderr, "wgrad plan N=%d %d->%d %dx%d: %s (%d,%d,%d,%d) blocks %d x slabs %d\n", N, Cin, Cout, H, W, p.v2 ? "v2" : "v1", p.wco, p.wci,
                p.wk, p.rpw, (p.CinP / (32 * p.wci)) * (p.CoutP / (32 * p.wco)), p.ksplit);
    float* bias_slab = gb ? workspace + (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) : nullptr;
    const int bias_rows = p.ksplit * p.bparts;
    hipError_t e;
    if (p.v2 && p.wt == 16) {
        if (p.wco == 1 && p.wci == 1) e = launch_wgrad_v2<1, 1, 8, 1, 16>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.wco == 1) e = launch_wgrad_v2<1, 2, 4, 1, 16>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.wci == 1) e = launch_wgrad_v2<2, 1, 4, 1, 16>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.rpw == 2) e = launch_wgrad_v2<2, 2, 2, 2, 16>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else e = launch_wgrad_v2<2, 2, 2, 1, 16>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
    } else if (p.v2) {
        if (p.wco == 1 && p.wci == 1) e = launch_wgrad_v2<1, 1, 8, 1>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.wco == 1) e = launch_wgrad_v2<1, 2, 4, 1>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.wci == 1) e = launch_wgrad_v2<2, 1, 4, 1>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else if (p.rpw == 2) e = launch_wgrad_v2<2, 2, 2, 2>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
        else e = launch_wgrad_v2<2, 2, 2, 1>(in, g, workspace, N, Cin, H, W, Cout, p, bias_slab, s);
    } else {
        const int blocks = (p.CinP / (32 * p.wci)) * (p.CoutP / (32 * p.wco));
        const dim3 grid((unsigned)(blocks * p.ksplit));
#define SSTEM_WGRAD(A, B)                                                                                  \
    hipLaunchKernelGGL((conv3x3_wgrad_mfma<A, B>), grid, dim3(64 * A * B), 0, s, in, g, workspace, N, Cin, H, W, \
                       Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab)
        if (p.wco == 2 && p.wci == 2) SSTEM_WGRAD(2, 2);
        else if (p.wco == 2) SSTEM_WGRAD(2, 1);
        else if (p.wci == 2) SSTEM_WGRAD(1, 2);
        else SSTEM_WGRAD(1, 1);
#undef SSTEM_WGRAD
        e = hipGetLastError();
    }
    if (e != hipSuccess) return e;
    return launch_conv3x3_wgrad_reduce(workspace, gw, Cin, Cout, p.CinP, p.CoutP, p.ksplit, bias_slab, gb, bias_rows, s, accumulate);
}

}  // namespace sstem
