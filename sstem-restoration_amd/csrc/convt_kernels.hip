// ConvTranspose2d(k=3, s=2, p=1, output_padding=1) for MI355X: forward, data gradient, weight (+ bias) gradient on the
// fp32 matrix cores, by output parity ("sub-pixel") decomposition -- no zero-inserted tensor, no discarded results.
//
// Replaces what the reference delegates to cuDNN through nn.ConvTranspose2d in
//   sff_scripts_fusion/model/model_unet.py:32,70            (bottleneck / expansive blocks: ConvTranspose + BN + ReLU)
//   sff_scripts_fusion/model/model_fusionnet.py:21-27       (conv_trans_block)
//
//   out[n,co,Y,X] = sum_ci sum_{ky,kx} in[n,ci,y,x] * W[ci,co,ky,kx],   Y = 2y - 1 + ky,  X = 2x - 1 + kx,  output 2H x 2W.
// For output parity (Y & 1, X & 1) = (py, px) only some taps contribute:
//   py = 0: ky = 1 (y = Y/2)                    py = 1: ky = 0 (y = (Y+1)/2) and ky = 2 (y = (Y-1)/2)       (same in x)
// i.e. 1 + 2 + 2 + 4 = 9 multiply-adds per 2x2 output block and channel pair: the same MFMA count as a 3x3 convolution at the
// INPUT resolution (the zero-insert route runs one at the output resolution: 4x the flops, a 4x-sized zero tensor written and
// read, and 3/4 of its data gradient thrown away).
//
// Forward  convT3x3s2_mfma:   D[co][input pixel] += W[co][k] * In[k][pixel (+1 row / +1 col)], four accumulator sets (one per
//          parity), k = (ci, tap); workgroup = 4 waves, tile = 8 x 32 input pixels = 16 x 64 output pixels x 32 channels.
// Dgrad    convT3x3s2_dgrad_mfma: a stride-2 3x3 convolution of g; the g tile sits in LDS with even and odd columns apart, so
//          the stride-2 reads of 32 consecutive pixels are consecutive words.
// Wgrad    convT3x3s2_wgrad_mfma: gW[ci][co][tap] = sum over input pixels of in[ci][p] * g[co][2p - 1 + tap]; M = co, N = ci,
//          K = pixels, nine accumulator tiles; g tile parity-split in LDS; the two pixel rows of a tile on two waves, added inside
//          the workgroup; split over pixel tiles into slabs, fixed-order reduce.
// v_mfma_f32_32x32x2_f32 throughout: exact fp32 products, k-ordered sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_kernels.h"

namespace sstem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KC = 8;                 // channels of the reduction per K chunk
constexpr int KK = KC * 9;            // 72 k-values per chunk
constexpr int TH = 8, TW = 32;        // tile in INPUT pixels (forward) / gradient-input pixels (dgrad)
constexpr int CO = 32;                // output channels per workgroup

__device__ __forceinline__ float act_apply(float v, int act, float slope)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

inline int grid_1d(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- weight packing (same layout as conv_kernels.hip: Wp[cb][chunk][k'][32], k' = (cl%4)*9 + tap + 36*(cl/4)) ------------
// W is the ConvTranspose weight [Cin][Cout][3][3].
//   forward: rows = co, reduction = ci:  Wp = W[ci][co][tap]
//   dgrad  : rows = ci, reduction = co:  Wp = W[ci][co][tap]      (reduce_over_first = 0)
__global__ void convT_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int nchunks, int ncb,
                                   int rows_are_cout)
{
    const int64_t total = (int64_t)ncb * nchunks * KK * CO;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = idx % CO;
        int64_t r = idx / CO;
        const int kp = r % KK; r /= KK;
        const int chunk = r % nchunks;
        const int cb = r / nchunks;
        const int half = kp / 36, rem = kp % 36;
        const int cl = rem / 9 + 4 * half, tap = rem % 9;
        const int kch = chunk * KC + cl, row = cb * CO + col;
        float v = 0.f;
        if (rows_are_cout) { if (kch < Cin && row < Cout) v = w[((int64_t)kch * Cout + row) * 9 + tap]; }
        else               { if (kch < Cout && row < Cin) v = w[((int64_t)row * Cout + kch) * 9 + tap]; }
        wp[idx] = v;
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
constexpr int F_R = TH + 1, F_PW = 34;                 // 9 input rows (one below the tile), 33 columns used of a 34 pitch
constexpr int F_TILE = KC * F_R * F_PW;                // 2448 floats
constexpr int F_WT = KK * CO;                          // 2304 floats
constexpr int F_BUF = F_TILE + F_WT;

__global__ __launch_bounds__(256, 2) void convT3x3s2_mfma(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int act, float slope,
    int ksplit, float* __restrict__ slab, ConvExtra ex)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int ks = blockIdx.z % ksplit, n = (blockIdx.z / ksplit) / ncb, cb = (blockIdx.z / ksplit) % ncb;
    const int X0 = bx * TW, Y0 = by * TH;
    const int cpk = nchunks / ksplit;
    const int c_first = ks * cpk, c_end = c_first + cpk;
    const int64_t plane = (int64_t)H * W;
    const int OH = 2 * H, OW = 2 * W;
    const int64_t oplane = (int64_t)OH * OW;

    constexpr int IN_PER_T = (F_TILE + 255) / 256;      // 10
    constexpr int W_V4 = F_WT / 4;                      // 576
    constexpr int W_PER_T = (W_V4 + 255) / 256;         // 3
    uint32_t in_off[IN_PER_T];
    int in_cl[IN_PER_T];
#pragma unroll
    for (int k = 0; k < IN_PER_T; ++k) {
        const int e = tid + 256 * k;
        const int cl = e / (F_R * F_PW);
        const int rem = e - cl * (F_R * F_PW);
        const int r = rem / F_PW, cc = rem - r * F_PW;
        const int y = Y0 + r, x = X0 + cc;
        const bool ok = (e < F_TILE) && y < H && x < W;
        in_off[k] = ok ? ((uint32_t)cl * (uint32_t)plane + (uint32_t)(y * W + x)) * 4u : 0u;
        in_cl[k] = ok ? cl : -1;
    }
    const float* in_n = in + (int64_t)n * Cin * plane;
    const float* wp_cb = wp + (int64_t)cb * nchunks * F_WT;

    float in_r[IN_PER_T];
    f32x4 w_r[W_PER_T];
    auto stage_load = [&](int chunk) {
        const float* cbase = in_n + (int64_t)chunk * KC * plane;
        const int cl_lim = Cin - chunk * KC;
#pragma unroll
        for (int k = 0; k < IN_PER_T; ++k) {
            float v = 0.f;
            if (in_cl[k] >= 0 && in_cl[k] < cl_lim)
                v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cbase) + in_off[k]);
            in_r[k] = v;
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(wp_cb + (int64_t)chunk * F_WT);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            w_r[k] = (e < W_V4) ? src[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](int buf) {
        float* b = lds + buf * F_BUF;
#pragma unroll
        for (int k = 0; k < IN_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < F_TILE) b[e] = in_r[k];
        }
        f32x4* wdst = reinterpret_cast<f32x4*>(b + F_TILE);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < W_V4) wdst[e] = w_r[k];
        }
    };

    f32x16 acc[4][2];                    // [parity py*2+px][row of the wave]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[p][rr][q] = 0.f;

    stage_load(c_first);
    stage_store(0);
    __syncthreads();

    const int b_base = h * (4 * F_R * F_PW) + (2 * wave) * F_PW + j;
    const int a_base = F_TILE + h * (36 * CO) + j;

    for (int c = c_first; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) stage_load(c + 1);
        const float* buf = lds + ((c - c_first) & 1) * F_BUF;
        const float* bp = buf + b_base;
        const float* ap = buf + a_base;
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int cl = s / 9, ky = (s % 9) / 3, kx = s % 3;
            const int py = (ky == 1) ? 0 : 1, dy = (ky == 0) ? 1 : 0;     // Y = 2y - 1 + ky
            const int px = (kx == 1) ? 0 : 1, dx = (kx == 0) ? 1 : 0;
            const float a = ap[s * CO];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float b = bp[(cl * F_R + dy + rr) * F_PW + dx];
                acc[py * 2 + px][rr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[py * 2 + px][rr], 0, 0, 0);
            }
        }
        if (more) stage_store((c + 1 - c_first) & 1);
        __syncthreads();
    }

    // ---- epilogue: acc[py*2+px][rr][q] = out[co = cb*32 + (q&3) + 8*(q>>2) + 4*h][2*(Y0+2*wave+rr) + py][2*(X0+j) + px]
    const int x = X0 + j;
    const bool xin = x < W;
    if (ex.bn_part && ksplit == 1) {
        float* red = lds;                                                     // [4 waves][32]
        const int rows_in = min(TH, H - Y0), cols_in = min(TW, W - X0);
        const float cnt = (float)(4 * rows_in * cols_in);
        const int tile = (n * gridDim.y + by) * gridDim.x + bx;
        float mean_q[16], part[16];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = cb * CO + (q & 3) + 8 * (q >> 2) + 4 * h;
                const float bsv = (bias && co < Cout) ? bias[co] : 0.f;
                float sacc = 0.f;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const bool in_img = xin && (Y0 + 2 * wave + rr) < H;
                        const float v = acc[p][rr][q] + bsv;
                        const float d = pass == 0 ? v : (v - mean_q[q]) * (v - mean_q[q]);
                        sacc += in_img ? d : 0.f;
                    }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
                part[q] = sacc;
            }
            float mine = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) mine = (j == q) ? part[q] : mine;
            if (j < 16) red[wave * 32 + h * 16 + j] = mine;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float tot = ((red[0 * 32 + h * 16 + q] + red[1 * 32 + h * 16 + q]) + red[2 * 32 + h * 16 + q]) + red[3 * 32 + h * 16 + q];
                if (pass == 0) mean_q[q] = tot / cnt;
                else part[q] = tot;
            }
            __syncthreads();
        }
        if (wave == 0 && j < 16) {
            float m2 = 0.f, mn = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { m2 = (j == q) ? part[q] : m2; mn = (j == q) ? mean_q[q] : mn; }
            const int co = cb * CO + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (co < Cout) {
                float* dst = ex.bn_part + ((int64_t)co * ex.bn_tiles + tile) * 3;
                dst[0] = cnt; dst[1] = mn; dst[2] = m2;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int co = cb * CO + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (co >= Cout) continue;
        const float bs = (ksplit == 1 && bias) ? bias[co] : 0.f;
        const float sc = (ksplit == 1 && scale) ? scale[co] : 1.f;
        const float sh = (ksplit == 1 && shift) ? shift[co] : 0.f;
        float* obase = (ksplit > 1) ? slab + (((int64_t)ks * N + n) * Cout + co) * oplane : out + ((int64_t)n * Cout + co) * oplane;
        const float* rbase = (ksplit == 1 && ex.residual) ? ex.residual + ((int64_t)n * Cout + co) * oplane : nullptr;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int y = Y0 + 2 * wave + rr;
            if (!(xin && y < H)) continue;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int64_t o = (int64_t)(2 * y + py) * OW + 2 * x;            // even: 8-byte aligned pair (px = 0, 1)
                f32x2 v;
                v[0] = acc[py * 2 + 0][rr][q]; v[1] = acc[py * 2 + 1][rr][q];
                if (ksplit == 1) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) v[e] = act_apply((v[e] + bs) * sc + sh, act, slope);
                    if (rbase) {
                        const f32x2 r = *reinterpret_cast<const f32x2*>(rbase + o);
                        v[0] = (v[0] + r[0]) * ex.res_scale; v[1] = (v[1] + r[1]) * ex.res_scale;
                    }
                }
                *reinterpret_cast<f32x2*>(obase + o) = v;
            }
        }
    }
}

// =====================================================================================================================
// data gradient: gin[n,ci,y,x] = sum_co sum_{ky,kx} g[n,co,2y-1+ky,2x-1+kx] * W[ci,co,ky,kx]
// =====================================================================================================================
constexpr int D_R = 2 * TH + 1;          // 17 rows of g: 2*Y0 - 1 .. 2*Y0 + 15
constexpr int D_HALF = 36;               // words per column parity (33 used: tile columns 2*X0 - 1 .. 2*X0 + 63 -> 33 odd-offset, 32 even)
constexpr int D_PW = 2 * D_HALF;         // 72: row pitch; 4 channels * 17 rows * 72 = 4896 = 32 (mod 64): the two lane halves hit disjoint banks
constexpr int D_TILE = KC * D_R * D_PW;  // 9792 floats
constexpr int D_WT = KK * CO;
// LDS word of tile column tc (0..64, tc = 0 is image column 2*X0 - 1): parity tc & 1 apart, so lane j's read of column 2j + kx is
// word (kx & 1) * 36 + j + (kx >> 1): consecutive in j
__device__ __forceinline__ int d_col(int tc) { return (tc & 1) * D_HALF + (tc >> 1); }

__global__ __launch_bounds__(256, 2) void convT3x3s2_dgrad_mfma(
    const float* __restrict__ g, const float* __restrict__ wp, float* __restrict__ gin,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int ksplit, float* __restrict__ slab)
{
    // single LDS buffer (39 KB tile + 9 KB weights: three workgroups per CU); the next chunk is prefetched into registers
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int ks = blockIdx.z % ksplit, n = (blockIdx.z / ksplit) / ncb, cb = (blockIdx.z / ksplit) % ncb;
    const int X0 = bx * TW, Y0 = by * TH;
    const int cpk = nchunks / ksplit;
    const int c_first = ks * cpk, c_end = c_first + cpk;
    const int OH = 2 * H, OW = 2 * W;
    const int64_t oplane = (int64_t)OH * OW, plane = (int64_t)H * W;

    constexpr int ROW_E = 65;                                   // tile columns per row
    constexpr int G_E = KC * D_R * ROW_E;                       // 8840 elements to stage per chunk
    constexpr int G_PER_T = (G_E + 255) / 256;                  // 35
    constexpr int W_V4 = D_WT / 4, W_PER_T = (W_V4 + 255) / 256;
    const float* g_n = g + (int64_t)n * Cout * oplane;
    const float* wp_cb = wp + (int64_t)cb * nchunks * D_WT;

    float g_r[G_PER_T];
    f32x4 w_r[W_PER_T];
    auto stage_load = [&](int chunk) {
        const float* cbase = g_n + (int64_t)chunk * KC * oplane;
        const int cl_lim = Cout - chunk * KC;
#pragma unroll
        for (int k = 0; k < G_PER_T; ++k) {
            const int e = tid + 256 * k;
            const int cl = e / (D_R * ROW_E);
            const int rem = e - cl * (D_R * ROW_E);
            const int r = rem / ROW_E, tc = rem - r * ROW_E;
            const int yy = 2 * Y0 - 1 + r, xx = 2 * X0 - 1 + tc;
            const bool ok = e < G_E && cl < cl_lim && yy >= 0 && yy < OH && xx >= 0 && xx < OW;
            g_r[k] = ok ? cbase[(int64_t)cl * oplane + (int64_t)yy * OW + xx] : 0.f;
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(wp_cb + (int64_t)chunk * D_WT);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            w_r[k] = (e < W_V4) ? src[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < G_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < G_E) {
                const int cl = e / (D_R * ROW_E);
                const int rem = e - cl * (D_R * ROW_E);
                const int r = rem / ROW_E, tc = rem - r * ROW_E;
                lds[(cl * D_R + r) * D_PW + d_col(tc)] = g_r[k];
            }
        }
        f32x4* wdst = reinterpret_cast<f32x4*>(lds + D_TILE);
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) {
            const int e = tid + 256 * k;
            if (e < W_V4) wdst[e] = w_r[k];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[rr][q] = 0.f;

    stage_load(c_first);
    stage_store();
    __syncthreads();

    const int b_base = h * (4 * D_R * D_PW) + (2 * (2 * wave)) * D_PW + j;       // tile row of (y = Y0 + 2*wave, ky = 0) is 2*(2*wave)
    const int a_base = D_TILE + h * (36 * CO) + j;
    for (int c = c_first; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) stage_load(c + 1);
        const float* bp = lds + b_base;
        const float* ap = lds + a_base;
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int cl = s / 9, ky = (s % 9) / 3, kx = s % 3;
            const float a = ap[s * CO];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float b = bp[(cl * D_R + 2 * rr + ky) * D_PW + (kx & 1) * D_HALF + (kx >> 1)];
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[rr], 0, 0, 0);
            }
        }
        __syncthreads();                 // everyone is done reading this chunk
        if (more) { stage_store(); __syncthreads(); }
    }

    const int x = X0 + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int ci = cb * CO + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (ci >= Cin) continue;
        float* obase = (ksplit > 1) ? slab + (((int64_t)ks * N + n) * Cin + ci) * plane : gin + ((int64_t)n * Cin + ci) * plane;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int y = Y0 + 2 * wave + rr;
            if (y < H && x < W) obase[(int64_t)y * W + x] = acc[rr][q];
        }
    }
}

// plain fixed-order sum of split-K slabs (data gradient: no epilogue arithmetic)
__global__ __launch_bounds__(256) void convT_sum_slabs(const float* __restrict__ slab, float* __restrict__ out, int64_t total, int ksplit)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
        out[i] = v;
    }
}

// split-K epilogue of the forward (same arithmetic as the unsplit store)
__global__ __launch_bounds__(256) void convT_splitk_epilogue(
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
        v = act_apply(v, act, slope);
        if (residual) v = (v + residual[i]) * res_scale;
        out[i] = v;
    }
}

// =====================================================================================================================
// weight (+ bias) gradient: gW[ci][co][ky][kx] = sum_{n,y,x} in[n,ci,y,x] * g[n,co,2y-1+ky,2x-1+kx]
// =====================================================================================================================
// Workgroup = 4 waves = 32 co x 64 ci x two halves of the pixel tile (wave = (ci half wj, pixel row kr)); pixel tile = 2 input
// rows x 32 columns.  g tile per channel: 5 rows (2*Y0 - 1 .. 2*Y0 + 3) x 65 columns, columns parity-split as in the dgrad
// kernel.  D[co][ci] += g[co][2p - 1 + tap] * in[ci][p]: nine accumulator tiles per wave.
constexpr int WG_R = 5, WG_PW = 2 * D_HALF;                  // 5 rows x 72 words
constexpr int WG_GP = WG_R * WG_PW + 1;                      // 361: odd pitch between channels (column reads of 32 channels: no conflicts)
constexpr int WG_IP = 2 * TW + 1;                            // 65: in tile [ci][2 rows x 32 cols]
constexpr int WGT_CO = 32, WGT_CI = 64;
constexpr int WG_GE = WGT_CO * WG_R * 65;                    // 10400 g elements per tile
constexpr int WG_T = 512;                                    // threads: 8 waves = 2 ci halves x 2 pixel rows x 2 column halves
constexpr int WG_GK = (WG_GE + WG_T - 1) / WG_T;             // 21 per thread
constexpr int WG_IK = WGT_CI * 64 / WG_T;                    // 8 input elements per thread
constexpr int WG_STAGE = WGT_CO * WG_GP + WGT_CI * WG_IP;    // 15712 floats = 62.8 KB
constexpr int WG_RED = 8 * 1024;                             // in-workgroup reduction: one 32x32 tile per wave

__global__ __launch_bounds__(WG_T, 1) void convT3x3s2_wgrad_mfma(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ slab,
    int N, int Cin, int H, int W, int Cout, int CinP, int CoutP, int ksplit, int tiles_x, int tiles_y,
    float* __restrict__ bias_slab)
{
    // Staging with the loads of tile t+1 in flight during tile t's MFMAs (29 registers per thread; the first version loaded and
    // stored element by element: 110 us per layer at batch 2, 240 us at batch 16).  The 64 pixels of a tile are split over four
    // waves (row kr, column half kc); their accumulators are added inside the workgroup (wave order: fixed) and ONE slab leaves it.
    __shared__ __attribute__((aligned(16))) float lds_w[WG_STAGE > WG_RED ? WG_STAGE : WG_RED];
    float* g_t = lds_w;
    float* i_t = lds_w + WGT_CO * WG_GP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int wj = wave & 1, kr = (wave >> 1) & 1, kc = wave >> 2;     // ci half, pixel row, column half of the tile
    const int nib = CinP / WGT_CI;
    const int blk = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
    const int cb = blk / nib, ib = blk % nib;
    const int OH = 2 * H, OW = 2 * W;
    const int64_t plane = (int64_t)H * W, oplane = (int64_t)OH * OW;
    const int ntiles = N * tiles_y * tiles_x;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const bool do_bias = (bias_slab != nullptr) && (ib == 0);
    float bsum = 0.f;                                            // thread (channel = tid % 32, part = tid / 32): 16 parts

    auto geometry = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        const int tx = tile % tiles_x;
        const int r0 = tile / tiles_x;
        n = r0 / tiles_y; X0 = tx * TW; Y0 = (r0 % tiles_y) * 2;
    };
    float gv[WG_GK], iv[WG_IK];
    auto issue = [&](int tile) __attribute__((always_inline)) {          // loads only: uniform 64-bit base + 32-bit lane offset, clamped
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        int t0 = tid;
        asm volatile("" : "+v"(t0));      // the per-element index arithmetic is redone per tile: hoisted out of the loop it held ~60 registers and spilled
        const char* in_b = reinterpret_cast<const char*>(in + ((int64_t)n * Cin + ib * WGT_CI) * plane);      // 64 * plane * 4 < 4 GiB (launcher)
        const char* g_b = reinterpret_cast<const char*>(g + ((int64_t)n * Cout + cb * WGT_CO) * oplane);      // 32 * oplane * 4 < 4 GiB
        const uint32_t plane_u = (uint32_t)plane, oplane_u = (uint32_t)oplane;
#pragma unroll
        for (int k = 0; k < WG_IK; ++k) {
            const int e = t0 + WG_T * k;
            const int c = e >> 6, p = e & 63;
            const int y = Y0 + (p >> 5), x = X0 + (p & 31);
            const bool ok = ib * WGT_CI + c < Cin && y < H && x < W;
            const uint32_t off = ok ? ((uint32_t)c * plane_u + (uint32_t)(y * W + x)) * 4u : 0u;
            iv[k] = *reinterpret_cast<const float*>(in_b + off);
        }
#pragma unroll
        for (int k = 0; k < WG_GK; ++k) {
            const int e = t0 + WG_T * k;
            const int c = e / (WG_R * 65);
            const int rem = e - c * (WG_R * 65);
            const int r = rem / 65, tc = rem - r * 65;
            const int yy = 2 * Y0 - 1 + r, xx = 2 * X0 - 1 + tc;
            const bool ok = e < WG_GE && cb * WGT_CO + c < Cout && yy >= 0 && yy < OH && xx >= 0 && xx < OW;
            const uint32_t off = ok ? ((uint32_t)c * oplane_u + (uint32_t)(yy * OW + xx)) * 4u : 0u;
            gv[k] = *reinterpret_cast<const float*>(g_b + off);
        }
    };
    auto commit = [&](int tile) __attribute__((always_inline)) {         // masks recomputed from the tile index + LDS stores
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        int t0 = tid;
        asm volatile("" : "+v"(t0));      // the per-element index arithmetic is redone per tile: hoisted out of the loop it held ~60 registers and spilled
#pragma unroll
        for (int k = 0; k < WG_IK; ++k) {
            const int e = t0 + WG_T * k;
            const int c = e >> 6, p = e & 63;
            const bool ok = ib * WGT_CI + c < Cin && (Y0 + (p >> 5)) < H && (X0 + (p & 31)) < W;
            i_t[c * WG_IP + p] = ok ? iv[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < WG_GK; ++k) {
            const int e = t0 + WG_T * k;
            const int c = e / (WG_R * 65);
            const int rem = e - c * (WG_R * 65);
            const int r = rem / 65, tc = rem - r * 65;
            const int yy = 2 * Y0 - 1 + r, xx = 2 * X0 - 1 + tc;
            // rows / columns that belong to input pixels outside the image contribute nothing: those in[] entries are 0
            const bool ok = cb * WGT_CO + c < Cout && yy >= 0 && yy < OH && xx >= 0 && xx < OW;
            if (e < WG_GE) g_t[c * WG_GP + r * WG_PW + d_col(tc)] = ok ? gv[k] : 0.f;
        }
    };

    if (ks < ntiles) issue(ks);
    for (int tile = ks; tile < ntiles; tile += ksplit) {
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        commit(tile);
        __syncthreads();
        if (tile + ksplit < ntiles) issue(tile + ksplit);                 // in flight during this tile's MFMAs
        if (do_bias) {
            // bias gradient = sum of g over the OUTPUT pixels of this tile: rows 2*Y0 .. 2*Y0+3 (tile rows 1..4), columns 2*X0 .. 2*X0+63
            // (tile columns 1..64); thread = (channel tid % 32, part tid / 32 of 16): part p adds tile columns 1 + 4p .. 4 + 4p
            // (positions outside the image hold zeros)
            const int c = tid & 31, part = tid >> 5;
#pragma unroll
            for (int r = 1; r < 5; ++r)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) bsum += g_t[c * WG_GP + r * WG_PW + d_col(1 + 4 * part + k2)];
        }
        // ---- MFMAs: this wave's half (kc) of pixel row kr: 16 pixels = 8 k-steps x 9 taps
        const float* gp = g_t + j * WG_GP;                        // A: row = co j, k = pixel (lane half h = odd pixel of the pair)
        const float* ip = i_t + (wj * 32 + j) * WG_IP + kr * 32;  // B: col = ci
#pragma unroll 4
        for (int s2 = 0; s2 < 8; ++s2) {
            const int px = kc * 16 + 2 * s2 + h;                   // pixel column within the tile row (k index of this lane half)
            const float b = ip[px];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                const float a = gp[(2 * kr + ky) * WG_PW + (kx & 1) * D_HALF + px + (kx >> 1)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- the four (kr, kc) accumulators of every ci half added (fixed order), one slab per workgroup: slab[ks][tap][co][ci]
    float* red = lds_w;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave * 1024 + ((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + j] = acc[t][q];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {                            // 2 ci halves x 1024 elements over 512 threads
            const int o = tid + WG_T * e;
            const int half = o >> 10, rc = o & 1023;
            const float v = ((red[half * 1024 + rc] + red[(2 + half) * 1024 + rc]) + red[(4 + half) * 1024 + rc]) + red[(6 + half) * 1024 + rc];   // wave = kc*4 + kr*2 + wj
            const int co = cb * WGT_CO + (rc >> 5);
            const int ci = ib * WGT_CI + half * 32 + (rc & 31);
            slab[(((int64_t)ks * 9 + t) * CoutP + co) * CinP + ci] = v;
        }
        __syncthreads();
    }
    if (do_bias) bias_slab[((int64_t)ks * 16 + (tid >> 5)) * CoutP + cb * WGT_CO + (tid & 31)] = bsum;
}

// Fixed-order sum of the slices (16 slice groups per workgroup, as conv3x3_wgrad_reduce), written (or added) to gW[ci][co][3][3];
// the bias rows likewise to gb[co].
constexpr int TRED_KG = 16;
__global__ __launch_bounds__(64 * TRED_KG) void convT_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ gw, int Cin, int Cout,
                                                                   int CinP, int CoutP, int nslices, const float* __restrict__ bias_slab,
                                                                   float* __restrict__ gb, int bias_rows, int wblocks, int accumulate)
{
    __shared__ float part[TRED_KG][64];
    // 4, 8 or 16 slice groups: follows the slice count (launcher); the body is shared with the grouped launch (conv_kernels.h)
    convT_wgrad_reduce_body(slab, gw, Cin, Cout, CinP, CoutP, nslices, bias_slab, gb, bias_rows, wblocks, accumulate, (int)blockIdx.x,
                            (int)(blockDim.x >> 6), part);
}

// K slices of the reduction channels on small grids (as conv3x3_ksplit)
int convT_ksplit(int64_t wgs, int nchunks)
{
    static const bool off = [] { const char* e = getenv("SSTEM_CONV_KSPLIT"); return e && atoi(e) == 0; }();
    if (off) return 1;
    int ks = 1;
    while (wgs * ks < 512 && ks < 8 && nchunks % (ks * 2) == 0 && nchunks / (ks * 2) >= 2) ks *= 2;
    return ks;
}

struct WgPlan { int CinP, CoutP, ksplit, tx, ty; };
WgPlan convT_wgrad_plan(int N, int Cin, int H, int W, int Cout)
{
    WgPlan p;
    p.CinP = (Cin + WGT_CI - 1) / WGT_CI * WGT_CI;
    p.CoutP = (Cout + WGT_CO - 1) / WGT_CO * WGT_CO;
    p.tx = (W + TW - 1) / TW;
    p.ty = (H + 1) / 2;
    const int64_t ntiles = (int64_t)N * p.tx * p.ty;
    const int blocks = (p.CinP / WGT_CI) * (p.CoutP / WGT_CO);
    int64_t k = (256 + blocks - 1) / blocks;          // ~256 workgroups of 8 waves (one per CU); at least two tiles each (the prefetch)
    if (k > ntiles / 2) k = ntiles / 2;
    if (k < 1) k = 1;
    p.ksplit = (int)k;
    return p;
}

}  // namespace

// ---- host interface -------------------------------------------------------------------------------------------------
static inline int64_t packed_floats(int rows, int kch)
{
    return (int64_t)((rows + CO - 1) / CO) * ((kch + KC - 1) / KC) * KK * CO;
}

int64_t convT3x3s2_forward_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    const int ks = convT_ksplit((int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N * ncb, nchunks);
    return packed_floats(Cout, Cin) + (ks > 1 ? (int64_t)ks * N * Cout * 4 * H * W : 0);
}

int64_t convT3x3s2_dgrad_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const int ncb = (Cin + CO - 1) / CO, nchunks = (Cout + KC - 1) / KC;
    const int ks = convT_ksplit((int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N * ncb, nchunks);
    return packed_floats(Cin, Cout) + (ks > 1 ? (int64_t)ks * N * Cin * H * W : 0);
}

int64_t convT3x3s2_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const WgPlan p = convT_wgrad_plan(N, Cin, H, W, Cout);
    return (int64_t)p.ksplit * 9 * p.CoutP * p.CinP + (int64_t)p.ksplit * 16 * p.CoutP;
}

int64_t convT3x3s2_bn_partials(int N, int Cin, int H, int W, int Cout)
{
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    const int ks = convT_ksplit((int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N * ncb, nchunks);
    if (ks > 1) return 0;                    // a split launch leaves the statistics to the BatchNorm's own pass
    return (int64_t)N * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
}

hipError_t launch_convT3x3s2_mfma(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                  float* out, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout,
                                  int act, float slope, int prepacked, hipStream_t s, const ConvExtra& ex_in)
{
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + KC - 1) / KC;
    const int64_t wtotal = (int64_t)ncb * nchunks * KK * CO;
    hipError_t e;
    if (!prepacked) {
        hipLaunchKernelGGL(convT_pack_weights, dim3(grid_1d(wtotal, 256)), dim3(256), 0, s, w, workspace, Cin, Cout, nchunks, ncb, 1);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const dim3 tiles((W + TW - 1) / TW, (H + TH - 1) / TH);
    int ksplit = convT_ksplit((int64_t)tiles.x * tiles.y * N * ncb, nchunks);
    const int64_t out_elems = (int64_t)N * Cout * 4 * H * W;
    if (ksplit > 1 && workspace_floats < wtotal + (int64_t)ksplit * out_elems) ksplit = 1;
    ConvExtra ex = ex_in;
    if (ex.bn_part && (ksplit > 1 || scale || shift || act != 0 || ex.residual)) return hipErrorInvalidValue;
    ex.bn_tiles = (int)(N * tiles.x * tiles.y);
    if ((int64_t)N * ncb * ksplit > 65535) return hipErrorInvalidValue;
    float* slab = workspace + wtotal;
    const size_t lds_bytes = 2 * (size_t)F_BUF * sizeof(float);
    auto k = convT3x3s2_mfma;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tiles.x, tiles.y, (unsigned)(N * ncb * ksplit)), dim3(256), lds_bytes, s, in, workspace, bias, scale,
                       shift, out, N, Cin, H, W, Cout, nchunks, ncb, act, slope, ksplit, slab, ex);
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    hipLaunchKernelGGL(convT_splitk_epilogue, dim3(grid_1d(out_elems, 256)), dim3(256), 0, s, slab, bias, scale, shift, out, out_elems,
                       (int64_t)4 * H * W, Cout, ksplit, act, slope, ex.residual, ex.res_scale);
    return hipGetLastError();
}

hipError_t launch_convT3x3s2_dgrad_mfma(const float* g, const float* w, float* gin, float* workspace, int64_t workspace_floats,
                                        int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    const int ncb = (Cin + CO - 1) / CO, nchunks = (Cout + KC - 1) / KC;
    const int64_t wtotal = (int64_t)ncb * nchunks * KK * CO;
    hipLaunchKernelGGL(convT_pack_weights, dim3(grid_1d(wtotal, 256)), dim3(256), 0, s, w, workspace, Cin, Cout, nchunks, ncb, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 tiles((W + TW - 1) / TW, (H + TH - 1) / TH);
    int ksplit = convT_ksplit((int64_t)tiles.x * tiles.y * N * ncb, nchunks);
    const int64_t out_elems = (int64_t)N * Cin * H * W;
    if (ksplit > 1 && workspace_floats < wtotal + (int64_t)ksplit * out_elems) ksplit = 1;
    if ((int64_t)N * ncb * ksplit > 65535) return hipErrorInvalidValue;
    float* slab = workspace + wtotal;
    const size_t lds_bytes = (size_t)(D_TILE + D_WT) * sizeof(float);
    auto k = convT3x3s2_dgrad_mfma;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tiles.x, tiles.y, (unsigned)(N * ncb * ksplit)), dim3(256), lds_bytes, s, g, workspace, gin, N, Cin, H, W,
                       Cout, nchunks, ncb, ksplit, slab);
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    hipLaunchKernelGGL(convT_sum_slabs, dim3(grid_1d(out_elems, 256)), dim3(256), 0, s, slab, gin, out_elems, ksplit);
    return hipGetLastError();
}

hipError_t launch_convT3x3s2_wgrad_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                        int H, int W, int Cout, hipStream_t s, int accumulate)
{
    const WgPlan p = convT_wgrad_plan(N, Cin, H, W, Cout);
    const int64_t slab_floats = (int64_t)p.ksplit * 9 * p.CoutP * p.CinP;
    float* bias_slab = gb ? workspace + slab_floats : nullptr;
    const int blocks = (p.CinP / WGT_CI) * (p.CoutP / WGT_CO);
    hipLaunchKernelGGL(convT3x3s2_wgrad_mfma, dim3((unsigned)(blocks * p.ksplit)), dim3(WG_T), 0, s, in, g, workspace, N, Cin, H, W, Cout,
                       p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int64_t rblocks = (int64_t)9 * p.CoutP * ((p.CinP + 63) / 64);
    if (rblocks > 256 * 64) rblocks = 256 * 64;
    const int bblocks = gb ? (p.CoutP + 63) / 64 : 0;
    const int groups = p.ksplit >= 64 ? 16 : (p.ksplit >= 24 ? 8 : 4);
    if (accumulate & 2) {          // deferred: one grouped launch at the end of the backward pass (conv_kernels.h)
        wgrad_defer(WgradReduceJob{workspace, gw, bias_slab, gb, Cin, Cout, p.CinP, p.CoutP, p.ksplit, p.ksplit * 16, (int)rblocks, bblocks,
                                   groups, 2, accumulate & 1, 0});
        return hipSuccess;
    }
    hipLaunchKernelGGL(convT_wgrad_reduce, dim3((unsigned)(rblocks + bblocks)), dim3(64 * groups), 0, s, workspace, gw, Cin, Cout, p.CinP,
                       p.CoutP, p.ksplit, bias_slab, gb, p.ksplit * 16, (int)rblocks, accumulate);
    return hipGetLastError();
}

}  // namespace sstem
