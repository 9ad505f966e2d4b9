// HBM-bound edge kernels of the hot path's callers:
//   * uint8 image <-> fp32 tensor conversions of the inference scripts
//       sff_scripts_interp/inference_singleImage.py:55-66,76  (x3 channel replicate, /255, *255 truncation)
//       sp_scripts_test/utils/gray2tensor.py:7-20
//   * Adam update over one flat parameter / gradient buffer (torch.optim.Adam semantics as used by
//       sff_scripts_interp/main_ms.py:315, sff_scripts_fusion/main_fusion.py, betas 0.9/0.999, eps 1e-8)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "misc_kernels.h"

namespace sstem {

// w0 a + w1 b with the rounding spelled out (one multiply, one fma): the three forward kernels below share it, so WHICH of them a plane's
// size selects does not change a bit of the result (left to the compiler's contraction they differed by one ulp on ~8 % of the outputs)
__device__ __forceinline__ float lerp2(float w0, float a, float w1, float b) { return __fmaf_rn(w1, b, __fmul_rn(w0, a)); }

// ---- bilinear x2 up-sampling, align_corners = True ------------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True): sff_scripts_interp/model/model_interp.py:17,
// sp_scripts_train/networks.py:27,213 -- five times in an IFNet trunk and once per kernel head, where it writes
// [B,51,H,W] planes (1.7 GB per head at C2).  PyTorch's formula (the arithmetic the reference delegates to):
//   src = dst * (in - 1) / (out - 1);  i0 = (int)src;  step = i0 < in - 1;  l1 = src - i0;  l0 = 1 - l1
//   out = l0y * (l0x * v[i0y][i0x] + l1x * v[i0y][i0x + stepx]) + l1y * (l0x * v[i0y + stepy][i0x] + l1x * v[..][i0x + stepx])
// One thread per four output pixels of a row (one 16-B store); the <= 4 source pixels they touch per source row are
// read once (clamped) and picked by index.  HBM-bound: the output is 4x the input.
__global__ __launch_bounds__(256) void upsample_bilinear2x_ac(const float* __restrict__ in, float* __restrict__ out,
                                                              int planes, int H, int W, float ry, float rx)
{
    // ry, rx = (in - 1) / (out - 1), divided on the HOST (IEEE, as torch computes its scale): a device division that is
    // one ulp off moves the source coordinate by 62 ulps at x = 62 -- measured as 1.1e-6 of the value range
    // grid: x = 256-thread chunks of one plane's (row, 4-pixel group) pairs, y = planes (grid-stride beyond 65535):
    // all index arithmetic is 32-bit (a first version with 64-bit div/mod per element ran below torch's kernel)
    const uint32_t OH = 2u * H, OW = 2u * W;
    const uint32_t q_per_row = OW / 4u;                 // OW = 2W is a multiple of 4 when W is even (launcher)
    const uint32_t per_plane = OH * q_per_row;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= per_plane) return;
    const uint32_t oy = e / q_per_row, q = e - oy * q_per_row;
    // source coordinates: ROUNDED products (torch rounds src = scale * dst before taking the fraction; letting the compiler
    // fuse the multiply into the subtraction below moves the weight by ~1 ulp of the coordinate, 4e-6 at x = 31)
    const float sy = __fmul_rn(ry, (float)oy);
    const int y0 = (int)sy;
    const int ystep = (y0 < H - 1) ? 1 : 0;
    const float l1y = sy - (float)y0, l0y = 1.f - l1y;
    const uint32_t ox = q * 4u;
    const int xlo = (int)__fmul_rn(rx, (float)ox);      // first source column of the four outputs
    int xs[4];
    float l1x[4];
    int d0[4], d1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xs[k] = (xlo + k < W) ? xlo + k : W - 1;
        const float sx = __fmul_rn(rx, (float)(ox + k));
        const int x0 = (int)sx;
        l1x[k] = sx - (float)x0;
        d0[k] = x0 - xlo;                               // 0..2: four outputs span at most two source pixels + one
        d1[k] = d0[k] + ((x0 < W - 1) ? 1 : 0);
    }
    // one-hot lane masks of the two source pixels of each output among the four loaded ones (d0, d1 in 0..3)
    uint32_t m0[4][4], m1[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { m0[k][j] = d0[k] == j ? 0xffffffffu : 0u; m1[k][j] = d1[k] == j ? 0xffffffffu : 0u; }
    auto pick4 = [](const float (&v)[4], const uint32_t (&m)[4]) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, v[0]) & m[0]) | (__builtin_bit_cast(uint32_t, v[1]) & m[1]) |
                                             (__builtin_bit_cast(uint32_t, v[2]) & m[2]) | (__builtin_bit_cast(uint32_t, v[3]) & m[3]));
    };
    // the next plane's eight source values are requested before this plane's outputs are formed and stored (a thread walks several
    // planes: with one exposed load latency per plane the 512 x 512 planes of the kernel heads ran behind aten's kernel)
    float na[4], nb[4];
    // the four source pixels of a row as ONE 16-byte load at a 4-byte aligned address (gfx950 global loads take it) wherever they are
    // all inside the row: eight dword loads per thread and plane were most of this kernel's instructions
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const bool xvec = xlo + 3 < W;
    auto request = [&](int pl) {
        const float* r0 = in + ((int64_t)pl * H + y0) * W;
        const float* r1 = r0 + (int64_t)ystep * W;
#if defined(SSTEM_UPS_ABLATE) && (SSTEM_UPS_ABLATE & 2)
        if (pl >= 0) {                     // developer ablation: no loads (wrong by design)
#pragma unroll
            for (int k = 0; k < 4; ++k) { na[k] = (float)(pl + k); nb[k] = (float)(pl - k); }
            return;
        }
#endif
        if (xvec) {
            const f4u va = *reinterpret_cast<const f4u*>(r0 + xlo), vb = *reinterpret_cast<const f4u*>(r1 + xlo);
#pragma unroll
            for (int k = 0; k < 4; ++k) { na[k] = va[k]; nb[k] = vb[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { na[k] = r0[xs[k]]; nb[k] = r1[xs[k]]; }
        }
    };
    if ((int)blockIdx.y < planes) request(blockIdx.y);
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = na[k]; b[k] = nb[k]; }
        if (pl + (int)gridDim.y < planes) request(pl + gridDim.y);
        // picks by per-thread bit masks (set up once, before the plane loop): written as nested selects the compiler turned the picks into
        // divergent branches around the loads -- with neither loads nor stores the kernel still took 0.54 of its 0.63 ms
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a0 = pick4(a, m0[k]), a1 = pick4(a, m1[k]);
            const float b0 = pick4(b, m0[k]), b1 = pick4(b, m1[k]);
            const float l0x = 1.f - l1x[k];
            o[k] = lerp2(l0y, lerp2(l0x, a0, l1x[k], a1), l1y, lerp2(l0x, b0, l1x[k], b1));
        }
#if defined(SSTEM_UPS_ABLATE) && (SSTEM_UPS_ABLATE & 1)
        if (o[0] == 12345.678f)            // developer ablation: no stores (wrong by design)
#endif
        *reinterpret_cast<float4*>(out + ((int64_t)pl * OH + oy) * OW + ox) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Tiled form for planes at least 128 outputs wide: a workgroup owns 4 output rows x 256 output columns of a plane and walks the planes.
// Its <= 4 source rows x <= 131 source columns go through LDS (indices clamped to the plane while staging, so "x0 + 1" and "y0 + 1" are
// always the next staged element), every thread owns ONE output column: the horizontal interpolation of a source row is formed once and
// used by the (up to three) output rows that read it, and which staged rows an output row reads is the same for the whole workgroup --
// no per-lane picks (the 4-outputs-per-thread kernel above spends its time on them: 0.54 of 0.60 ms with neither loads nor stores).
// Same expression per output as above.  (Tried: the next FOUR planes' windows in flight instead of one -- no change: not load latency.)
constexpr int UT_ROWS = 4, UT_COLS = 256, UT_SR = 4, UT_SW = 132;
__global__ __launch_bounds__(256) void upsample_bilinear2x_ac_tiled(const float* __restrict__ in, float* __restrict__ out, int planes, int H,
                                                                    int W, float ry, float rx, int tiles_x)
{
    __shared__ float win[UT_SR * UT_SW];
    const int OH = 2 * H, OW = 2 * W;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X0 = tx * UT_COLS, Y0 = ty * UT_ROWS;
    const int ox = X0 + threadIdx.x;
    const int xs_lo = (int)__fmul_rn(rx, (float)X0), ys_lo = (int)__fmul_rn(ry, (float)Y0);
    // this thread's column
    const float sx = __fmul_rn(rx, (float)(ox < OW ? ox : OW - 1));
    const int x0 = (int)sx;
    const float l1x = sx - (float)x0, l0x = 1.f - l1x;
    const int c0 = x0 - xs_lo;                                   // 0 .. 129; c0 + 1 is staged too (clamped to W - 1)
    // the four output rows (the same numbers in every thread)
    int r0[UT_ROWS];
    float l1y[UT_ROWS], l0y[UT_ROWS];
#pragma unroll
    for (int i = 0; i < UT_ROWS; ++i) {
        const int oy = Y0 + i < OH ? Y0 + i : OH - 1;
        const float sy = __fmul_rn(ry, (float)oy);
        const int y0 = (int)sy;
        l1y[i] = sy - (float)y0; l0y[i] = 1.f - l1y[i];
        r0[i] = y0 - ys_lo;                                      // 0 .. 2; row r0 + 1 is staged too (clamped to H - 1)
    }
    // staging: element e = t + 256 k of the UT_SR x UT_SW window, offsets inside a plane (clamped)
    constexpr int NE = (UT_SR * UT_SW + 255) / 256;
    int goff[NE], slot[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int r = e / UT_SW, c = e - r * UT_SW;
        int y = ys_lo + r, x = xs_lo + c;
        y = y < H ? y : H - 1; x = x < W ? x : W - 1;
        slot[k] = e < UT_SR * UT_SW ? e : -1;
        goff[k] = y * W + x;
    }
    float pre[NE];
    auto request = [&](int pl) {
        const float* ip = in + (int64_t)pl * H * W;
#pragma unroll
        for (int k = 0; k < NE; ++k) pre[k] = ip[goff[k]];
    };
    if ((int)blockIdx.y < planes) request(blockIdx.y);
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        __syncthreads();                                         // the previous plane's window is consumed
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (slot[k] >= 0) win[slot[k]] = pre[k];
        __syncthreads();
        if (pl + (int)gridDim.y < planes) request(pl + gridDim.y);
        float hl[UT_SR];
#pragma unroll
        for (int r = 0; r < UT_SR; ++r) {
            const float a0 = win[r * UT_SW + c0], a1 = win[r * UT_SW + c0 + 1];
            hl[r] = lerp2(l0x, a0, l1x, a1);
        }
        float* op = out + ((int64_t)pl * OH + Y0) * OW + ox;
#pragma unroll
        for (int i = 0; i < UT_ROWS; ++i) {
            const float h0 = r0[i] == 0 ? hl[0] : (r0[i] == 1 ? hl[1] : hl[2]);          // uniform conditions
            const float h1 = r0[i] == 0 ? hl[1] : (r0[i] == 1 ? hl[2] : hl[3]);
            if (ox < OW && Y0 + i < OH) op[(int64_t)i * OW] = lerp2(l0y[i], h0, l1y[i], h1);
        }
    }
}

// Wide form (round 4): every thread owns FOUR adjacent output columns of four output rows and stores them as 16-byte vectors -- a wave-store
// is 1 KB of one output row instead of 256 B (the one-column kernel above writes the 1.7 GB of a kernel head's up-sampling at 3.8 TB/s where
// torch's elementwise kernels reach 6.3 TB/s on the same box).  Workgroup = TXL x (256 / TXL) threads: 4 TXL columns x 4 (256 / TXL) rows; the
// source window goes through LDS as above.  Same expression per output as the kernels above: same bits.
template <int TXL>
__global__ __launch_bounds__(256) void upsample_bilinear2x_ac_wide(const float* __restrict__ in, float* __restrict__ out, int planes, int H,
                                                                   int W, float ry, float rx, int tiles_x)
{
    constexpr int TY = 256 / TXL, ROWS = 4 * TY, COLS = 4 * TXL;
    constexpr int SR = ROWS / 2 + 2, SW = COLS / 2 + 2, SWP = SW + 1;          // staged source rows / columns (+1: odd pitch)
    __shared__ float win[SR * SWP];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int OH = 2 * H, OW = 2 * W;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int X0 = bx * COLS, Y0 = by * ROWS;
    const int tx = threadIdx.x % TXL, ty = threadIdx.x / TXL;
    const int xs_lo = (int)__fmul_rn(rx, (float)X0), ys_lo = (int)__fmul_rn(ry, (float)Y0);
    int c0[4]; float l0x[4], l1x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = X0 + 4 * tx + k;
        const float sx = __fmul_rn(rx, (float)(ox < OW ? ox : OW - 1));
        const int x0 = (int)sx;
        l1x[k] = sx - (float)x0; l0x[k] = 1.f - l1x[k];
        c0[k] = x0 - xs_lo;
    }
    int r0[4]; float l0y[4], l1y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = Y0 + 4 * ty + i < OH ? Y0 + 4 * ty + i : OH - 1;
        const float sy = __fmul_rn(ry, (float)oy);
        const int y0 = (int)sy;
        l1y[i] = sy - (float)y0; l0y[i] = 1.f - l1y[i];
        r0[i] = y0 - ys_lo;                                        // row r0 + 1 is staged too (clamped to H - 1)
    }
    const int rb = r0[0];                                          // the thread's four output rows read staged rows rb .. rb + 3 (wave-uniform)
    constexpr int NE = (SR * SW + 255) / 256;
    int goff[NE], slot[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int r = e / SW, c = e - r * SW;
        int y = ys_lo + r, x = xs_lo + c;
        y = y < H ? y : H - 1; x = x < W ? x : W - 1;
        slot[k] = e < SR * SW ? r * SWP + c : -1;
        goff[k] = y * W + x;
    }
    float pre[NE];
    auto request = [&](int pl) {
        const float* ip = in + (int64_t)pl * H * W;
#pragma unroll
        for (int k = 0; k < NE; ++k) pre[k] = ip[goff[k]];
    };
    if ((int)blockIdx.y < planes) request(blockIdx.y);
    const bool colok = X0 + 4 * tx < OW;                           // OW % 4 == 0 (the launcher): a thread's four columns are in or out together
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (slot[k] >= 0) win[slot[k]] = pre[k];
        __syncthreads();
        if (pl + (int)gridDim.y < planes) request(pl + gridDim.y);
        float hl[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = rb + r < SR ? rb + r : SR - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a0 = win[rr * SWP + c0[k]], a1 = win[rr * SWP + c0[k] + 1];
                hl[r][k] = lerp2(l0x[k], a0, l1x[k], a1);
            }
        }
        float* op = out + ((int64_t)pl * OH + Y0 + 4 * ty) * OW + X0 + 4 * tx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = r0[i] - rb;                               // 0 .. 2, wave-uniform
            f4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float h0 = d == 0 ? hl[0][k] : (d == 1 ? hl[1][k] : hl[2][k]);
                const float h1 = d == 0 ? hl[1][k] : (d == 1 ? hl[2][k] : hl[3][k]);
                v[k] = lerp2(l0y[i], h0, l1y[i], h1);
            }
            if (colok && Y0 + 4 * ty + i < OH) *reinterpret_cast<f4*>(op + (int64_t)i * OW) = v;
        }
    }
}

hipError_t launch_upsample_bilinear2x(const float* in, float* out, int64_t planes, int H, int W, hipStream_t s)
{
    const uint32_t per_plane = (2u * H) * (2u * W / 4u);
    const unsigned gx = (per_plane + 255u) / 256u;
    // small planes: one grid row per plane; large planes: fewer rows that stride over the planes, so that the per-thread
    // coordinate arithmetic is shared by several planes
    int64_t gy = planes;
    if ((int64_t)gx * gy > 256 * 64 * 4) gy = (256 * 64 * 4 + gx - 1) / gx;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    const float ry = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float rx = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    static const bool tiled_off = [] { const char* e = getenv("SSTEM_UPSAMPLE_TILED"); return e && atoi(e) == 0; }();     // developer knob (A/B runs)
    const char* env_wide = getenv("SSTEM_UPSAMPLE_WIDE");                  // developer knob, read per launch (A/B runs, the bit-equality test)
    const int wide = env_wide ? atoi(env_wide) : 1;
    if (wide && !tiled_off && 2 * W >= 256 && (2 * W) % 4 == 0 && (int64_t)H * W < ((int64_t)1 << 29) && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int OW = 2 * W, OH = 2 * H;
        const int txl = OW >= 1024 ? 256 : (OW >= 512 ? 128 : 64);
        const int cols = 4 * txl, rows = 4 * (256 / txl);
        const int tiles_x = (OW + cols - 1) / cols, tiles_y = (OH + rows - 1) / rows;
        const unsigned tgx = (unsigned)(tiles_x * tiles_y);
        int64_t tgy = planes;
        if ((int64_t)tgx * tgy > 256 * 16) tgy = (256 * 16 + tgx - 1) / tgx;
        if (tgy > 65535) tgy = 65535;
        if (tgy < 1) tgy = 1;
        const dim3 grid(tgx, (unsigned)tgy);
        if (txl == 256) hipLaunchKernelGGL(upsample_bilinear2x_ac_wide<256>, grid, dim3(256), 0, s, in, out, (int)planes, H, W, ry, rx, tiles_x);
        else if (txl == 128) hipLaunchKernelGGL(upsample_bilinear2x_ac_wide<128>, grid, dim3(256), 0, s, in, out, (int)planes, H, W, ry, rx, tiles_x);
        else hipLaunchKernelGGL(upsample_bilinear2x_ac_wide<64>, grid, dim3(256), 0, s, in, out, (int)planes, H, W, ry, rx, tiles_x);
        return hipGetLastError();
    }
    if (!tiled_off && 2 * W >= 128 && (int64_t)H * W < ((int64_t)1 << 29)) {
        const int tiles_x = (2 * W + UT_COLS - 1) / UT_COLS, tiles_y = (2 * H + UT_ROWS - 1) / UT_ROWS;
        const unsigned tgx = (unsigned)(tiles_x * tiles_y);
        int64_t tgy = planes;
        if ((int64_t)tgx * tgy > 256 * 64) tgy = (256 * 64 + tgx - 1) / tgx;
        if (tgy > 65535) tgy = 65535;
        if (tgy < 1) tgy = 1;
        hipLaunchKernelGGL(upsample_bilinear2x_ac_tiled, dim3(tgx, (unsigned)tgy), dim3(256), 0, s, in, out, (int)planes, H, W, ry, rx, tiles_x);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(upsample_bilinear2x_ac, dim3(gx, (unsigned)gy), dim3(256), 0, s, in, out, (int)planes, H, W, ry, rx);
    return hipGetLastError();
}

// Gradient of upsample_bilinear2x_ac with respect to its input, as a GATHER (deterministic, no atomics; aten's kernel scatters
// with atomic adds and takes 80 us per call in a 256x256 training step).  Input row y receives from the output rows whose source
// row pair (y0, y0 + step) contains y: y0 in {y - 1, y}, i.e. oy in [2y - 2, 2y + 3]; the weights are
// recomputed with the forward kernel's own float arithmetic, so forward and backward are exact transposes of each other.
// Workgroup = one TW x TH input tile (256 pixels) of one plane: its (2 TW + 4) x (2 TH + 4) window of grad_output goes through LDS
// once.  64 x 4 tiles on planes at least 64 wide (whole 256-B / 528-B row segments: the 16 x 16 tile's 64-B stores and 144-B loads ran
// the [16,128,256,256] gradient of the SP U-Nets at 2.1 TB/s), 32 x 8 from 32, 16 x 16 below; same per-pixel arithmetic in all three.
template <int TW, int TH>
__global__ __launch_bounds__(256) void upsample_bilinear2x_ac_backward(const float* __restrict__ g, float* __restrict__ gin,
                                                                       int planes, int H, int W, float ry, float rx, int tiles_x)
{
    static_assert(TW * TH == 256, "one thread per input pixel of the tile");
    constexpr int WIN_W = 2 * TW + 4, WIN_H = 2 * TH + 4, UB_P = WIN_W + 1;
    __shared__ float win[WIN_H * UB_P];
    const int OH = 2 * H, OW = 2 * W;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X0 = tx * TW, Y0 = ty * TH;
    const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
    const int x = X0 + lx, y = Y0 + ly;
    // weights of the six candidate output rows / columns of this thread's input pixel (plane-independent)
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int oy = 2 * y - 2 + k;
        float w = 0.f;
        if (oy >= 0 && oy < OH) {
            const float sy = __fmul_rn(ry, (float)oy);
            const int y0 = (int)sy;
            const float l1 = sy - (float)y0, l0 = 1.f - l1;
            const int y1 = y0 + ((y0 < H - 1) ? 1 : 0);
            w = (y0 == y ? l0 : 0.f) + (y1 == y ? l1 : 0.f);
        }
        wy[k] = w;
        const int ox = 2 * x - 2 + k;
        float v = 0.f;
        if (ox >= 0 && ox < OW) {
            const float sx = __fmul_rn(rx, (float)ox);
            const int x0 = (int)sx;
            const float l1 = sx - (float)x0, l0 = 1.f - l1;
            const int x1 = x0 + ((x0 < W - 1) ? 1 : 0);
            v = (x0 == x ? l0 : 0.f) + (x1 == x ? l1 : 0.f);
        }
        wx[k] = v;
    }
    // the thread's window elements: offsets inside a plane (or -1) and LDS slots, plane-independent; the next plane's values are
    // requested before this plane's sums are formed (a workgroup walks several planes: one exposed load latency per plane was most
    // of the kernel's time)
    constexpr int NE = (WIN_H * WIN_W + 255) / 256;
    int goff[NE], slot[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int r = e / WIN_W, c = e - r * WIN_W;
        const int oy = 2 * Y0 - 2 + r, ox = 2 * X0 - 2 + c;
        slot[k] = e < WIN_H * WIN_W ? r * UB_P + c : -1;
        goff[k] = (e < WIN_H * WIN_W && oy >= 0 && oy < OH && ox >= 0 && ox < OW) ? oy * OW + ox : -1;
    }
    float pre[NE];
    auto request = [&](int pl) {
        const float* gp = g + (int64_t)pl * OH * OW;
#pragma unroll
        for (int k = 0; k < NE; ++k) pre[k] = goff[k] >= 0 ? gp[goff[k]] : 0.f;
    };
    if ((int)blockIdx.y < planes) request(blockIdx.y);
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        __syncthreads();                                        // the previous plane's window is consumed
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (slot[k] >= 0) win[slot[k]] = pre[k];
        __syncthreads();
        if (pl + (int)gridDim.y < planes) request(pl + gridDim.y);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float* row = win + (2 * ly + i) * UB_P + 2 * lx;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) s += wx[j] * row[j];
            acc += wy[i] * s;
        }
        if (y < H && x < W) gin[((int64_t)pl * H + y) * W + x] = acc;
    }
}

hipError_t launch_upsample_bilinear2x_backward(const float* g, float* gin, int64_t planes, int H, int W, hipStream_t s)
{
    const int TW = W >= 64 ? 64 : (W >= 32 ? 32 : 16), TH = 256 / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const unsigned gx = (unsigned)(tiles_x * tiles_y);
    int64_t gy = planes;
    if ((int64_t)gx * gy > 256 * 64) gy = (256 * 64 + gx - 1) / gx;       // larger grids stride over the planes (weights reused)
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    const float ry = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float rx = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    if (TW == 64)
        hipLaunchKernelGGL((upsample_bilinear2x_ac_backward<64, 4>), dim3(gx, (unsigned)gy), dim3(256), 0, s, g, gin, (int)planes, H, W, ry, rx, tiles_x);
    else if (TW == 32)
        hipLaunchKernelGGL((upsample_bilinear2x_ac_backward<32, 8>), dim3(gx, (unsigned)gy), dim3(256), 0, s, g, gin, (int)planes, H, W, ry, rx, tiles_x);
    else
        hipLaunchKernelGGL((upsample_bilinear2x_ac_backward<16, 16>), dim3(gx, (unsigned)gy), dim3(256), 0, s, g, gin, (int)planes, H, W, ry, rx, tiles_x);
    return hipGetLastError();
}

// ---- 2 x 2 / stride 2 pooling (nn.MaxPool2d(2), nn.AvgPool2d((2, 2), (2, 2)): model_unet.py:35-39, model_fusionnet.py:24, networks.py:33,134,
// model_interp.py:27) -- streaming kernels with torch's arithmetic: the average is ((a + b) + c) + d over the window in row-major order,
// times 0.25; the maximum is the FIRST largest element in row-major order (val > max || isnan(val)), its position 0..3 kept in one
// byte per output for the backward (aten keeps 64-bit flat indices and its backward for [16,32,256,256] took 132 us: 40 here).
// One thread per output pixel pair (two 16-byte... 8-byte row reads), OH = H / 2, OW = W / 2 (floor, as torch).
template <bool MAXP>
__global__ __launch_bounds__(256) void pool2x2_forward(const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ idx,
                                                       int64_t total, int H, int W, int OH, int OW)
{
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int ox = (int)(o % OW);
        const int64_t t = o / OW;
        const int oy = (int)(t % OH);
        const int64_t pl = t / OH;
        const float* r0 = in + (pl * H + 2 * oy) * W + 2 * ox;
        const float2 a = *reinterpret_cast<const float2*>(r0);            // W even or not: 2 * ox is even, rows start at pl*H*W + y*W
        const float2 b = *reinterpret_cast<const float2*>(r0 + W);
        if (MAXP) {
            float m = a.x; int k = 0;
            if (a.y > m || a.y != a.y) { m = a.y; k = 1; }
            if (b.x > m || b.x != b.x) { m = b.x; k = 2; }
            if (b.y > m || b.y != b.y) { m = b.y; k = 3; }
            out[o] = m;
            if (idx) idx[o] = (uint8_t)k;
        } else {
            out[o] = (((a.x + a.y) + b.x) + b.y) * 0.25f;
        }
    }
}

// generic (unaligned rows: W odd) form of the same
template <bool MAXP>
__global__ __launch_bounds__(256) void pool2x2_forward_scalar(const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ idx,
                                                              int64_t total, int H, int W, int OH, int OW)
{
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int ox = (int)(o % OW);
        const int64_t t = o / OW;
        const int oy = (int)(t % OH);
        const int64_t pl = t / OH;
        const float* r0 = in + (pl * H + 2 * oy) * W + 2 * ox;
        const float ax = r0[0], ay = r0[1], bx = r0[W], by = r0[W + 1];
        if (MAXP) {
            float m = ax; int k = 0;
            if (ay > m || ay != ay) { m = ay; k = 1; }
            if (bx > m || bx != bx) { m = bx; k = 2; }
            if (by > m || by != by) { m = by; k = 3; }
            out[o] = m;
            if (idx) idx[o] = (uint8_t)k;
        } else {
            out[o] = (((ax + ay) + bx) + by) * 0.25f;
        }
    }
}

// gradient: one thread per INPUT pixel pair of a row (both belong to one window); rows / columns beyond 2 * OH, 2 * OW get 0
template <bool MAXP>
__global__ __launch_bounds__(256) void pool2x2_backward(const float* __restrict__ g, const uint8_t* __restrict__ idx, float* __restrict__ gin,
                                                        int64_t total_in, int H, int W, int OH, int OW)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_in; e += (int64_t)gridDim.x * 256) {
        const int x = (int)(e % W);
        const int64_t t = e / W;
        const int y = (int)(t % H);
        const int64_t pl = t / H;
        const int oy = y >> 1, ox = x >> 1;
        float v = 0.f;
        if (oy < OH && ox < OW) {
            const int64_t o = (pl * OH + oy) * OW + ox;
            const float gv = g[o];
            if (MAXP) v = (idx[o] == (uint8_t)((y & 1) * 2 + (x & 1))) ? gv : 0.f;
            else v = gv * 0.25f;
        }
        gin[e] = v;
    }
}

__global__ __launch_bounds__(256) void gray_u8_to_f32(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                      int64_t npix, int replicas)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = __fdiv_rn((float)img[i], 255.0f);     // == numpy float32(k) / float32(255)
        for (int r = 0; r < replicas; ++r) out[(int64_t)r * npix + i] = v;
    }
}

__global__ __launch_bounds__(256) void f32_to_gray_u8(const float* __restrict__ pred, uint8_t* __restrict__ out,
                                                      int64_t npix, int clamp01)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        float p = pred[i];
        if (clamp01) p = p > 1.f ? 1.f : (p < 0.f ? 0.f : p);          // TrainTensor2mask, gray2tensor.py:26-31
        const float v = __fmul_rn(p, 255.0f);
        // numpy's float -> uint8 astype on x86-64: truncate toward zero to a wide integer, keep the low 8 bits
        // (no clamp: 256.0 -> 0, -1.0 -> 255); NaN and |v| >= 2^63 -> 0
        long long w = 0;
        if (v == v && fabsf(v) < 9.0e18f) w = (long long)v;
        out[i] = (uint8_t)(w & 0xFF);
    }
}

// p, g, m, v: flat fp32 buffers of n elements.  torch.optim.Adam (no amsgrad, L2 weight decay added to the
// gradient): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_step(float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                                 float bc1, float bc2_sqrt)
{
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);            // lerp form, as torch does
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

static inline int grid_1d(int64_t n)
{
    int64_t g = (n + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_gray_u8_to_f32(const uint8_t* img, float* out, int64_t npix, int replicas, hipStream_t s)
{
    hipLaunchKernelGGL(gray_u8_to_f32, dim3(grid_1d(npix)), dim3(256), 0, s, img, out, npix, replicas);
    return hipGetLastError();
}

hipError_t launch_f32_to_gray_u8(const float* pred, uint8_t* out, int64_t npix, int clamp01, hipStream_t s)
{
    hipLaunchKernelGGL(f32_to_gray_u8, dim3(grid_1d(npix)), dim3(256), 0, s, pred, out, npix, clamp01);
    return hipGetLastError();
}

hipError_t launch_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, hipStream_t s)
{
    hipLaunchKernelGGL(adam_step, dim3(grid_1d(n)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2_sqrt);
    return hipGetLastError();
}

// ---- L1 loss, forward and gradient in ONE launch --------------------------------------------------------------------------------
// loss = mean |pred - target| (nn.L1Loss / F.l1_loss, reduction 'mean': sff_scripts_fusion/main_fusion.py:252, sff_scripts_interp/
// main_ms.py:205, sp_scripts_train/main_fusion.py:237-249) and grad = d loss / d pred = sign(pred - target) / n (sign(0) = 0, torch's
// convention) in one pass over the two tensors: torch spends ~10 launches on the pair (sub, abs, a two-stage mean, and in the backward
// sign, an expanded division and a multiply) -- 40-50 us of a 3 ms step at 2 samples per GPU.  Deterministic: every workgroup leaves
// its partial sum in ws[workgroup], the LAST one to arrive (one atomic counter) adds the partials in index order and writes the mean;
// it also resets the counter, so the workspace (L1_WS_FLOATS floats, zero when first used) needs no launch of its own between calls
// or between replays of a captured graph.
constexpr int L1_MAX_WGS = 1024;
__global__ __launch_bounds__(256) void l1_mean_fwd_grad(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                         float inv_n, float* __restrict__ loss, float* __restrict__ grad, float* __restrict__ ws)
{
    __shared__ float red[4];
    __shared__ int last;
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (vec) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const f4 a = reinterpret_cast<const f4*>(pred)[i], b = reinterpret_cast<const f4*>(target)[i];
            f4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = a[e] - b[e];
                acc += fabsf(d);
                g[e] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : (d == 0.f ? 0.f : d));      // NaN stays NaN
            }
            reinterpret_cast<f4*>(grad)[i] = g;
        }
    }
    for (int64_t i = (vec ? n4 * 4 : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        acc += fabsf(d);
        grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : (d == 0.f ? 0.f : d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        const unsigned prev = atomicAdd(reinterpret_cast<unsigned*>(ws + L1_MAX_WGS), 1u);
        last = prev == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last workgroup: partials in index order, four per thread, then the fixed tree above
    float t = 0.f;
    for (int i = threadIdx.x * 4; i < (int)gridDim.x; i += 1024)
        for (int e = 0; e < 4 && i + e < (int)gridDim.x; ++e) t += __hip_atomic_load(ws + i + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
        *reinterpret_cast<unsigned*>(ws + L1_MAX_WGS) = 0u;
    }
}

hipError_t launch_l1_mean_fwd_grad(const float* pred, const float* target, int64_t n, float* loss, float* grad, float* ws, hipStream_t s)
{
    int64_t g = (n + 1023) / 1024;          // a thread handles one float4 per trip
    if (g > L1_MAX_WGS) g = L1_MAX_WGS;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(l1_mean_fwd_grad, dim3((unsigned)g), dim3(256), 0, s, pred, target, n, 1.0f / (float)n, loss, grad, ws);
    return hipGetLastError();
}

static inline unsigned pool_grid(int64_t n)
{
    int64_t g = (n + 255) / 256;
    if (g > 256 * 64) g = 256 * 64;
    if (g < 1) g = 1;
    return (unsigned)g;
}

hipError_t launch_pool2x2_forward(const float* in, float* out, uint8_t* idx, int64_t planes, int H, int W, int is_max, hipStream_t s)
{
    const int OH = H / 2, OW = W / 2;
    const int64_t total = planes * OH * OW;
    if (total <= 0) return hipSuccess;
    const bool vec = (W % 2 == 0) && (reinterpret_cast<uintptr_t>(in) & 7) == 0;
    if (is_max) {
        if (vec) hipLaunchKernelGGL(pool2x2_forward<true>, dim3(pool_grid(total)), dim3(256), 0, s, in, out, idx, total, H, W, OH, OW);
        else hipLaunchKernelGGL(pool2x2_forward_scalar<true>, dim3(pool_grid(total)), dim3(256), 0, s, in, out, idx, total, H, W, OH, OW);
    } else {
        if (vec) hipLaunchKernelGGL(pool2x2_forward<false>, dim3(pool_grid(total)), dim3(256), 0, s, in, out, idx, total, H, W, OH, OW);
        else hipLaunchKernelGGL(pool2x2_forward_scalar<false>, dim3(pool_grid(total)), dim3(256), 0, s, in, out, idx, total, H, W, OH, OW);
    }
    return hipGetLastError();
}

hipError_t launch_pool2x2_backward(const float* g, const uint8_t* idx, float* gin, int64_t planes, int H, int W, int is_max, hipStream_t s)
{
    const int OH = H / 2, OW = W / 2;
    const int64_t total_in = planes * H * W;
    if (total_in <= 0) return hipSuccess;
    if (is_max) hipLaunchKernelGGL(pool2x2_backward<true>, dim3(pool_grid(total_in)), dim3(256), 0, s, g, idx, gin, total_in, H, W, OH, OW);
    else hipLaunchKernelGGL(pool2x2_backward<false>, dim3(pool_grid(total_in)), dim3(256), 0, s, g, idx, gin, total_in, H, W, OH, OW);
    return hipGetLastError();
}

}  // namespace sstem
