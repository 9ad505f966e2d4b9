// Separable 51-tap local convolution for MI355X (gfx950 / CDNA4).
//
// Replaces the reference's three CUDA kernels
//   libs/sepconv/src/SeparableConvolution_kernel.cu:25-52   (updateOutput)
//   libs/sepconv/src/SeparableConvolution_kernel.cu:77-112  (updateGradVertical)
//   libs/sepconv/src/SeparableConvolution_kernel.cu:115-150 (updateGradHorizontal)
// with a different algorithm (see DESIGN.md "Kernels"):
//
//   The op is  out[c;p] = sum_fy V[fy;p] * T[c,fy;p],   T[c,fy;p] = sum_fx H[fx;p] * in[c, y+fy, x+fx]
//   for pixel p=(y,x).  For FOUR neighbouring pixels x_b..x_b+3 of one row the
//   inner sums are a small banded matrix product
//       T[(c,fy), j] = sum_{t=0..53} in[c, y+fy, x_b+t] * Hs[t, j],   Hs[t,j] = H[t-j; (y,x_b+j)] (0 off-band)
//   i.e. D(4 rows x 4 pixels) += A(4x1) * B(1x4) per t -- exactly one block of
//   v_mfma_f32_4x4x1_16b_f32, which runs 16 such blocks (64 pixels of a row) per
//   instruction in exact fp32 (each product is one fmaf).  51 of the 54 k-steps
//   and 51 of the 52 rows per channel are useful: 92.6 % of the issued MFMA flops.
//   The A operand comes from an LDS image of the input tile (XOR-swizzled so the
//   ds_read_b128 of 4 rows x 16 blocks is conflict-free), the B operand (the
//   per-pixel horizontal coefficients, skewed by the pixel's position in its
//   block) lives in 54 VGPRs for a whole pixel row, V is applied to the 4x4
//   accumulator tile in registers.
//
//   gradVertical reuses the same T tiles (gV[fy] = sum_c g[c] * T[c,fy]).
//   gradHorizontal uses the transposed formulation with a column-major LDS image
//       G[t, j] = sum_k in[c, y0+k, x_b+t] * Vs[k; (y, x_b+j)],  gH[fx;(y,x_b+j)] = sum_c g[c] * G_c[fx+j, j].
//
// Wave = 64 lanes everywhere.  fp32 in, fp32 accumulate.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sepconv_kernels.h"

namespace sstem {

constexpr int F = 51;          // filter taps (reference: FILTER_LENGTH, kernel.cu:9)
constexpr int KSTEPS = 54;     // 51 taps + 3 skew positions of a 4-pixel block
constexpr int PITCH = 128;     // dwords per LDS row of the row-major input image
constexpr int TILE_COLS = 116; // 64 pixels + 50 halo, rounded up to whole 16-B chunks

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Direct kernels: one lane per output element, any C / any filter length.  Used for shapes the
// MFMA kernels do not take and as the in-library cross-check (tests compare both to the oracle).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sepconv_fwd_direct(
    const float* __restrict__ in, const float* __restrict__ ver, const float* __restrict__ hor,
    float* __restrict__ out, int64_t B, int64_t C, int64_t H, int64_t W, int filt)
{
    const int64_t plane = H * W;
    const int64_t Hin = H + filt - 1, Win = W + filt - 1;
    const int64_t n = B * plane;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / plane;
        const int64_t yx = p - b * plane;
        const int64_t y = yx / W, x = yx - y * W;
        const float* vp = ver + b * filt * plane + yx;
        const float* hp = hor + b * filt * plane + yx;
        for (int64_t c = 0; c < C; ++c) {
            const float* ip = in + ((b * C + c) * Hin + y) * Win + x;
            float acc = 0.f;
            for (int fy = 0; fy < filt; ++fy) {
                float t = 0.f;
                for (int fx = 0; fx < filt; ++fx)
                    t = fmaf(ip[(int64_t)fy * Win + fx], hp[(int64_t)fx * plane], t);
                acc = fmaf(vp[(int64_t)fy * plane], t, acc);
            }
            out[(b * C + c) * plane + yx] = acc;
        }
    }
}

// one lane per (b, f, y, x) of gradVertical / gradHorizontal
template <bool VERTICAL>
__global__ __launch_bounds__(256) void sepconv_grad_direct(
    const float* __restrict__ g, const float* __restrict__ in, const float* __restrict__ coef,
    float* __restrict__ gout, int64_t B, int64_t C, int64_t H, int64_t W, int filt)
{
    const int64_t plane = H * W;
    const int64_t Hin = H + filt - 1, Win = W + filt - 1;
    const int64_t n = B * filt * plane;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / (filt * plane);
        const int64_t r = p - b * filt * plane;
        const int64_t f = r / plane;
        const int64_t yx = r - f * plane;
        const int64_t y = yx / W, x = yx - y * W;
        const float* cp = coef + b * filt * plane + yx;
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c) {
            const float gg = g[(b * C + c) * plane + yx];
            const float* ip = in + ((b * C + c) * Hin + y) * Win + x;
            float t = 0.f;
            if (VERTICAL) {  // f = fy, sum over fx with H
                for (int fx = 0; fx < filt; ++fx)
                    t = fmaf(ip[f * Win + fx], cp[(int64_t)fx * plane], t);
            } else {         // f = fx, sum over fy with V
                for (int fy = 0; fy < filt; ++fy)
                    t = fmaf(ip[(int64_t)fy * Win + f], cp[(int64_t)fy * plane], t);
            }
            acc = fmaf(gg, t, acc);
        }
        gout[p] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA kernels
// ---------------------------------------------------------------------------------------------
// Workgroup = WAVES waves, tile = 64 pixels wide x (WAVES*RPW) rows tall; wave w owns rows
// w, w+WAVES, ...  The input tile (CH channels x (TR+51) rows x 116 cols) sits in LDS.
//
// LDS image (row-major kernels): dword index of tile element (c, r, col)
//     (c*ROWS + r)*PITCH + ((((col>>2) ^ ((r&3)<<2)) << 2) | (col&3))
// A 16-B chunk q of row r is stored at chunk q ^ ((r&3)<<2): the ds_read_b128 a wave issues
// (16 blocks = chunks q0..q0+15, 4 consecutive rows) then touches 16 distinct chunks mod 16 in
// every 16-lane service group (groups {0-3,12-15,20-27}, ... => blocks {0,3,5,6} etc., whose
// low two bits are distinct) -- conflict-free.

struct TileArgs {
    int64_t B, C, H, W;     // output sizes
    int64_t tiles_x, tiles_y;
    int c0;                 // first channel of this launch's channel chunk
};

__device__ __forceinline__ void decode_block(const TileArgs& a, int64_t& b, int64_t& ty, int64_t& tx)
{
    // XCD-aware order: blocks i and i+8 share an XCD (round-robin dispatch), so give each XCD a
    // contiguous run of tiles -- neighbouring tiles share input halos through that XCD's L2.
    const int64_t nwg = gridDim.x;
    int64_t id = blockIdx.x;
    if ((nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);
    tx = id % a.tiles_x;
    const int64_t r = id / a.tiles_x;
    ty = r % a.tiles_y;
    b = r / a.tiles_y;
}

template <int CH, int THREADS, int ROWS>
__device__ __forceinline__ void load_tile_rowmajor(float* lds, const float* __restrict__ in,
                                                   int64_t b, int64_t C, int c0, int64_t Hin,
                                                   int64_t Win, int64_t y0, int64_t x0)
{
    // 128 threads span one row (116 live columns); THREADS/128 rows per pass.
    const int col = threadIdx.x & 127;
    const int rsub = threadIdx.x >> 7;
    constexpr int RSTEP = THREADS / 128;
    const bool col_ok = (col < TILE_COLS) && (x0 + col < Win);
    for (int cr = rsub; cr < CH * ROWS; cr += RSTEP) {
        const int c = cr / ROWS;
        const int r = cr - c * ROWS;
        if (col < TILE_COLS) {
            float v = 0.f;
            if (col_ok && (y0 + r < Hin))
                v = in[((b * C + (c0 + c)) * Hin + (y0 + r)) * Win + x0 + col];
            lds[(c * ROWS + r) * PITCH + ((((col >> 2) ^ ((r & 3) << 2)) << 2) | (col & 3))] = v;
        }
    }
}

// Forward (MODE 0) and gradVertical (MODE 1) share the T-tile pipeline.
//   MODE 0: out[c] = sum_fy V[fy] * T[c,fy]          (writes output [B,C,H,W])
//   MODE 1: gV[fy] = sum_c g[c] * T[c,fy]            (writes grad_vertical [B,51,H,W])
template <int MODE, int CH, int WAVES, int RPW>
__global__ __launch_bounds__(WAVES * 64) void sepconv_rowmajor_mfma(
    const float* __restrict__ in, const float* __restrict__ ver_or_g,
    const float* __restrict__ hor, float* __restrict__ out, TileArgs args)
{
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;          // +50 halo +1 pad row (fy = 51, coefficient 0)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W, C = args.C;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int64_t y0 = ty * TR, x0 = tx * 64;

    load_tile_rowmajor<CH, WAVES * 64, ROWS>(lds, in, b, C, args.c0, Hin, Win, y0, x0);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int blk = lane >> 2;   // 4-pixel block within the 64-pixel row
    const int sub = lane & 3;    // pixel within block (B/D column j) == A row i
    const int64_t x = x0 + lane;
    const bool xok = x < W;
    const int64_t xc = xok ? x : (W - 1);

    for (int rr = 0; rr < RPW; ++rr) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        if (y >= H) break;  // wave-uniform

        // ---- B operand: horizontal coefficients of my pixel, skewed by my position in the block
        const float* hp = hor + (b * F) * plane + y * W + xc;
        float hs[KSTEPS];
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) {
            const int fx = t - sub;
            const bool ok = xok && (fx >= 0) && (fx < F);
            const int fxc = fx < 0 ? 0 : (fx >= F ? F - 1 : fx);
            const float v = hp[(int64_t)fxc * plane];
            hs[t] = ok ? v : 0.f;
        }

        // per-pixel row data of the epilogue
        const float* vp = ver_or_g + (b * F) * plane + y * W + xc;   // MODE 0: vertical
        float gch[CH];
        float oacc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { oacc[c] = 0.f; gch[c] = 0.f; }
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                gch[c] = xok ? ver_or_g[((b * C + args.c0 + c) * H + y) * W + xc] : 0.f;
        }

        // ---- A operand addressing: lane (blk, i=sub) reads row (yl + 4*ft + i), chunk blk + tq
        const int r0 = yl + sub;
        const int swz = (r0 & 3) << 2;
        const float* arow = lds + r0 * PITCH;

        for (int ft = 0; ft < 13; ++ft) {
            f32x4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* abase = arow + ft * 4 * PITCH;
#pragma unroll
            for (int tq = 0; tq < 14; ++tq) {
                f32x4 a[CH];
                const int chunk = ((blk + tq) ^ swz) << 2;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[c] = *reinterpret_cast<const f32x4*>(abase + c * ROWS * PITCH + chunk);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tq * 4 + e;
                    if (t < KSTEPS) {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[c][e], hs[t], acc[c], 0, 0, 0);
                    }
                }
            }
            // ---- epilogue of this 4-row tile: lane holds T[c, fy=4ft+i ; my pixel] in acc[c][i]
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int fy = ft * 4 + i;
                    if (fy < F) {   // fy == 51 is the pad row: never used
                        const float vv = vp[(int64_t)fy * plane];
#pragma unroll
                        for (int c = 0; c < CH; ++c) oacc[c] = fmaf(vv, acc[c][i], oacc[c]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int fy = ft * 4 + i;
                    if (fy < F) {
                        float s = 0.f;
#pragma unroll
                        for (int c = 0; c < CH; ++c) s = fmaf(gch[c], acc[c][i], s);
                        if (xok) {
                            float* dst = out + ((b * F + fy) * H + y) * W + x;
                            if (args.c0 == 0) *dst = s; else *dst += s;
                        }
                    }
                }
            }
        }
        if (MODE == 0 && xok) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                out[((b * C + args.c0 + c) * H + y) * W + x] = oacc[c];
        }
    }
}

// ---- gradHorizontal: column-major LDS image ---------------------------------------------------
// dword index of tile element (c, col, r):  (c*TCOLS + col)*PITCH_T + r, PITCH_T = 4*odd so the
// ds_read_b128 of 64 consecutive columns (same 4-row chunk) is conflict-free.
constexpr int TCOLS = 120;   // 64 + 50 halo, + t-tiles reach col 4*13+3+63 = 118

template <int CH, int WAVES, int RPW>
__global__ __launch_bounds__(WAVES * 64) void sepconv_gradh_mfma(
    const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ ver,
    float* __restrict__ gh, TileArgs args)
{
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;                         // aligned 4-row chunks may start 3 rows early... (see k0)
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;        // multiple of 4 dwords, (PITCH_T/4) odd
    static_assert(((PITCH_T / 4) & 1) == 1 && PITCH_T >= ROWS, "pitch");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W, C = args.C;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int64_t y0 = ty * TR, x0 = tx * 64;

    // stage the tile transposed: thread -> column (coalesced global read along x)
    {
        const int col = threadIdx.x & 127;
        const int rsub = threadIdx.x >> 7;
        constexpr int RSTEP = (WAVES * 64) / 128;
        if (col < TCOLS) {
            const bool col_ok = x0 + col < Win;
            for (int cr = rsub; cr < CH * ROWS; cr += RSTEP) {
                const int c = cr / ROWS;
                const int r = cr - c * ROWS;
                float v = 0.f;
                if (col_ok && (y0 + r < Hin))
                    v = in[((b * C + (args.c0 + c)) * Hin + (y0 + r)) * Win + x0 + col];
                lds[(c * TCOLS + col) * PITCH_T + r] = v;
            }
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 3;
    const int64_t x = x0 + lane;
    const bool xok = x < W;
    const int64_t xc = xok ? x : (W - 1);

    for (int rr = 0; rr < RPW; ++rr) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        if (y >= H) break;

        // K runs over 4-row aligned chunks starting at row k0 = yl & ~3; vertical coefficient of
        // LDS row k0 + k is V[k - (yl&3)], zero outside [0,51).  56 k-steps.
        const int k0 = yl & ~3;
        const int sh = yl & 3;
        const float* vp = ver + (b * F) * plane + y * W + xc;
        float vs[56];
#pragma unroll
        for (int k = 0; k < 56; ++k) {
            const int fy = k - sh;
            const bool ok = xok && (fy >= 0) && (fy < F);
            const int fyc = fy < 0 ? 0 : (fy >= F ? F - 1 : fy);
            const float v = vp[(int64_t)fyc * plane];
            vs[k] = ok ? v : 0.f;
        }
        float gch[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            gch[c] = xok ? g[((b * C + args.c0 + c) * H + y) * W + xc] : 0.f;

        // A operand: lane (blk, i) <-> tile column lane + 4*tt, rows k0 + 4*kq .. +3
        const float* abase = lds + lane * PITCH_T + k0;
        for (int tt = 0; tt < 14; ++tt) {
            f32x4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* acol = abase + tt * 4 * PITCH_T;
#pragma unroll
            for (int kq = 0; kq < 14; ++kq) {
                f32x4 a[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[c] = *reinterpret_cast<const f32x4*>(acol + c * TCOLS * PITCH_T + kq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[c][e], vs[kq * 4 + e], acc[c], 0, 0, 0);
                }
            }
            // acc[c][i] = G_c[t = 4tt+i ; my pixel j=sub];  gH[fx = t - j]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int fx = tt * 4 + i - sub;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) s = fmaf(gch[c], acc[c][i], s);
                if (xok && fx >= 0 && fx < F) {
                    float* dst = gh + ((b * F + fx) * H + y) * W + x;
                    if (args.c0 == 0) *dst = s; else *dst += s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static inline int grid_1d(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    const int64_t cap = 256 * 32;   // 256 CUs x 8 blocks x 4: grid-stride the rest
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_fwd_direct(const float* in, const float* ver, const float* hor, float* out,
                             int64_t B, int64_t C, int64_t H, int64_t W, int filt, hipStream_t s)
{
    hipLaunchKernelGGL(sepconv_fwd_direct, dim3(grid_1d(B * H * W, 256)), dim3(256), 0, s,
                       in, ver, hor, out, B, C, H, W, filt);
    return hipGetLastError();
}

hipError_t launch_bwd_direct(const float* g, const float* in, const float* ver, const float* hor,
                             float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                             int filt, hipStream_t s)
{
    const int grid = grid_1d(B * filt * H * W, 256);
    hipLaunchKernelGGL(sepconv_grad_direct<true>, dim3(grid), dim3(256), 0, s,
                       g, in, hor, gv, B, C, H, W, filt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sepconv_grad_direct<false>, dim3(grid), dim3(256), 0, s,
                       g, in, ver, gh, B, C, H, W, filt);
    return hipGetLastError();
}

constexpr int MF_WAVES = 8;
constexpr int MF_RPW = 4;

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int MODE, int CH>
static hipError_t launch_rowmajor(const float* in, const float* vg, const float* hor, float* out,
                                  const TileArgs& a, hipStream_t s)
{
    constexpr int TR = MF_WAVES * MF_RPW;
    constexpr size_t lds_bytes = (size_t)CH * (TR + F) * PITCH * sizeof(float);
    auto k = sepconv_rowmajor_mfma<MODE, CH, MF_WAVES, MF_RPW>;
    hipError_t e = set_lds(k, lds_bytes);
    if (e != hipSuccess) return e;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(MF_WAVES * 64), lds_bytes, s, in, vg, hor, out, a);
    return hipGetLastError();
}

template <int CH>
static hipError_t launch_gradh(const float* in, const float* g, const float* ver, float* gh,
                               const TileArgs& a, hipStream_t s)
{
    constexpr int TR = MF_WAVES * MF_RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    constexpr size_t lds_bytes = (size_t)CH * TCOLS * PITCH_T * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_gradh_mfma<CH, MF_WAVES, MF_RPW>;
    hipError_t e = set_lds(k, lds_bytes);
    if (e != hipSuccess) return e;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(MF_WAVES * 64), lds_bytes, s, in, g, ver, gh, a);
    return hipGetLastError();
}

static TileArgs make_args(int64_t B, int64_t C, int64_t H, int64_t W)
{
    TileArgs a;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.tiles_x = (W + 63) / 64;
    a.tiles_y = (H + MF_WAVES * MF_RPW - 1) / (MF_WAVES * MF_RPW);
    a.c0 = 0;
    return a;
}

bool mfma_grid_ok(int64_t B, int64_t H, int64_t W)
{
    TileArgs a = make_args(B, 1, H, W);
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    return nwg > 0 && nwg <= 0x7fffffffLL;
}

hipError_t launch_fwd_mfma(const float* in, const float* ver, const float* hor, float* out,
                           int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s)
{
    TileArgs a = make_args(B, C, H, W);
    hipError_t e = hipSuccess;
    for (int64_t c0 = 0; c0 < C && e == hipSuccess; c0 += 3) {
        a.c0 = (int)c0;
        const int64_t ch = (C - c0) < 3 ? (C - c0) : 3;
        if (ch == 3) e = launch_rowmajor<0, 3>(in, ver, hor, out, a, s);
        else if (ch == 2) e = launch_rowmajor<0, 2>(in, ver, hor, out, a, s);
        else e = launch_rowmajor<0, 1>(in, ver, hor, out, a, s);
    }
    return e;
}

hipError_t launch_bwd_mfma(const float* g, const float* in, const float* ver, const float* hor,
                           float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                           hipStream_t s)
{
    // C <= 3 (checked by the caller): a single channel chunk, c0 == 0.
    TileArgs a = make_args(B, C, H, W);
    hipError_t e;
    if (C == 3) e = launch_rowmajor<1, 3>(in, g, hor, gv, a, s);
    else if (C == 2) e = launch_rowmajor<1, 2>(in, g, hor, gv, a, s);
    else e = launch_rowmajor<1, 1>(in, g, hor, gv, a, s);
    if (e != hipSuccess) return e;
    if (C == 3) e = launch_gradh<3>(in, g, ver, gh, a, s);
    else if (C == 2) e = launch_gradh<2>(in, g, ver, gh, a, s);
    else e = launch_gradh<1>(in, g, ver, gh, a, s);
    return e;
}

}  // namespace sstem
