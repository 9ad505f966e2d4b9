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
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "sepconv_kernels.h"

#ifndef SSTEM_ABLATE
#define SSTEM_ABLATE 0   // developer builds (trusted-gray kernel): 1 no H loads, 2 no V loads, 4 no tile staging, 8 no MFMAs / LDS reads
#endif
#ifndef SSTEM_HPF
#define SSTEM_HPF 11     // trusted-gray kernel, B-operand prefetch: next-row horizontal taps requested per MFMA group (11: all by group 4)
#endif
#ifndef SSTEM_BLK_SKEWLD
#define SSTEM_BLK_SKEWLD 0 // blocked coefficients: 1 = the B operand's taps are requested already skewed (load_taps_skewed_buf) instead of coalesced
                           // requests + the in-register skew.  Measured and left off: -0.5 % on the 64-row shape, +3 % on the 32-row default
                           // (profiles/r03/a_*): the ~150 VALU instructions per pixel row it removes were not what the kernel waits for
#endif
#ifndef SSTEM_GRAY_DMA
#define SSTEM_GRAY_DMA 1 // trusted-gray forward / fused-apply kernel: tile staging by LDS-DMA (0: through registers, the round-1 loader)
#endif
#ifndef SSTEM_RGB_RING
#define SSTEM_RGB_RING 2 // three-channel streaming kernel: A-operand register ring (2: one chunk = 12 MFMAs of LDS latency covered; 3: two)
#endif
#ifndef SSTEM_GRAY16_ABLATE
#define SSTEM_GRAY16_ABLATE 0 // 1: the timing-only ablation variants of sepconv_gray16_mfma (SSTEM_GRAY16_VAR) are compiled in
#endif
#ifndef SSTEM_COEF_AUX
#define SSTEM_COEF_AUX 0   // cache-policy bits of the coefficient loads of the trusted-gray kernel (gfx950: 1 sc0, 2 nt, 16 sc1)
#endif

namespace sstem {

constexpr int F = 51;          // filter taps (reference: FILTER_LENGTH, kernel.cu:9)
constexpr int KSTEPS = 54;     // 51 taps + 3 skew positions of a 4-pixel block
constexpr int TILE_COLS = 116; // 64 pixels + 50 halo, rounded up to whole 16-B chunks

// Row-major LDS image: element (row r, channel c, col) at dword (r*CH + c)*pitch(CH) + col.
// pitch is chosen so that the row stride CH*pitch, counted in 16-B chunks, is 4 or 12 (mod 16):
// the ds_read_b128 a wave issues (16 blocks = 16 consecutive chunks, x 4 consecutive rows) then hits
// 16 distinct chunk slots (mod 256 B) inside every 16-lane service group -- the groups hold blocks
// {0,3,5,6}, {1,2,4,7} (+8), whose pairwise differences are never 4, 8 or 12 -- conflict-free,
// and every address is ONE base register + an immediate offset.
__host__ __device__ constexpr int rm_pitch(int ch) { return ch == 2 ? 120 : 144; }
// Tall tiles (16 waves x 3 rows) only fit in 160 KB with the unpadded pitch: a few 2-way conflicts on
// the A reads (LDS is ~27 % busy in this kernel) in exchange for 1.5x more rows per staged halo.
__host__ __device__ constexpr int rm_pitch_tile(int ch, int waves, int rpw)
{
    return (waves * rpw > 36) ? TILE_COLS : rm_pitch(ch);
}
static_assert((1 * rm_pitch(1) / 4) % 16 == 4 && (2 * rm_pitch(2) / 4) % 16 == 12 &&
              (3 * rm_pitch(3) / 4) % 16 == 12 && rm_pitch(2) >= TILE_COLS, "LDS row stride");

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Direct kernels: one lane per output element, any C / any filter length.  Used for shapes the
// MFMA kernels do not take and as the in-library cross-check (tests compare both to the oracle).
// ---------------------------------------------------------------------------------------------
// BF: the coefficient tensors are bf16 (widened exactly: the upper half of an fp32); skip_if_gray: the device word of
// detect_identical_channels -- non-zero means the trusted-gray kernel launched next to this one owns the call.
template <bool BF> __device__ __forceinline__ float coef_at(const float* t, int64_t i)
{
    if constexpr (BF) return __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t*>(t)[i] << 16);
    else return t[i];
}
template <bool BF = false>
__global__ __launch_bounds__(256) void sepconv_fwd_direct(
    const float* __restrict__ in, const float* __restrict__ ver, const float* __restrict__ hor,
    float* __restrict__ out, int64_t B, int64_t C, int64_t H, int64_t W, int filt, const int* __restrict__ skip_if_gray = nullptr)
{
    if (skip_if_gray && *skip_if_gray != 0) return;
    const int64_t plane = H * W;
    const int64_t Hin = H + filt - 1, Win = W + filt - 1;
    const int64_t n = B * plane;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / plane;
        const int64_t yx = p - b * plane;
        const int64_t y = yx / W, x = yx - y * W;
        const int64_t cbase = b * filt * plane + yx;
        for (int64_t c = 0; c < C; ++c) {
            const float* ip = in + ((b * C + c) * Hin + y) * Win + x;
            float acc = 0.f;
            for (int fy = 0; fy < filt; ++fy) {
                float t = 0.f;
                for (int fx = 0; fx < filt; ++fx)
                    t = fmaf(ip[(int64_t)fy * Win + fx], coef_at<BF>(hor, cbase + (int64_t)fx * plane), t);
                acc = fmaf(coef_at<BF>(ver, cbase + (int64_t)fy * plane), t, acc);
            }
            out[(b * C + c) * plane + yx] = acc;
        }
    }
}

// one lane per (b, f, y, x) of gradVertical / gradHorizontal
template <bool VERTICAL, bool BF = false>
__global__ __launch_bounds__(256) void sepconv_grad_direct(
    const float* __restrict__ g, const float* __restrict__ in, const float* __restrict__ coef,
    float* __restrict__ gout, int64_t B, int64_t C, int64_t H, int64_t W, int filt, const int* __restrict__ skip_if_gray = nullptr)
{
    if (skip_if_gray && *skip_if_gray != 0) return;
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
        const int64_t cbase = b * filt * plane + yx;
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c) {
            const float gg = g[(b * C + c) * plane + yx];
            const float* ip = in + ((b * C + c) * Hin + y) * Win + x;
            float t = 0.f;
            if (VERTICAL) {  // f = fy, sum over fx with H
                for (int fx = 0; fx < filt; ++fx)
                    t = fmaf(ip[f * Win + fx], coef_at<BF>(coef, cbase + (int64_t)fx * plane), t);
            } else {         // f = fx, sum over fy with V
                for (int fy = 0; fy < filt; ++fy)
                    t = fmaf(ip[(int64_t)fy * Win + f], coef_at<BF>(coef, cbase + (int64_t)fy * plane), t);
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
struct TileArgs {
    int64_t B, C, H, W;     // output sizes
    int64_t tiles_x, tiles_y;
    int c0;                 // first channel of this launch's channel chunk
    int in_planes;          // trusted-gray fused apply: planes per image in the frame tensors (3 = replicated frames [B,3,H,W],
                            // 1 = the single-plane entry point, frames [B,1,H,W])
    int dbg;                // developer ablation flags (SSTEM_DEBUG_FLAGS): 1 skip tile staging,
                            // 2 skip H loads, 4 one row-tile only, 8 no identical-channel fast path, 16 force it.  0 in production.
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

// All global addressing below is "wave-uniform 64-bit base (SGPRs) + one 32-bit per-lane byte offset"
// so the loads/stores use the saddr form and no 64-bit per-lane pointers occupy VGPR pairs.
// The global address space is spelled out: a pointer that went through pin_uniform (or any integer round trip) would
// otherwise be a generic pointer and load through flat_load (64-bit per-lane addresses, out-of-order counters).
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ float ldg(const float* ubase, uint32_t lane_byte_off)
{
    return *reinterpret_cast<const gfloat*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off);
}
__device__ __forceinline__ gfloat* stg_ptr(float* ubase, uint32_t lane_byte_off)
{
    return reinterpret_cast<gfloat*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off);
}

// Both tile loaders return whether, for the elements THIS thread staged, channels 1.. are bit-identical to
// channel 0 (the grayscale-replicated case the kernels exploit, see sepconv_rowmajor_mfma).  Branch-free:
// every load is unconditional from a clamped (valid) address; elements outside the image are zeroed by select.
template <int CH, int THREADS, int ROWS, int P>
__device__ __forceinline__ bool load_tile_rowmajor(float* lds, const float* __restrict__ in,
                                                   int64_t b, int64_t C, int c0, int64_t Hin,
                                                   int64_t Win, int64_t y0, int64_t x0)
{
    // 128 threads span one row (116 live columns); THREADS/128 rows per pass.
    const int col = threadIdx.x & 127;
    const int rsub = threadIdx.x >> 7;
    constexpr int RSTEP = THREADS / 128;
    if (col >= TILE_COLS) return true;
    const bool col_ok = (x0 + col < Win);
    const int64_t xs = col_ok ? (x0 + col) : (Win - 1);
    constexpr int NPASS = (ROWS + RSTEP - 1) / RSTEP;
    unsigned diff = 0u;     // OR of the bit differences to channel 0
    float v0[NPASS];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const float* src = in + ((b * C + (c0 + c)) * Hin) * Win + xs;
        float* dst = lds + c * P + col;
        float v[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {          // every load of this channel in flight at once
            const int64_t yy = y0 + rsub + k * RSTEP;
            const bool ok = col_ok && (yy < Hin);
            const float t = src[(yy < Hin ? yy : Hin - 1) * Win];
            v[k] = ok ? t : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int r = rsub + k * RSTEP;
            if (r < ROWS) dst[r * CH * P] = v[k];
            if (c == 0) v0[k] = v[k];
            else diff |= __float_as_uint(v[k]) ^ __float_as_uint(v0[k]);
        }
    }
    return diff == 0u;
}

// Same image, but read from the UNPADDED tensor [B,C,H,W] with ReplicationPad2d(25) folded in
// (model_interp.py:46,90-91): padded element (yp, xp) = src(clamp(yp-25, 0, H-1), clamp(xp-25, 0, W-1)).
template <int CH, int THREADS, int ROWS, int P, int CT = CH>   // CH channels staged out of CT in the tensor
__device__ __forceinline__ bool load_tile_rowmajor_replicate(float* lds, const float* __restrict__ in,
                                                             int64_t b, int64_t H, int64_t W,
                                                             int64_t y0, int64_t x0)
{
    const int col = threadIdx.x & 127;
    const int rsub = threadIdx.x >> 7;
    constexpr int RSTEP = THREADS / 128;
    if (col >= TILE_COLS) return true;
    const int Hi = (int)H, Wi = (int)W;                 // H*W < 2^31 (checked by the C-ABI)
    int xs = (int)x0 + col - (F / 2);
    xs = xs < 0 ? 0 : (xs > Wi - 1 ? Wi - 1 : xs);
    constexpr int NPASS = (ROWS + RSTEP - 1) / RSTEP;
    // per-pass in-plane offsets are the same for every channel: compute them once (32-bit)
    uint32_t off[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        int ys = (int)y0 + rsub + k * RSTEP - (F / 2);
        ys = ys < 0 ? 0 : (ys > Hi - 1 ? Hi - 1 : ys);
        off[k] = ((uint32_t)ys * (uint32_t)Wi + (uint32_t)xs) * 4u;
    }
    unsigned diff = 0u;
    float v0[NPASS];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const float* src = in + (b * CT + c) * H * W;     // uniform
        float* dst = lds + c * P + col;
        float v[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) v[k] = ldg(src, off[k]);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int r = rsub + k * RSTEP;
            if (r < ROWS) dst[r * CH * P] = v[k];
            if (c == 0) v0[k] = v[k];
            else diff |= __float_as_uint(v[k]) ^ __float_as_uint(v0[k]);
        }
    }
    return diff == 0u;
}

// Coefficient vector of one pixel, skewed by `shift` (0..3) positions: dst[t] = coef[t - shift]
// (0 outside [0,51)), coef[f] = row_base[f*plane + x].  `row_base` is wave-uniform and points at
// (b, tap 0, y, x0); the load of entry t uses the uniform pointer of tap (t - 3) plus the per-lane
// byte offset ((3 - shift)*plane)*4 + xoff.  Lanes whose tap falls outside [0,51) (only possible
// for t < 3 and t > 50) read the nearest valid tap instead and the value is discarded, so every
// load is unconditional (no exec-mask branches).  N = 54 (row-major kernels) or 56 (gradH).
// Keeps a wave-uniform pointer in an SGPR pair and opaque to the optimiser: the loads that use it take the
// "saddr + 32-bit lane offset" form and a running pointer stays a running pointer (two SALU adds per step) instead of
// being re-associated into per-lane 64-bit address arithmetic.
template <typename T>
__device__ __forceinline__ void pin_uniform(T*& p) { asm volatile("" : "+s"(p)); }

template <int N>
__device__ __forceinline__ void load_skewed(float (&dst)[N], const float* row_base, int64_t plane,
                                            uint32_t xoff, int shift, bool ok, const int t0 = 0, const int t1 = N)
{
    // entries t0..t1-1 only (constants after unrolling: the trusted-gray kernel spreads a row's requests over
    // its MFMA groups)
    const uint32_t plane4 = (uint32_t)plane * 4u;
    const uint32_t skew_off = (uint32_t)(3 - shift) * plane4 + xoff;
    const float* ub = row_base + (int64_t)(t0 - 3) * plane;      // uniform: tap (t - 3) of the row
    pin_uniform(ub);
#pragma unroll
    for (int t = 0; t < N; ++t) {
        if (t < t0 || t >= t1) continue;
#if SSTEM_ABLATE & 1
        dst[t] = 0.25f; continue;
#endif
        if (t >= F + 3) { dst[t] = 0.f; continue; }              // t - shift >= 51 for every shift
        if (t >= 3 && t < F) {
            const float v = ldg(ub, skew_off);
            dst[t] = ok ? v : 0.f;
        } else {
            int sh2 = shift;
            if (t < 3) sh2 = shift < t ? shift : t;
            if (t >= F) sh2 = shift > (t - F + 1) ? shift : (t - F + 1);
            const float v = ldg(ub, (uint32_t)(3 - sh2) * plane4 + xoff);
            dst[t] = (ok && sh2 == shift) ? v : 0.f;
        }
        ub += plane;
        pin_uniform(ub);
    }
}

// second operand set of the fused interpolation apply (MODE 2); unused by the other modes
struct FusedArgs {
    const float* in2;
    const float* ver2;
    const float* hor2;
    const int* gray_flag;   // device word written by detect_identical_channels (nullptr: no device-side dispatch)
    uint8_t* out_u8;        // fused apply on planes (trusted-gray kernel), nullable: the result ALSO as (v * 255).astype(uint8) -- fp32
                            // multiply, truncation toward zero, low 8 bits, NO clamp (sff_scripts_interp/inference_singleImage.py:76) --
                            // [B,H,W] bytes, stored by the launch that holds the value
};

// (pred * 255).astype(np.uint8) as numpy does it on x86-64 (misc_kernels.hip, f32_to_gray_u8): truncate to a wide integer, low 8 bits;
// NaN and |v| >= 2^63 give 0
__device__ __forceinline__ uint8_t numpy_u8_of(float p)
{
    const float v = __fmul_rn(p, 255.0f);
    long long w = 0;
    if (v == v && fabsf(v) < 9.0e18f) w = (long long)v;
    return (uint8_t)(w & 0xFF);
}

// Forward (MODE 0), gradVertical (MODE 1) and the fused interpolation apply (MODE 2) share the T-tile
// pipeline.
//   MODE 0: out[c] = sum_fy V[fy] * T[c,fy]          (writes output [B,C,H,W])
//   MODE 1: gV[fy] = sum_c g[c] * T[c,fy]            (writes grad_vertical [B,51,H,W])
//   MODE 2: out = mean_c( sepconv(pad(in), ver, hor)[c] + sepconv(pad(in2), ver2, hor2)[c] )   [B,1,H,W]
//           = model_interp.py:90-97 in one launch: inputs are the UNPADDED images (replication padding
//           is folded into the tile staging), the two tiles are staged one after the other into the same
//           LDS, per-channel sums stay in registers, only the channel mean is written.
//
// Grayscale frames replicated x3 are handled twice: per tile by the vote below (any input, exact), and for whole
// calls by the dedicated kernel sepconv_gray_mfma further down, selected on the device by a flag
// (detect_identical_channels): this kernel reads the flag first and returns at once when the gray kernel owns the call.
template <int MODE, int CH, int WAVES, int RPW>
__global__ __launch_bounds__(WAVES * 64, 1) void sepconv_rowmajor_mfma(
    const float* __restrict__ in_a, const float* __restrict__ ver_or_g_a,
    const float* __restrict__ hor_a, float* __restrict__ out, TileArgs args, FusedArgs fa)
{
    if (fa.gray_flag) {                    // device-side dispatch between the two builds (uniform scalar load)
        const int f = *fa.gray_flag;
        if (f != 0) return;
    }
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;          // +50 halo +1 pad row (fy = 51, coefficient 0)
    constexpr int CHL = CH;               // channels staged in LDS
    constexpr int P = rm_pitch_tile(CHL, WAVES, RPW);   // dwords between channels of one row
    constexpr int RS = CHL * P;           // dwords between rows
    constexpr int RING = (WAVES >= 16) ? 2 : 3;   // A-operand register ring (see below)
    constexpr int VQD = (WAVES >= 16) ? 1 : 3;    // vertical-coefficient queue depth (16 waves: deeper queues measured no faster)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W, C = args.C;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int64_t y0 = ty * TR, x0 = tx * 64;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar for the compiler
    const int blk = lane >> 2;   // 4-pixel block within the 64-pixel row
    const int sub = lane & 3;    // pixel within block (B/D column j) == A row i
    const int64_t x = x0 + lane;
    const bool xok = x < W;
    const bool ld_ok = xok && !(args.dbg & 2);
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;                       // byte offset in a row

    constexpr int NPH = (MODE == 2) ? 2 : 1;

#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
    const float* in = (MODE == 2 && ph) ? fa.in2 : in_a;
    const float* ver_or_g = (MODE == 2 && ph) ? fa.ver2 : ver_or_g_a;
    const float* hor = (MODE == 2 && ph) ? fa.hor2 : hor_a;
    const float* hor_b = hor + (b * F) * plane + x0;                             // uniform bases
    const float* vg_b = ver_or_g + (MODE == 1 ? (b * C + args.c0) * plane : (b * F) * plane) + x0;

    // B operand of my first row: issued before the tile staging so both are in flight together
    // (MODE 2 has no registers to spare at 4 waves/SIMD: it loads them after the staging instead).
    float hs[KSTEPS];
    const int64_t yf = (y0 + wave < H) ? (y0 + wave) : (H - 1);
    if (MODE != 2) load_skewed<KSTEPS>(hs, hor_b + yf * W, plane, xoff, sub, ld_ok);

    if (MODE == 2 && ph) __syncthreads();      // every wave is done reading the first image's tile
    bool same = false;
    if (!(args.dbg & 1) && !(SSTEM_ABLATE & 4)) {
        if (MODE == 2) {
            // the clamped per-thread offsets do not depend on the phase: keep the compiler from hoisting them
            // out of the phase loop (they would stay live across both images' MFMA loops and spill)
            int zero = 0;
            asm volatile("" : "+s"(zero));
            same = load_tile_rowmajor_replicate<CHL, WAVES * 64, ROWS, P, CH>(lds, in, b, H, W, y0 + zero, x0);
        }
        else same = load_tile_rowmajor<CHL, WAVES * 64, ROWS, P>(lds, in, b, C, args.c0, Hin, Win, y0, x0);
    }
    if (MODE == 2) load_skewed<KSTEPS>(hs, hor_b + yf * W, plane, xoff, sub, ld_ok);
    // Barrier + vote: `gray` is true iff the three channel tiles are bit-identical (what every caller of the
    // reference feeds: one grayscale frame replicated x3, inference_singleImage.py:55-61, test_fusion.py:105-106,
    // sp main_fusion.py:210-211).  Then T[c,fy] is the same for every c and is computed ONCE; the results are
    // bit-identical to the generic path because the per-channel arithmetic and its order are unchanged.
    // Workgroup-uniform, exact, no hint from the caller.  SSTEM_DEBUG_FLAGS: 8 disables it, 16 forces it (A/B runs).
    const bool gray = (__syncthreads_and((MODE != 1 && CH == 3 && !(args.dbg & 8)) ? (int)(same || (args.dbg & 16)) : 0) != 0);

#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        if (y >= H) break;  // wave-uniform

        // ---- vertical coefficients (MODE 0): a 3-deep queue of 4-row groups, issued BEFORE the
        // next-row prefetch below (vmcnt retires in order, so the first group must not queue behind it)
        const float* vp = vg_b + y * W;   // uniform: MODE 0 V tap 0 of this row; MODE 1 grad_out chan c0
        float vq[VQD + 1][4];
        if (MODE != 1 && !(CH == 3 && gray)) {
#pragma unroll
            for (int q = 0; q < VQD; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    vq[q][i] = ldg(vp + (int64_t)(q * 4 + i) * plane, xoff);
        }

        // ---- prefetch the B operand of my NEXT row (lands while this row computes).  Only in the
        // 2-waves-per-SIMD shapes (256-register budget); the 3-waves-per-SIMD shapes reload at the
        // row end and rely on the other two waves of the SIMD to cover the latency.
        constexpr bool PF = (WAVES <= 8);
        float hn[PF ? KSTEPS : 1];
        const bool more = (rr + 1 < RPW) && (y + WAVES < H);
        if constexpr (PF) {
            // Unconditional (a branch here makes the compiler drain vmcnt at the join, turning the prefetch into a
            // blocking load).  On a wave's last row every "tap" re-reads tap 0 of the current row (plane stride 0):
            // one hot 256-B segment instead of a second pass over 51 planes.
            const int64_t yn = more ? (y + WAVES) : y;
            load_skewed<KSTEPS>(hn, hor_b + yn * W, more ? plane : 0, xoff, sub, ld_ok && more);
        }

        float gch[CH];
        float oacc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { oacc[c] = 0.f; gch[c] = 0.f; }
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                gch[c] = xok ? ldg(vp + (int64_t)c * plane, xoff) : 0.f;
        }

        // ---- A operand addressing: lane (blk, i=sub) reads row (yl + 4*ft + i), chunk blk + tq
        const float* arow = lds + (yl + sub) * RS + blk * 4;

        if (MODE != 1 && CH == 3 && gray) {
            // ===== identical channels: the concurrent accumulator chains are NG consecutive 4-row tiles of
            // channel 0 (6 groups of 2 tiles, then tile 12 on its own).  Same MFMA sequence per tile and the same
            // fy-ascending V accumulation as the generic path => bit-identical T and out, one third of the MFMAs.
            constexpr int NG = 2;
            float o = 0.f;
            f32x4 ar[RING][NG];
#pragma unroll
            for (int q = 0; q < RING - 1; ++q)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    ar[q][g] = *reinterpret_cast<const f32x4*>(arow + g * 4 * RS + q * 4);
            // one group: tiles 2fg, 2fg+1 with taps vc[0..7]; requests the taps of the NEXT group (fg == 5: tile 12) into vl
            auto group = [&](int fg, float (&vc)[8], float (&vl)[8]) __attribute__((always_inline)) {
                f32x4 acc[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* abase = arow + fg * (NG * 4) * RS;
                const float* anext = arow + (fg + 1) * (NG * 4) * RS;    // fg == 5: tile 12 (chain 0 only)
                const int gstep = (fg == 5) ? 0 : 4 * RS;               // keep chain 1 inside the image then
                {
                    const float* vt = vp + (int64_t)((fg + 1) * 8) * plane;
#pragma unroll
                    for (int i = 0; i < 8; ++i)                          // fg == 5: taps 48..50 only (clamped)
#if SSTEM_ABLATE & 2
                        vl[i] = 0.5f;
#else
                        vl[i] = ldg(vt + (int64_t)((fg == 5 && i > 2) ? 2 : i) * plane, xoff);
#endif
                }
#pragma unroll
                for (int tq = 0; tq < 14; ++tq) {
                    {
                        constexpr int D = RING - 1;
                        if (tq + D < 14) {
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                ar[(tq + D) % RING][g] = *reinterpret_cast<const f32x4*>(abase + g * 4 * RS + (tq + D) * 4);
                        } else {
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                ar[(tq + D) % RING][g] = *reinterpret_cast<const f32x4*>(anext + g * gstep + (tq + D - 14) * 4);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = tq * 4 + e;
                        if (t < KSTEPS) {
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[tq % RING][g][e], hs[t], acc[g], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (RING == 3) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) { ar[1][g] = ar[0][g]; ar[0][g] = ar[2][g]; }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o = fmaf(vc[g * 4 + i], acc[g][i], o);   // fy = 8fg + 4g + i, ascending
            };
            // two tap buffers used alternately: a loaded register is never copied (a v_mov of a value still in flight
            // would make the wave wait for the load it has just issued)
            float va[8], vbuf[8];
#pragma unroll
#if SSTEM_ABLATE & 2
            for (int i = 0; i < 8; ++i) va[i] = 0.5f;
#else
            for (int i = 0; i < 8; ++i) va[i] = ldg(vp + (int64_t)i * plane, xoff);
#endif
#pragma unroll 1
            for (int fg = 0; fg < 6; fg += 2) { group(fg, va, vbuf); group(fg + 1, vbuf, va); }
            {   // tile 12: rows fy = 48, 49, 50 (+ the pad row); its taps were requested by group 5 into va[0..2]
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* abase = arow + 48 * RS;
#pragma unroll
                for (int tq = 0; tq < 14; ++tq) {
                    constexpr int D = RING - 1;
                    if (tq + D < 14) ar[(tq + D) % RING][0] = *reinterpret_cast<const f32x4*>(abase + (tq + D) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = tq * 4 + e;
                        if (t < KSTEPS) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[tq % RING][0][e], hs[t], acc0, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) o = fmaf(va[i], acc0[i], o);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) oacc[c] = o;
        } else {

        // A operand ring: chunk g lives in ar[g % RING]; chunk g+RING-1 is requested before chunk g's
        // MFMAs (RING = 3 covers two chunks = 24 MFMAs of LDS latency; the 16-wave shape has a
        // 128-register budget and uses RING = 2).
        f32x4 ar[RING][CH];
#pragma unroll
        for (int q = 0; q < RING - 1; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                ar[q][c] = *reinterpret_cast<const f32x4*>(arow + c * P + q * 4);

        const int nft = (args.dbg & 4) ? 1 : 13;
#pragma unroll 1
        for (int ft = 0; ft < nft; ++ft) {
            f32x4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* abase = arow + ft * 4 * RS;
            // next tile's first chunks wrap to tile 0 after the last tile (valid address, unused)
            const float* anext = arow + ((ft == 12) ? 0 : (ft + 1) * 4 * RS);
            if (MODE != 1) {   // vertical coefficients VQD 4-row tiles ahead (clamped to the last tile,
                               // whose 4th row is the pad row: it re-reads tap 50 and is never used)
                const int ftn = (ft + VQD < 12) ? (ft + VQD) : 12;
                const float* vt = vp + (int64_t)(ftn * 4) * plane;          // uniform
                vq[VQD][0] = ldg(vt, xoff);
                vq[VQD][1] = ldg(vt + plane, xoff);
                vq[VQD][2] = ldg(vt + 2 * plane, xoff);
                vq[VQD][3] = ldg(vt + ((ftn == 12) ? 2 : 3) * plane, xoff);
            }
#pragma unroll
            for (int tq = 0; tq < 14; ++tq) {
                {
                    constexpr int D = RING - 1;
                    const float* src = (tq + D < 14) ? (abase + (tq + D) * 4) : (anext + (tq + D - 14) * 4);
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        ar[(tq + D) % RING][c] = *reinterpret_cast<const f32x4*>(src + c * P);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tq * 4 + e;
                    if (t < KSTEPS) {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[tq % RING][c][e], hs[t], acc[c], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (RING == 3) {   // 14 % 3 == 2: the next tile's chunks 0 and 1 sit in ar[2], ar[0]
#pragma unroll
                for (int c = 0; c < CH; ++c) { ar[1][c] = ar[0][c]; ar[0][c] = ar[2][c]; }
            }                            // RING == 2: 14 % 2 == 0, already in place
            // ---- epilogue of this 4-row tile: lane holds T[c, fy=4ft+i ; my pixel] in acc[c][i]
            if (MODE != 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int fy = ft * 4 + i;
                    if (fy < F) {   // fy == 51 is the pad row: never used
#pragma unroll
                        for (int c = 0; c < CH; ++c) oacc[c] = fmaf(vq[0][i], acc[c][i], oacc[c]);
                    }
#pragma unroll
                    for (int q = 0; q < VQD; ++q) vq[q][i] = vq[q + 1][i];
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
                            gfloat* dst = stg_ptr(out + ((b * F + fy) * H + y) * W + x0, xoff);
                            if (args.c0 == 0) *dst = s; else *dst += s;
                        }
                    }
                }
            }
        }
        }   // generic / gray
        if (MODE == 0 && xok) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                *stg_ptr(out + ((b * C + args.c0 + c) * H + y) * W + x0, xoff) = oacc[c];
        }
        if (MODE == 2) {   // channel sum of this image's result for this row (mean = sum/C is linear).  The first
            // phase parks it in the output element itself (same lane writes and re-reads it: no registers held
            // across the second image's whole MFMA loop, no extra buffer); the second phase finishes the mean.
            float csum = oacc[0];
#pragma unroll
            for (int c = 1; c < CH; ++c) csum += oacc[c];
            if (xok) {
                gfloat* dst = stg_ptr(out + (b * H + y) * W + x0, xoff);
                if (ph == 0) *dst = csum;
                else *dst = (*dst + csum) * (1.0f / CH);
            }
        }
        if constexpr (PF) {
#pragma unroll
            for (int t = 0; t < KSTEPS; ++t) hs[t] = hn[t];
        } else {
            if (more) load_skewed<KSTEPS>(hs, hor_b + (y + WAVES) * W, plane, xoff, sub, ld_ok);
        }
    }
    }   // phases

}

// ---- buffer addressing for the coefficient streams of the trusted-gray kernel --------------------
// One buffer resource per (tensor, image): base = the image's 51 planes, 51*plane*4 bytes (< 4 GiB, checked by
// the launcher).  A load is  base + soffset (SGPR: row and tap, a running offset advanced by ONE scalar add per
// tap) + voffset (VGPR: lane, plus the lane's skew in whole planes): no per-lane 64-bit addresses, no VALU
// address arithmetic next to the MFMAs.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t coef_rsrc(const float* image_planes, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image_planes), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, SSTEM_COEF_AUX));
}
// bf16 coefficient tensors (round 5: the ..._bf16coef entry points of include/sstem_sepconv.h -- BASELINE config 5, "bf16 activations
// with fp32 sepconv accumulate"; SURVEY 8(b), 8(d): the two 51 H W terms of the byte model halved).  The kernels keep ALL their
// offset arithmetic in fp32 bytes; a bf16 tensor's element sits at half that offset, is requested as 16 bits and widened to fp32 by
// one shift (bf16 IS the upper half of an fp32): the same products, the same fp32 sums as on an fp32 tensor that holds the rounded
// values.
template <bool BF>
__device__ __forceinline__ float bldc(rsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (BF) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)(voff >> 1), (int)(soff >> 1), SSTEM_COEF_AUX);
        return __builtin_bit_cast(float, v << 16);
    } else {
        return bld(r, voff, soff);
    }
}
// resource over one image's coefficients: `tensor` + elem_off ELEMENTS (fp32 or bf16), fp32_bytes = the image's size as an fp32 tensor
template <bool BF>
__device__ __forceinline__ rsrc_t coef_rsrc_c(const float* tensor, int64_t elem_off, uint32_t fp32_bytes)
{
    const char* base = reinterpret_cast<const char*>(tensor) + elem_off * (BF ? 2 : 4);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(BF ? fp32_bytes >> 1 : fp32_bytes), 0x00020000);
}
// keeps a running scalar offset a running offset (the optimiser would otherwise precompute one SGPR per tap)
__device__ __forceinline__ void pin_s(uint32_t& v) { asm volatile("" : "+s"(v)); }

// Horizontal taps of the lane's own pixel, coalesced: dst[f] = H[f] for f in [t0, t1) (constants after unrolling;
// f < 51).  Every lane of the instruction reads the same tap plane (two 128-B lines per wave-instruction, like
// the vertical taps).  rowoff = byte offset of (tap 0, row, x0); pstride = plane bytes, or 0 to re-read one hot
// segment (results unused).
// Measured (profiles/r01/q_ablation_gray_stream.txt): requesting the taps already skewed (lane j reads tap t-j:
// four planes = eight lines per instruction, every segment asked for by four consecutive instructions) costs the
// coefficient stream 6-12 % of its rate -- the per-CU L1 has to merge those requests; bypassing it (nt / sc1)
// costs 25-30 %.  So the skew is applied in registers instead (skew_taps_in_place).
template <bool BF = false>
__device__ __forceinline__ void load_taps_buf(float (&dst)[KSTEPS], rsrc_t r, uint32_t rowoff, uint32_t pstride,
                                              uint32_t xoff, const int t0 = 0, const int t1 = F)
{
    uint32_t soff = rowoff + (uint32_t)t0 * pstride;
    pin_s(soff);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (f < t0 || f >= t1) continue;
#if SSTEM_ABLATE & 1
        dst[f] = 0.25f; continue;
#endif
        dst[f] = bldc<BF>(r, xoff, soff);
        soff += pstride;
        pin_s(soff);
    }
}

// Blocked coefficients: entry t of the B operand straight from memory -- lane (block, j) reads tap t - j of its own pixel.  The four taps
// of one entry are four neighbouring 256-byte runs inside the row segment's 13 KB (in NCHW they would be four planes: eight cache lines
// from four DRAM pages per instruction, which cost the stream 6-12 %, hence the in-register skew there); no select per entry except
// for the three entries at either end, whose out-of-range lanes read a valid tap and are zeroed.  t in [t0, t1) (constants after
// unrolling).  Same values as load_taps_buf + skew_taps_in_place: bit-identical results, ~150 VALU instructions per pixel row less
// (vector instructions do not co-issue with the 4x4x1 MFMA).
__device__ __forceinline__ void load_taps_skewed_buf(float (&dst)[KSTEPS], rsrc_t r, uint32_t rowoff, uint32_t pstride,
                                                     uint32_t xoff, int sub, const int t0 = 0, const int t1 = KSTEPS)
{
    const uint32_t vo = xoff + (uint32_t)(3 - sub) * pstride;            // interior entries: tap (t - 3) + (3 - sub)
    uint32_t soff = rowoff + (uint32_t)((t0 > 3 ? t0 : 3) - 3) * pstride;
    pin_s(soff);
#pragma unroll
    for (int t = 0; t < KSTEPS; ++t) {
        if (t < t0 || t >= t1) continue;
#if SSTEM_ABLATE & 1
        dst[t] = 0.25f; continue;
#endif
        if (t >= 3 && t < F) {
            dst[t] = bld(r, vo, soff);
            soff += pstride;
            pin_s(soff);
        } else {
            int f = t - sub;
            const bool ok = f >= 0 && f < F;
            f = f < 0 ? 0 : (f > F - 1 ? F - 1 : f);
            const float v = bld(r, xoff + (uint32_t)f * pstride, rowoff);
            dst[t] = ok ? v : 0.f;
        }
    }
}

// B operand of the banded 4x4x1 formulation from the raw taps, in place: h[t] <- H[t - j] for the lane's position
// j = sub in its 4-pixel block, 0 outside [0,51).  Top-down, so that h[t] is overwritten only after entries
// t+1..t+3 (its other readers) are done.  Pure selects: the values, and hence the results, are bit-identical to
// loading them skewed.  Lanes beyond the image edge carry whatever lane 0's address holds (finite coefficient
// data): column j of a block only ever feeds pixel j's own accumulators, which are never stored.
__device__ __forceinline__ void skew_taps_in_place(float (&h)[KSTEPS], int sub)
{
    const bool m1 = sub >= 1, m2 = sub >= 2, m3 = sub == 3;
#pragma unroll
    for (int t = KSTEPS - 1; t >= 0; --t) {
        const float a0 = (t < F) ? h[t] : 0.f;
        const float a1 = (t - 1 >= 0 && t - 1 < F) ? h[t - 1] : 0.f;
        const float a2 = (t - 2 >= 0 && t - 2 < F) ? h[t - 2] : 0.f;
        const float a3 = (t - 3 >= 0 && t - 3 < F) ? h[t - 3] : 0.f;
        h[t] = m3 ? a3 : (m2 ? a2 : (m1 ? a1 : a0));
    }
}

// Channel 0 of one image tile -> LDS rows of P dwords, in BATCH-row groups per thread (the trusted-gray kernel
// holds ~110 coefficient registers while it stages).  REPL: the image is the UNPADDED [Hs, Ws] plane and
// ReplicationPad2d(25) is folded in (clamped coordinates); otherwise it is the padded plane and elements outside it
// are zero.  All addressing is 32-bit (H*W < 2^31, checked by the C-ABI).
template <int THREADS, int ROWS, int P, bool REPL, int BATCH = 11>
__device__ __forceinline__ void stage_gray_tile(float* lds, const float* __restrict__ img, int Hs, int Ws, int y0, int x0)
{
    const int col = threadIdx.x & 127;
    const int rsub = threadIdx.x >> 7;
    constexpr int RSTEP = THREADS / 128;
    constexpr int NPASS = (ROWS + RSTEP - 1) / RSTEP;
    if (col >= TILE_COLS) return;
    int xs = x0 + col - (REPL ? F / 2 : 0);
    const bool col_ok = REPL || xs < Ws;
    xs = xs < 0 ? 0 : (xs > Ws - 1 ? Ws - 1 : xs);
    float* dst = lds + col;
#pragma unroll 1
    for (int k0 = 0; k0 < NPASS; k0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int r = rsub + (k0 + k) * RSTEP;
            int ys = y0 + r - (REPL ? F / 2 : 0);
            const bool ok = col_ok && (REPL || ys < Hs);
            ys = ys < 0 ? 0 : (ys > Hs - 1 ? Hs - 1 : ys);
            const float t = ldg(img, ((uint32_t)ys * (uint32_t)Ws + (uint32_t)xs) * 4u);
            v[k] = ok ? t : 0.f;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int r = rsub + (k0 + k) * RSTEP;
            if (r < ROWS) dst[r * P] = v[k];
        }
    }
}

// The same tile by LDS-DMA (buffer_load ... lds): one 256-byte row piece per wave-instruction straight into LDS, every piece of the
// tile in flight at once, no registers -- ONE memory latency per tile instead of NPASS / BATCH dependent batches.  Elements outside
// the image are read from clamped (valid, finite) addresses instead of being zeroed: they only ever meet coefficients of exactly 0.
typedef __attribute__((address_space(3))) float lds_float;
template <int THREADS, int ROWS, int P, bool REPL>
__device__ __forceinline__ void stage_gray_tile_dma(float* lds, const float* __restrict__ img, int Hs, int Ws, int y0, int x0)
{
    static_assert(P >= 128, "a row is two pieces of 64 columns");
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)((uint32_t)Hs * (uint32_t)Ws * 4u), 0x00020000);
    uint32_t voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int xs = x0 + h * 64 + lane - (REPL ? F / 2 : 0);
        xs = xs < 0 ? 0 : (xs > Ws - 1 ? Ws - 1 : xs);
        voff[h] = (uint32_t)xs * 4u;
    }
#pragma unroll 2
    for (int item = wave; item < ROWS * 2; item += NW) {              // (row, half): wave-uniform
        const int h = item & 1, row = item >> 1;
        int ys = y0 + row - (REPL ? F / 2 : 0);
        ys = ys < 0 ? 0 : (ys > Hs - 1 ? Hs - 1 : ys);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_float*)(lds + row * P + h * 64), 4, (int)(h ? voff[1] : voff[0]),
                                                 (int)((uint32_t)ys * (uint32_t)Ws * 4u), 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): my pieces have landed (the caller's barrier publishes them)
}

// ---- trusted-gray streaming kernel ---------------------------------------------------------------
// The three channels are known to be identical (the device flag written by detect_identical_channels says
// so): only channel 0 is staged (one 576-B row per image row), T[fy] is computed once per pixel row and the
// result is written to all three channels (MODE 0) or folded into the channel mean (MODE 2).  Same MFMA
// sequence per 4-row tile, same fy-ascending accumulation as the generic kernel => bit-identical results.
//
// One third of the MFMAs of the generic kernel means this kernel lives in the memory regime, and what bounds
// it there is the number of coefficient bytes each CU keeps in flight (Little: 24 GB/s per CU x the loaded
// HBM latency).  So both coefficient streams are requested a whole pixel row ahead:
//   * vertical taps: vs[51] holds the current row's taps; the 8 taps an MFMA group has just consumed are
//     re-requested in place for the wave's NEXT row (in flight for a whole row time, no second buffer);
//   * horizontal taps (the B operand, live for the whole row): PFH = true requests the next row's 54 values
//     into a second register set, 9 per MFMA group (two rows per loop trip, the sets swap roles, no copies);
//     PFH = false re-requests them at the row end and relies on the other waves of the SIMD.
// On a wave's last row the requests go to the next phase's first row (MODE 2) or re-read one hot 256-B
// segment of the current row (plane stride 0): every request is unconditional, no exec-mask branches.
//
// BLK: the four coefficient tensors are in the ROW-SEGMENT layout [B][H][tiles_x][51][64] (sstem_sepconv.h, "blocked
// coefficients"): the 51 taps of one 64-pixel row segment are 51 consecutive 256-byte runs, so a wave's requests for one pixel
// row walk 13 KB of consecutive addresses per tensor instead of 256-byte pieces of 51 planes a plane apart.  Same values in the
// same registers => the same bits as the NCHW form.
template <int MODE, int WAVES, int RPW, int WPE, bool PFH, int RING, bool BLK = false, bool BF = false>
__global__ __launch_bounds__(WAVES * 64, WPE) void sepconv_gray_mfma(
    const float* __restrict__ in_a, const float* __restrict__ ver_a, const float* __restrict__ hor_a,
    float* __restrict__ out, TileArgs args, FusedArgs fa)
{
    static_assert(MODE == 0 || MODE == 2, "forward or fused interpolation apply");
    static_assert(!BLK || MODE == 2, "blocked coefficients: fused apply only");
    static_assert(!PFH || (RPW % 2) == 0, "row pairs");
    static_assert(!(BF && BLK), "bf16 coefficient tensors: NCHW only");
    if (fa.gray_flag && *fa.gray_flag == 0) return;   // not identical: the generic build owns this call
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;          // +50 halo +1 pad row (fy = 51, never used)
    constexpr int RS = rm_pitch(1);       // dwords between rows (conflict-free ds_read_b128, see rm_pitch)
    constexpr int NG = 2;                 // concurrent accumulator chains = consecutive 4-row tiles
    constexpr int D = RING - 1;           // A-operand chunks requested ahead of the MFMAs
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W, C = args.C;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    constexpr int YSTEP = WAVES;          // a wave's rows are WAVES apart
    const int ywave = wave;

    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, sub = lane & 3;
    const bool xok = (x0 + lane) < W;
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const int64_t yfirst = (y0 + ywave < H) ? (y0 + ywave) : (H - 1);

    constexpr int NPH = (MODE == 2) ? 2 : 1;
    float hs[KSTEPS], hn[PFH ? KSTEPS : 1], vs[F];

    // bytes between consecutive taps of one pixel row, bytes / elements of one image's coefficients, byte offset of (tap 0, row y, x0)
    const uint32_t plane4 = BLK ? 256u : (uint32_t)plane * 4u;
    const uint32_t seg_row = (uint32_t)args.tiles_x * (uint32_t)(F * 256);      // BLK: bytes of one pixel row of an image
    const uint32_t img_bytes = BLK ? (uint32_t)H * seg_row : (uint32_t)F * plane4;                    // < 4 GiB (launcher)
    const int64_t img_elems = BLK ? (int64_t)(img_bytes >> 2) : (int64_t)F * plane;
    const uint32_t seg_x = (uint32_t)tx * (uint32_t)(F * 256);
    auto rowoff = [&](int64_t y) __attribute__((always_inline)) -> uint32_t {
        return BLK ? (uint32_t)y * seg_row + seg_x : (uint32_t)(y * W + x0) * 4u;
    };
    const uint32_t firstoff = rowoff(yfirst);
    // coefficients of my first row (phase 0); later rows / the second phase arrive through the refills below
    {
        const rsrc_t rv = coef_rsrc_c<BF>(ver_a, b * img_elems, img_bytes);
        const rsrc_t rh = coef_rsrc_c<BF>(hor_a, b * img_elems, img_bytes);
        uint32_t soff = firstoff;
        pin_s(soff);
#pragma unroll
        for (int k = 0; k < F; ++k) {
#if SSTEM_ABLATE & 2
            vs[k] = 0.5f;
#else
            vs[k] = bldc<BF>(rv, xoff, soff);
            soff += plane4;
            pin_s(soff);
#endif
        }
        if constexpr (BLK && SSTEM_BLK_SKEWLD) load_taps_skewed_buf(hs, rh, firstoff, plane4, xoff, sub);
        else load_taps_buf<BF>(hs, rh, firstoff, plane4, xoff);
    }

#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
        const float* in = (MODE == 2 && ph) ? fa.in2 : in_a;
        const float* ver = (MODE == 2 && ph) ? fa.ver2 : ver_a;
        const float* hor = (MODE == 2 && ph) ? fa.hor2 : hor_a;
        // where the refills go: this phase's tensors, or (last row) the first row of the second phase, if there is one
        const bool next_ph = (MODE == 2) && (ph + 1 < NPH);
        const rsrc_t rv_cur = coef_rsrc_c<BF>(ver, b * img_elems, img_bytes);
        const rsrc_t rh_cur = coef_rsrc_c<BF>(hor, b * img_elems, img_bytes);
        const rsrc_t rv_nxt = coef_rsrc_c<BF>(next_ph ? fa.ver2 : ver, b * img_elems, img_bytes);
        const rsrc_t rh_nxt = coef_rsrc_c<BF>(next_ph ? fa.hor2 : hor, b * img_elems, img_bytes);

        if (ph) __syncthreads();          // every wave is done reading the first image's tile
#if !(SSTEM_ABLATE & 4)
#if SSTEM_GRAY_DMA
        if (MODE == 2) stage_gray_tile_dma<WAVES * 64, ROWS, RS, true>(lds, in + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
        else stage_gray_tile_dma<WAVES * 64, ROWS, RS, false>(lds, in + (b * C) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
#else
        if (MODE == 2) stage_gray_tile<WAVES * 64, ROWS, RS, true>(lds, in + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
        else stage_gray_tile<WAVES * 64, ROWS, RS, false>(lds, in + (b * C) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
#endif
#endif
        __syncthreads();

        // one pixel row: B operand hc, requests the wave's next row into hx (PFH) / back into hc (!PFH)
        auto do_row = [&](float (&hc)[KSTEPS], float (&hx)[PFH ? KSTEPS : 1], const int rr, const bool more) __attribute__((always_inline)) {
            const int yl = ywave + rr * YSTEP;
            const int64_t y = y0 + yl;
            const bool fetch = more || next_ph;                          // is there a next row to request?
            const uint32_t pn = fetch ? plane4 : 0u;
            const int64_t ynext = more ? (y + YSTEP) : (next_ph ? yfirst : y);
            const uint32_t nextoff = rowoff(ynext);                      // uniform: (tap 0, next row, x0)
            const rsrc_t rv = more ? rv_cur : rv_nxt;
            const rsrc_t rh = more ? rh_cur : rh_nxt;
            uint32_t vrun = nextoff;                                     // running offset: tap k of the next row
            pin_s(vrun);
            float* dst = out + (MODE == 2 ? (b * H + y) * W : ((b * C) * H + y) * W) + x0;
            pin_uniform(dst);
            float parked = 0.f;            // MODE 2: the first image's channel sum (second phase), requested now so that its
            if (MODE == 2) parked = *stg_ptr(dst, xoff);   // wait at the row end does not drain the refills behind it

            if constexpr (!(BLK && SSTEM_BLK_SKEWLD)) skew_taps_in_place(hc, sub);       // the raw taps requested a row ago (waits for them here); blocked
                                                                   // coefficients arrive skewed (load_taps_skewed_buf)
            const float* arow = lds + (yl + sub) * RS + blk * 4;
            f32x4 ar[RING][NG];
#pragma unroll
            for (int q = 0; q < D; ++q)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    ar[q][g] = *reinterpret_cast<const f32x4*>(arow + g * 4 * RS + q * 4);
            float o = 0.f;
#pragma unroll
            for (int fg = 0; fg < 6; ++fg) {                             // tiles 2fg, 2fg+1
                f32x4 acc[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* abase = arow + fg * (NG * 4) * RS;
                const float* anext = arow + (fg + 1) * (NG * 4) * RS;    // fg == 5: tile 12 (chain 0 only)
                const int gstep = (fg == 5) ? 0 : 4 * RS;               // keep chain 1 inside the image then
                if constexpr (PFH && BLK && SSTEM_BLK_SKEWLD) load_taps_skewed_buf(hx, rh, nextoff, pn, xoff, sub, (SSTEM_HPF * fg < KSTEPS) ? SSTEM_HPF * fg : KSTEPS,
                                                               (SSTEM_HPF * fg + SSTEM_HPF < KSTEPS) ? SSTEM_HPF * fg + SSTEM_HPF : KSTEPS);
                else if constexpr (PFH) load_taps_buf<BF>(hx, rh, nextoff, pn, xoff, (SSTEM_HPF * fg < F) ? SSTEM_HPF * fg : F,
                                                 (SSTEM_HPF * fg + SSTEM_HPF < F) ? SSTEM_HPF * fg + SSTEM_HPF : F);   // SSTEM_HPF taps per MFMA group
#pragma unroll
                for (int tq = 0; tq < 14; ++tq) {
                    const int cc = fg * 14 + tq;                         // running chunk number: ring slot cc % RING
                    if (tq + D < 14) {
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(abase + g * 4 * RS + (tq + D) * 4);
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(anext + g * gstep + (tq + D - 14) * 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = tq * 4 + e;
                        if (t < KSTEPS) {
#if SSTEM_ABLATE & 8
                            if (fg == 0) acc[0][e] += hc[t];
#else
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][g][e], hc[t], acc[g], 0, 0, 0);
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o = fmaf(vs[fg * 8 + g * 4 + i], acc[g][i], o);   // fy ascending
                asm volatile("" : "+v"(o));   // here, not sunk to the store: the accumulators and taps die now
#if !(SSTEM_ABLATE & 2)
#pragma unroll
                for (int i = 0; i < 8; ++i) { vs[fg * 8 + i] = bldc<BF>(rv, xoff, vrun); vrun += pn; pin_s(vrun); }
#endif
            }
            {   // tile 12: rows fy = 48, 49, 50 (+ the pad row)
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* abase = arow + 48 * RS;
#pragma unroll
                for (int tq = 0; tq < 14; ++tq) {
                    const int cc = 6 * 14 + tq;
                    if (tq + D < 14) ar[(cc + D) % RING][0] = *reinterpret_cast<const f32x4*>(abase + (tq + D) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = tq * 4 + e;
#if !(SSTEM_ABLATE & 8)
                        if (t < KSTEPS) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][0][e], hc[t], acc0, 0, 0, 0);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) o = fmaf(vs[48 + i], acc0[i], o);
                asm volatile("" : "+v"(o));
#if !(SSTEM_ABLATE & 2)
#pragma unroll
                for (int i = 0; i < 3; ++i) { vs[48 + i] = bldc<BF>(rv, xoff, vrun); vrun += pn; pin_s(vrun); }
#endif
            }
            if (xok) {
                if (MODE == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { *stg_ptr(dst, xoff) = o; dst += plane; pin_uniform(dst); }
                } else {   // channel sum, then the mean over channels of both images (model_interp.py:94-97)
                    const float csum = (o + o) + o;
                    const float res = ph ? (parked + csum) * (1.0f / 3) : csum;
                    *stg_ptr(dst, xoff) = res;
                    if (ph && fa.out_u8) fa.out_u8[(b * H + y) * W + x0 + lane] = numpy_u8_of(res);
                }
            }
            if constexpr (!PFH && BLK && SSTEM_BLK_SKEWLD) load_taps_skewed_buf(hc, rh, nextoff, pn, xoff, sub);
            else if constexpr (!PFH) load_taps_buf<BF>(hc, rh, nextoff, pn, xoff);
        };
        int nrows = 0;                      // rows of this tile that are mine (wave-uniform)
        if (y0 + ywave < H) {
            const int64_t left = (H - 1 - (y0 + ywave)) / YSTEP + 1;
            nrows = left < RPW ? (int)left : RPW;
        }
        if constexpr (PFH) {
#pragma unroll 1
            for (int rr = 0; rr + 1 < nrows; rr += 2) {
                do_row(hs, hn, rr, true);
                do_row(hn, hs, rr + 1, rr + 2 < nrows);
            }
            if (nrows & 1) {               // odd row count (bottom-edge tiles only)
                do_row(hs, hn, nrows - 1, false);
                if (next_ph) {             // the next phase's B operand was requested into hn
#pragma unroll
                    for (int t = 0; t < KSTEPS; ++t) hs[t] = hn[t];
                }
            }
        } else {
            float dummy[1];
#pragma unroll 1
            for (int rr = 0; rr < nrows; ++rr) do_row(hs, dummy, rr, rr + 1 < nrows);
        }
    }
}

// ---- fused apply on grayscale planes, 16x16x4 formulation (round 5) -------------------------------------------------------------
// Why: tools/micro/mfma_f32_shape_stream_power.hip (profiles/r05): beside a live coefficient stream and under the socket's 1400 W cap,
// v_mfma_f32_16x16x4_f32 at EQUAL USEFUL flops holds the stream 12 % higher than v_mfma_f32_4x4x1 (2170 vs 1909 MHz at the cap): the
// k = 1 shape reads and writes its accumulators and operands four times as often per multiply.
//
// For the 16 neighbouring pixels j of column group cg (pixel x0 + 16 cg + j) of one output row
//     T[fy, j] = sum_{t = 0..66} tile[fy, 16 cg + t] * Hs[t, j],      Hs[t, j] = H[t - j; pixel j]   (0 off-band)
// is D(16 rows x 16 pixels) += A(16 x 4) * B(4 x 16) over 17 k-chunks t = 4 m + k.  Lane l = 16 k + j holds B[m] = H[4 m + k - j] of
// ITS pixel; fy = 0..47 are three 16-row tiles; fy = 48..50 (51 = 3 * 16 + 3) run as ONE v_mfma_f32_4x4x1 per k-chunk on the SAME B
// registers -- block (k, j / 4) of that instruction multiplies rows 48..51 with k-step 4 m + k of pixels 4 (j / 4) .. + 3, i.e. it
// leaves the k-th quarter of the remainder's sum in lane (k, j): 68 instead of 272 matrix-pipe cycles for a fourth tile with 13 dead
// rows.  Matrix-pipe time per 64-pixel row segment and image: 4 x 17 x (3 x 32 + 8) = 7072 cycles (4x4x1 formulation: 5616).
//   * A operand: ds_read_b32 at (row i) * 132 + 16 cg + 4 m + k -- bank 4 i + k: conflict-free; the remainder's rows (i & 3) are
//     broadcasts.  Row pitch 132 dwords (two 64-column LDS-DMA pieces + the bank rotation).
//   * B operand: 13 requests per column group, L[n] = H[4 n + c; pixel] with c = (k - j) & 3: the 64 lanes of a request read four
//     neighbouring tap runs x 64 bytes, every dword once (as many cache-line touches as a coalesced 256-byte request); B[m] = L[m - s],
//     s = -((k - j) >> 2) in 0..4: three conditional register shifts (1, 2, 4), selects only.
//   * vertical stage: lane (g = l / 16, j) holds rows 16 T + 4 g + r of tile T: 12 + 3 vertical taps per column group, requested the
//     same way; the four lanes of a pixel are summed by two ds_bpermute steps ((a + b) + (c + d): the same bits in all four).
//   * both images' tiles are resident (no mid-kernel barrier, the first image's channel sum stays in a register: 134 MB per C2 launch
//     less than the parked form); items = (row, image) pairs; every coefficient register is re-requested in place for the wave's NEXT
//     item as soon as its column-group pair is done (pairs: the two 64-byte halves of a 128-byte line are requested back to back).
// Not bit-identical to the 4x4x1 kernels (another summation order; same products): tests compare it with the oracle at their
// tolerance and with itself across the two coefficient layouts bit for bit.
constexpr int G16_RS = 130;       // dwords between tile rows: ds_read_b32 banks (a / 4) mod 32 per 32-lane group -> 2 j + k: conflict-free
template <int WAVES, int RPW, bool BLK, int VAR = 0>
__global__ __launch_bounds__(WAVES * 64, 2) void sepconv_gray16_mfma(
    const float* __restrict__ in_a, const float* __restrict__ ver_a, const float* __restrict__ hor_a,
    float* __restrict__ out, TileArgs args, FusedArgs fa)
{
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;               // + 50 halo + the pad row (fy = 51: read by the remainder instruction, never used)
    constexpr int RS = G16_RS;
    constexpr int IMG = ROWS * RS;             // floats of one image's tile
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, j = lane & 15;

    const uint32_t tap4 = BLK ? 256u : (uint32_t)plane * 4u;                      // bytes between consecutive taps of one pixel
    const uint32_t seg_row = (uint32_t)args.tiles_x * (uint32_t)(F * 256);
    const uint32_t img_bytes = BLK ? (uint32_t)H * seg_row : (uint32_t)F * tap4;  // < 4 GiB (launcher)
    const int64_t img_elems = BLK ? (int64_t)(img_bytes >> 2) : (int64_t)F * plane;
    const uint32_t seg_x = (uint32_t)tx * (uint32_t)(F * 256);
    auto rowoff = [&](int64_t y) __attribute__((always_inline)) -> uint32_t {
        return BLK ? (uint32_t)y * seg_row + seg_x : (uint32_t)(y * W + x0) * 4u;
    };
    const rsrc_t rv0 = coef_rsrc(ver_a + b * img_elems, img_bytes), rh0 = coef_rsrc(hor_a + b * img_elems, img_bytes);
    const rsrc_t rv1 = coef_rsrc(fa.ver2 + b * img_elems, img_bytes), rh1 = coef_rsrc(fa.hor2 + b * img_elems, img_bytes);

    // per-lane pieces of the addresses (pixels beyond W read whatever follows -- the next row, or zeros behind the resource's end --
    // and their results are never stored; the four lanes that are summed share ONE pixel)
    const int d = kq - j;
    const int c = d & 3, sh = -(d >> 2);                                          // tap residue of this lane; register shift 0..4
    const bool s1 = sh & 1, s2 = sh & 2, s4 = sh & 4, c3 = c == 3;
    const uint32_t vh = (uint32_t)c * tap4 + (uint32_t)j * 4u;                      // + cg * 64 + n * 4 taps
    const uint32_t vv = (uint32_t)(4 * kq) * tap4 + (uint32_t)j * 4u;               // + cg * 64 + (16 T + r) taps
    const uint32_t vr = (uint32_t)j * 4u;                                           // + cg * 64 + (48 + r) taps

    float Bh[4][17], Vv[4][15];
    // requests of column groups cg0 .. cg0 + NC - 1 for the item whose (tap 0, row, x0) offset is `off` (NC = 2: the two 64-byte halves of
    // every 128-byte line back to back)
    auto request = [&](const int cg0, const int NC, const rsrc_t rh, const rsrc_t rv, const uint32_t off) __attribute__((always_inline)) {
        uint32_t so = off;
        pin_s(so);
#pragma unroll
        for (int n = 0; n < 13; ++n) {
#pragma unroll
            for (int q = 0; q < NC; ++q) Bh[cg0 + q][n] = bld(rh, vh + (uint32_t)(cg0 + q) * 64u, so);
            so += 4u * tap4;
            pin_s(so);
        }
#pragma unroll
        for (int T = 0; T < 3; ++T) {
            so = off + (uint32_t)(16 * T) * tap4;
            pin_s(so);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int q = 0; q < NC; ++q) Vv[cg0 + q][4 * T + r] = bld(rv, vv + (uint32_t)(cg0 + q) * 64u, so);
                so += tap4;
                pin_s(so);
            }
        }
        so = off + 48u * tap4;
        pin_s(so);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int q = 0; q < NC; ++q) Vv[cg0 + q][12 + r] = bld(rv, vr + (uint32_t)(cg0 + q) * 64u, so);
            so += tap4;
            pin_s(so);
        }
    };

    int nrows = 0;                          // rows of this tile that are mine (wave-uniform); rows are WAVES apart
    if (y0 + wave < H) {
        const int64_t left = (H - 1 - (y0 + wave)) / WAVES + 1;
        nrows = left < RPW ? (int)left : RPW;
    }
    if (nrows > 0) {                        // the first item's coefficients (image 0, my first row)
        const uint32_t off = rowoff(y0 + wave);
        request(0, 2, rh0, rv0, off);
        request(2, 2, rh0, rv0, off);
    }
    // both tiles: image 0 = (in_a, ver_a, hor_a), image 1 = (fa.in2, ...); ReplicationPad2d(25) folded into the staging
    stage_gray_tile_dma<WAVES * 64, ROWS, RS, true>(lds, in_a + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
    stage_gray_tile_dma<WAVES * 64, ROWS, RS, true>(lds + IMG, fa.in2 + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
    __syncthreads();

    const bool xok = (x0 + lane) < W;
    float first = 0.f;                      // image 0's channel sum of the current row
#pragma unroll 1
    for (int it = 0; it < 2 * nrows; ++it) {
        const int img = it & 1, rr = it >> 1;
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        // the next item: the same row of image 1, or my next row of image 0
        const bool more = it + 1 < 2 * nrows;
        const rsrc_t rhn = img ? rh0 : rh1, rvn = img ? rv0 : rv1;
        const uint32_t nextoff = rowoff(img ? y + WAVES : y);
        const float* abig = lds + img * IMG + (yl + j) * RS + kq;                 // + 16 T rows + 16 cg + 4 m
        const float* arem = lds + img * IMG + (yl + 48 + (j & 3)) * RS + kq;
        float res = 0.f;
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            float (&X)[17] = Bh[cg];
            // B operand from the raw taps, in place (waits for this column group's requests, issued an item ago)
            if constexpr (!(VAR & 2)) {                                             // (VAR bits 2..16: developer ablations, timing only)
            if (c3) X[12] = 0.f;                                                    // tap 4 * 12 + 3 = 51 does not exist
#pragma unroll
            for (int m = 16; m >= 0; --m) {
                const float a0 = m < 13 ? X[m] : 0.f, a1 = (m >= 1 && m - 1 < 13) ? X[m - 1] : 0.f;
                X[m] = s1 ? a1 : a0;
            }
#pragma unroll
            for (int m = 16; m >= 0; --m) X[m] = s2 ? (m >= 2 ? X[m - 2] : 0.f) : X[m];
#pragma unroll
            for (int m = 16; m >= 0; --m) X[m] = s4 ? (m >= 4 ? X[m - 4] : 0.f) : X[m];
            }

            f32x4 acc[3], accr = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < 3; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 17; ++m) {
                const int col = 16 * cg + 4 * m;
#pragma unroll
                for (int T = 0; T < 3; ++T) {
                    if constexpr (VAR & 8) acc[T][m & 3] += abig[16 * T * RS + col] * X[m];
                    else acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(abig[16 * T * RS + col], X[m], acc[T], 0, 0, 0);
                }
                if constexpr (VAR & 8) accr[m & 3] += arem[col] * X[m];
                else accr = __builtin_amdgcn_mfma_f32_4x4x1f32(arem[col], X[m], accr, 0, 0, 0);
            }
            float o = 0.f;
#pragma unroll
            for (int T = 0; T < 3; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) o = fmaf(Vv[cg][4 * T + r], acc[T][r], o);
#pragma unroll
            for (int r = 0; r < 3; ++r) o = fmaf(Vv[cg][12 + r], accr[r], o);
            if constexpr (!(VAR & 16)) {
                o += __shfl_xor(o, 16);
                o += __shfl_xor(o, 32);
            }
            res = (kq == cg) ? o : res;
            if constexpr (VAR & 4) { }
            else if constexpr (VAR & 1) { if (more) request(cg, 1, rhn, rvn, nextoff); }     // every column group as soon as it is done
            else if ((cg & 1) && more) request(cg - 1, 2, rhn, rvn, nextoff);              // wave-uniform: this pair's registers, for my next item
        }
        const float csum = (res + res) + res;
        if (img == 0) first = csum;
        else if (xok) *stg_ptr(out + (b * H + y) * W + x0, (uint32_t)lane * 4u) = (first + csum) * (1.0f / 3);
    }
}

// ---- the same formulation, one column-group PAIR per wave and two coefficient register sets (round 5, second form) --------------
// sepconv_gray16_mfma holds the coefficients of all four column groups of a row segment (128 registers) and can only re-request a
// pair's registers once both of its column groups are done: two bursts of 56 requests per item, at most two steps (of four) ahead of
// their use.  Here a wave owns ONE pair -- 32 pixels of the row segment: waves (2 rg + half), two row groups x two halves per
// workgroup -- so a coefficient set is 64 registers and there are TWO of them: while the matrix pipe works through the current item
// (row, image) out of one set, the next item's 56 requests go into the other, issued a whole item ahead and spread over the item's
// k-chunks.  The two 64-byte halves of every 128-byte line are still requested back to back (one wave owns both column groups).
template <int RPW, bool BLK, bool SKEW_AHEAD = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void sepconv_gray16p_mfma(
    const float* __restrict__ in_a, const float* __restrict__ ver_a, const float* __restrict__ hor_a,
    float* __restrict__ out, TileArgs args, FusedArgs fa)
{
    constexpr int RG = 2;                      // row groups per workgroup (a wave's rows are RG apart)
    constexpr int TR = RG * RPW;
    constexpr int ROWS = TR + F;
    constexpr int RS = G16_RS;
    constexpr int IMG = ROWS * RS;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1, rg = wave >> 1;
    const int64_t y0 = ty * TR, x0 = tx * 64;
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, j = lane & 15;

    const uint32_t tap4 = BLK ? 256u : (uint32_t)plane * 4u;
    const uint32_t seg_row = (uint32_t)args.tiles_x * (uint32_t)(F * 256);
    const uint32_t img_bytes = BLK ? (uint32_t)H * seg_row : (uint32_t)F * tap4;
    const int64_t img_elems = BLK ? (int64_t)(img_bytes >> 2) : (int64_t)F * plane;
    const uint32_t seg_x = (uint32_t)tx * (uint32_t)(F * 256);
    auto rowoff = [&](int64_t y) __attribute__((always_inline)) -> uint32_t {
        return BLK ? (uint32_t)y * seg_row + seg_x : (uint32_t)(y * W + x0) * 4u;
    };
    const rsrc_t rv0 = coef_rsrc(ver_a + b * img_elems, img_bytes), rh0 = coef_rsrc(hor_a + b * img_elems, img_bytes);
    const rsrc_t rv1 = coef_rsrc(fa.ver2 + b * img_elems, img_bytes), rh1 = coef_rsrc(fa.hor2 + b * img_elems, img_bytes);

    const int d = kq - j;
    const int c = d & 3, sh = -(d >> 2);
    const bool s1 = sh & 1, s2 = sh & 2, s4 = sh & 4, c3 = c == 3;
    const uint32_t pxb = (uint32_t)(32 * half + j) * 4u;                           // + 64 bytes for the pair's second column group
    const uint32_t vh = (uint32_t)c * tap4 + pxb;
    const uint32_t vv = (uint32_t)(4 * kq) * tap4 + pxb;
    const uint32_t vr = pxb;

    // one coefficient set: B-operand registers and vertical taps of the pair's two column groups
    struct Set { float Bh[2][17]; float Vv[2][15]; };
    Set A, Bs;
    // request number q (0 .. 55) of an item into set S: 26 horizontal (n = q / 2, column group q & 1), then 30 vertical
    auto request1 = [&](Set& S, const int q, const rsrc_t rh, const rsrc_t rv, const uint32_t off, const uint32_t tstep) __attribute__((always_inline)) {
        if (q < 26) {
            const int n = q >> 1, g = q & 1;
            S.Bh[g][n] = bld(rh, vh + (uint32_t)g * 64u, off + (uint32_t)(4 * n) * tstep);
        } else if (q < 50) {
            const int e = (q - 26) >> 1, g = q & 1, T = e >> 2, r = e & 3;
            S.Vv[g][4 * T + r] = bld(rv, vv + (uint32_t)g * 64u, off + (uint32_t)(16 * T + r) * tstep);
        } else {
            const int r = (q - 50) >> 1, g = q & 1;
            S.Vv[g][12 + r] = bld(rv, vr + (uint32_t)g * 64u, off + (uint32_t)(48 + r) * tstep);
        }
    };

    // elements hi .. lo (top-down) of pass `ps` (0: the tap-51 fix-up, 1 / 2 / 3: the conditional shifts by 1, 2, 4) of the B operand's
    // in-place skew (see sepconv_gray16_mfma); whole passes in order, each pass top-down = the one-shot skew
    auto skew_part = [&](float (&X)[17], const int ps, const int hi, const int lo) __attribute__((always_inline)) {
        if (ps == 0) { if (c3) X[12] = 0.f; return; }
#pragma unroll
        for (int m = 16; m >= 0; --m) {
            const bool on = m <= hi && m >= lo;                  // (a constant once the caller's loop is unrolled: every index stays static)
            const float a0 = (ps == 1 && m >= 13) ? 0.f : X[m];
            const int sft = ps == 1 ? 1 : (ps == 2 ? 2 : 4);
            const bool cnd = ps == 1 ? s1 : (ps == 2 ? s2 : s4);
            float a1 = 0.f;
            if (m - 1 >= 0 && sft == 1 && m - 1 < 13) a1 = X[m - 1];
            if (m - 2 >= 0 && sft == 2) a1 = X[m - 2];
            if (m - 4 >= 0 && sft == 4) a1 = X[m - 4];
            const float nv = cnd ? a1 : a0;
            X[m] = on ? nv : X[m];
        }
    };
    auto skew_all = [&](float (&X)[17]) __attribute__((always_inline)) {
        skew_part(X, 0, 16, 0); skew_part(X, 1, 16, 0); skew_part(X, 2, 16, 0); skew_part(X, 3, 16, 0);
    };

    int nrows = 0;
    if (y0 + rg < H) {
        const int64_t left = (H - 1 - (y0 + rg)) / RG + 1;
        nrows = left < RPW ? (int)left : RPW;
    }
    if (nrows > 0) {
        const uint32_t off = rowoff(y0 + rg);
#pragma unroll
        for (int q = 0; q < 56; ++q) request1(A, q, rh0, rv0, off, tap4);
    }
    stage_gray_tile_dma<256, ROWS, RS, true>(lds, in_a + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
    stage_gray_tile_dma<256, ROWS, RS, true>(lds + IMG, fa.in2 + (b * args.in_planes) * plane, (int)H, (int)W, (int)y0, (int)x0);
    __syncthreads();

    const bool xok = lane < 32 && (x0 + 32 * half + lane) < W;
    float first = 0.f;
    if constexpr (SKEW_AHEAD) { skew_all(A.Bh[0]); skew_all(A.Bh[1]); }
    // one item out of set C; the next item's requests go into set N
    auto item = [&](Set& C, Set& N, const int it) __attribute__((always_inline)) {
        const int img = it & 1, rr = it >> 1;
        const int yl = rg + rr * RG;
        const int64_t y = y0 + yl;
        const bool more = it + 1 < 2 * nrows;
        const rsrc_t rhn = img ? rh0 : rh1, rvn = img ? rv0 : rv1;
        const uint32_t nextoff = more ? rowoff(img ? y + RG : y) : rowoff(y);       // behind the last item: one hot segment, results unused
        const uint32_t tstep = more ? tap4 : 0u;
        const float* abig = lds + img * IMG + (yl + j) * RS + kq + 32 * half;
        const float* arem = lds + img * IMG + (yl + 48 + (j & 3)) * RS + kq + 32 * half;
        float res = 0.f;
        if constexpr (!SKEW_AHEAD) { skew_all(C.Bh[0]); skew_all(C.Bh[1]); }      // (SKEW_AHEAD: done during the previous item)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float (&X)[17] = C.Bh[g];
            f32x4 acc[3], accr = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < 3; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // A operand: rows 16 T + i of the three tiles and row 48 + (i & 3) of the remainder, two k-chunks ahead of the MFMAs
            float ar[3][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int T = 0; T < 3; ++T) ar[q][T] = abig[16 * T * RS + 16 * g + 4 * q];
                ar[q][3] = arem[16 * g + 4 * q];
            }
#pragma unroll
            for (int m = 0; m < 17; ++m) {
                if (m + 2 < 17) {
                    const int col = 16 * g + 4 * (m + 2);
#pragma unroll
                    for (int T = 0; T < 3; ++T) ar[(m + 2) % 3][T] = abig[16 * T * RS + col];
                    ar[(m + 2) % 3][3] = arem[col];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int T = 0; T < 3; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[m % 3][T], X[m], acc[T], 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[m % 3][3], X[m], accr, 0, 0, 0);
                // the next item's 56 requests: four per k-chunk of the item's first 14 (pinned here: left to itself the scheduler sinks
                // them to the end of the item, next to their first use -- the prefetch distance of a whole item becomes none)
                if (g == 0 && m < 14) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) request1(N, 4 * m + q, rhn, rvn, nextoff, tstep);
                }
                // SKEW_AHEAD: the next item's B operand is skewed under the second column group's k-chunks 4 .. 16 (its horizontal taps
                // were the first 26 requests of this item: 21 k-chunks ago at least) -- the selects issue while the matrix pipe works
                if constexpr (SKEW_AHEAD) {
                    if (g == 1 && m >= 4) {
                        const int st = m - 4;                                  // 0 .. 12: pass 1 in 0-3, pass 2 in 4-7, pass 3 in 8-12
                        if (st == 0) { skew_part(N.Bh[0], 0, 16, 0); skew_part(N.Bh[1], 0, 16, 0); }
                        const int ps = st < 4 ? 1 : (st < 8 ? 2 : 3);
                        const int k = st < 4 ? st : (st < 8 ? st - 4 : st - 8);   // part of the pass
                        const int parts = ps == 3 ? 5 : 4;
                        const int hi = 16 - (17 * k) / parts, lo = 16 - (17 * (k + 1)) / parts + 1;
                        skew_part(N.Bh[0], ps, hi, lo); skew_part(N.Bh[1], ps, hi, lo);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float o = 0.f;
#pragma unroll
            for (int T = 0; T < 3; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) o = fmaf(C.Vv[g][4 * T + r], acc[T][r], o);
#pragma unroll
            for (int r = 0; r < 3; ++r) o = fmaf(C.Vv[g][12 + r], accr[r], o);
            o += __shfl_xor(o, 16);
            o += __shfl_xor(o, 32);
            res = (kq == g) ? o : res;
        }
        const float csum = (res + res) + res;
        if (img == 0) first = csum;
        else if (xok) *stg_ptr(out + (b * H + y) * W + x0 + 32 * half, (uint32_t)lane * 4u) = (first + csum) * (1.0f / 3);
    };
#pragma unroll 1
    for (int it = 0; it < 2 * nrows; it += 2) {      // two items per trip (the sets swap roles, no copies); 2 * nrows is even
        item(A, Bs, it);
        item(Bs, A, it + 1);
    }
}

// ---- three independent channels on the streaming structure of the trusted-gray kernel (round 3) ---------------------------
// The op as the reference defines it (kernel.cu:25-52: three distinct channels).  sepconv_rowmajor_mfma (16 waves x 2 rows, 4 waves
// per SIMD) reloads its B operand at every row end, loads its vertical taps one 4-row tile ahead and stages its 115 KB tile through
// registers in six dependent batches with every wave of the CU waiting: 1.64-1.82 ms per C2 call where its 276 M MFMAs need 1.12 ms.
// Here: 8 waves x 4 rows at 2 waves per SIMD (256 registers), and what the gray kernel does --
//   * coefficients through buffer resources with running scalar offsets; horizontal taps requested coalesced and skewed in registers;
//   * both coefficient streams a whole pixel row (2106 MFMAs = 8-10 us) ahead: the 4 vertical taps a tile has consumed are re-requested in
//     place, the next row's horizontal taps arrive in a second register set (copied at the row end: 51 moves beside 2106 MFMAs);
//   * the tile goes global -> LDS by LDS-DMA (buffer_load ... lds, one 256-byte row piece per wave-instruction, every piece of the tile
//     in flight at once, no registers): ONE memory latency per tile instead of six.  Elements outside the image are read from clamped
//     (valid, finite) addresses instead of being zeroed: they only ever meet coefficients of exactly 0 (the same documented
//     non-finite-input deviation as everywhere in the banded formulation).
// Same MFMA sequence per (tile, channel), same fy-ascending accumulation, same channel-sum order: bit-identical to
// sepconv_rowmajor_mfma (tests/test_sepconv_gpu.py).  MODE 0: forward op; MODE 2: fused interpolation apply.
template <int THREADS, int ROWS, int P, bool REPL>
__device__ __forceinline__ void stage_tile3_dma(float* lds, const float* image, uint32_t chan_bytes, int Hs, int Ws, int y0, int x0)
{
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image), 0, (int)(3u * chan_bytes), 0x00020000);
    uint32_t voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int xs = x0 + h * 64 + lane - (REPL ? F / 2 : 0);
        xs = xs < 0 ? 0 : (xs > Ws - 1 ? Ws - 1 : xs);
        voff[h] = (uint32_t)xs * 4u;
    }
#pragma unroll 2
    for (int item = wave; item < ROWS * 6; item += NW) {              // (row, channel, half): wave-uniform
        const int h = item & 1, rc = item >> 1;
        const int row = rc / 3, c = rc - row * 3;
        int ys = y0 + row - (REPL ? F / 2 : 0);
        ys = ys < 0 ? 0 : (ys > Hs - 1 ? Hs - 1 : ys);
        const uint32_t soff = (uint32_t)c * chan_bytes + (uint32_t)ys * (uint32_t)Ws * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_float*)(lds + (row * 3 + c) * P + h * 64), 4,
                                                 (int)(h ? voff[1] : voff[0]), (int)soff, 0, 0);
    }
}

template <int MODE, int WAVES, int RPW>
__global__ __launch_bounds__(WAVES * 64, 1) void sepconv_rgb_stream_mfma(
    const float* __restrict__ in_a, const float* __restrict__ ver_a, const float* __restrict__ hor_a,
    float* __restrict__ out, TileArgs args, FusedArgs fa)
{
    // MODE 1 (round 4): gradVertical, gV[fy] = sum_c g[c] T[c, fy] (kernel.cu:77-112) -- `ver_a` is grad_output [B,3,H,W], `out` is
    // gradVertical [B,51,H,W]: the same tiles T, combined per 4-row tile with the row's three gradient values in the generic kernel's
    // order (s = fma(g2, T2, fma(g1, T1, fma(g0, T0, 0)))) and stored as 51 whole row segments through one buffer resource with a
    // running scalar offset; no vertical-tap stream.  Bit-identical to sepconv_rowmajor_mfma<1, 3, ...>.
    static_assert(MODE == 0 || MODE == 1 || MODE == 2, "forward, gradVertical or fused interpolation apply");
    if (fa.gray_flag && *fa.gray_flag != 0) return;   // identical channels: the trusted-gray kernel owns this call
    constexpr int CH = 3;
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;
    constexpr int P = rm_pitch(3);        // dwords between the channels of a row
    constexpr int RS = CH * P;            // dwords between rows (conflict-free ds_read_b128, see rm_pitch)
    static_assert(P >= 128, "a row piece of the LDS-DMA staging is 2 x 64 columns");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    constexpr int YSTEP = WAVES;
    const int ywave = wave;

    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, sub = lane & 3;
    const bool xok = (x0 + lane) < W;
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const int64_t yfirst = (y0 + ywave < H) ? (y0 + ywave) : (H - 1);

    constexpr int NPH = (MODE == 2) ? 2 : 1;
    float hs[KSTEPS], hn[KSTEPS], vs[MODE == 1 ? 1 : F];

    const uint32_t plane4 = (uint32_t)plane * 4u;
    const uint32_t img_bytes = (uint32_t)F * plane4;                    // < 4 GiB (launcher)
    const uint32_t firstoff = (uint32_t)(yfirst * W + x0) * 4u;
    {   // coefficients of my first row (phase 0); later rows / the second phase arrive through the refills below
        const rsrc_t rh = coef_rsrc(hor_a + (b * F) * plane, img_bytes);
        if constexpr (MODE != 1) {
            const rsrc_t rv = coef_rsrc(ver_a + (b * F) * plane, img_bytes);
            uint32_t soff = firstoff;
            pin_s(soff);
#pragma unroll
            for (int k = 0; k < F; ++k) { vs[k] = bld(rv, xoff, soff); soff += plane4; pin_s(soff); }
        }
        load_taps_buf(hs, rh, firstoff, plane4, xoff);
    }

#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
        const float* in = (MODE == 2 && ph) ? fa.in2 : in_a;
        const float* ver = (MODE == 2 && ph) ? fa.ver2 : ver_a;
        const float* hor = (MODE == 2 && ph) ? fa.hor2 : hor_a;
        const bool next_ph = (MODE == 2) && (ph + 1 < NPH);
        const rsrc_t rv_cur = coef_rsrc((MODE == 1 ? hor : ver) + (b * F) * plane, img_bytes);      // (MODE 1: unused)
        const rsrc_t rh_cur = coef_rsrc(hor + (b * F) * plane, img_bytes);
        const rsrc_t rv_nxt = coef_rsrc((MODE == 1 ? hor : (next_ph ? fa.ver2 : ver)) + (b * F) * plane, img_bytes);
        const rsrc_t rh_nxt = coef_rsrc((next_ph ? fa.hor2 : hor) + (b * F) * plane, img_bytes);
        const rsrc_t rgv = coef_rsrc(out + (MODE == 1 ? (b * F) * plane : 0), img_bytes);           // MODE 1: gradVertical of image b

        if (ph) __syncthreads();          // every wave is done reading the first image's tile
        if (MODE == 2) stage_tile3_dma<WAVES * 64, ROWS, P, true>(lds, in + (b * CH) * plane, plane4, (int)H, (int)W, (int)y0, (int)x0);
        else stage_tile3_dma<WAVES * 64, ROWS, P, false>(lds, in + (b * CH) * Hin * Win, (uint32_t)(Hin * Win) * 4u, (int)Hin, (int)Win, (int)y0, (int)x0);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): my pieces of the tile have landed in LDS
        __syncthreads();

        auto do_row = [&](const int rr, const bool more) __attribute__((always_inline)) {
            const int yl = ywave + rr * YSTEP;
            const int64_t y = y0 + yl;
            const bool fetch = more || next_ph;                          // is there a next row to request?
            const uint32_t pn = fetch ? plane4 : 0u;
            const int64_t ynext = more ? (y + YSTEP) : (next_ph ? yfirst : y);
            const uint32_t nextoff = (uint32_t)(ynext * W + x0) * 4u;    // uniform: (tap 0, next row, x0)
            const rsrc_t rv = more ? rv_cur : rv_nxt;
            const rsrc_t rh = more ? rh_cur : rh_nxt;
            uint32_t vrun = nextoff;
            pin_s(vrun);
            float* dst = out + (MODE == 2 ? (b * H + y) * W : ((b * CH) * H + y) * W) + x0;
            pin_uniform(dst);
            float parked = 0.f;            // MODE 2: the first image's channel sum (second phase), requested at the row start
            if (MODE == 2) parked = *stg_ptr(dst, xoff);
            float gch[CH];                 // MODE 1: the row's three gradient values
            uint32_t srun = (uint32_t)(y * W + x0) * 4u;                 // MODE 1: running store offset, tap fy of this row
            pin_s(srun);
            if constexpr (MODE == 1) {
                const float* gp = ver + (b * CH) * plane + y * W + x0;
                pin_uniform(gp);
#pragma unroll
                for (int c = 0; c < CH; ++c) { gch[c] = ldg(gp, xoff); gp += plane; pin_uniform(gp); }
            }

            skew_taps_in_place(hs, sub);       // the raw taps requested a row ago (waits for them here)
            const float* arow = lds + (yl + sub) * RS + blk * 4;
            constexpr int RING = SSTEM_RGB_RING, D = RING - 1;     // A-operand chunks requested ahead of the MFMAs
            f32x4 ar[RING][CH];
#pragma unroll
            for (int q = 0; q < D; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) ar[q][c] = *reinterpret_cast<const f32x4*>(arow + c * P + q * 4);
            float o[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) o[c] = 0.f;
#pragma unroll
            for (int ft = 0; ft < 13; ++ft) {                            // 4-row tiles: fy = 4 ft .. 4 ft + 3
                f32x4 acc[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* abase = arow + ft * 4 * RS;
                const float* anext = arow + ((ft == 12) ? 0 : (ft + 1) * 4 * RS);     // behind the last tile: a valid address, unused
                // the next row's horizontal taps, 5 per tile (all requested by tile 10)
                load_taps_buf(hn, rh, nextoff, pn, xoff, (5 * ft < F) ? 5 * ft : F, (5 * ft + 5 < F) ? 5 * ft + 5 : F);
#pragma unroll
                for (int tq = 0; tq < 14; ++tq) {
                    const int cc = ft * 14 + tq;                         // running chunk number: ring slot cc % RING
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        ar[(cc + D) % RING][c] = *reinterpret_cast<const f32x4*>((tq + D < 14 ? abase + (tq + D) * 4 : anext + (tq + D - 14) * 4) + c * P);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = tq * 4 + e;
                        if (t < KSTEPS) {
#pragma unroll
                            for (int c = 0; c < CH; ++c)
                                acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][c][e], hs[t], acc[c], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MODE == 1) {
                    if (xok) {     // lanes beyond the image edge store nothing; the running offset inside is a local (uniform in here)
                        uint32_t so = srun;
                        pin_s(so);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int fy = ft * 4 + i;
                            if (fy < F) {
                                float sacc = 0.f;
#pragma unroll
                                for (int c = 0; c < CH; ++c) sacc = fmaf(gch[c], acc[c][i], sacc);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgv, (int)xoff, (int)so, 0);
                                so += plane4;
                                pin_s(so);
                            }
                        }
                    }
                    srun += 4u * plane4;
                    pin_s(srun);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int fy = ft * 4 + i;
                        if (fy < F) {   // fy == 51 is the pad row: never used
#pragma unroll
                            for (int c = 0; c < CH; ++c) o[c] = fmaf(vs[fy], acc[c][i], o[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c) asm volatile("" : "+v"(o[c]));   // the accumulators and taps die here, not at the store
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int fy = ft * 4 + i;
                        if (fy < F) { vs[fy] = bld(rv, xoff, vrun); vrun += pn; pin_s(vrun); }
                    }
                }
            }
            if (xok && MODE != 1) {
                if (MODE == 0) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) { *stg_ptr(dst, xoff) = o[c]; dst += plane; pin_uniform(dst); }
                } else {   // channel sum in the generic kernel's order, then the mean over channels of both images (model_interp.py:94-97)
                    float csum = o[0];
                    csum += o[1];
                    csum += o[2];
                    *stg_ptr(dst, xoff) = ph ? (parked + csum) * (1.0f / CH) : csum;
                }
            }
#pragma unroll
            for (int f = 0; f < F; ++f) hs[f] = hn[f];                   // the next row's raw taps (skewed at its start)
        };
        int nrows = 0;                      // rows of this tile that are mine (wave-uniform)
        if (y0 + ywave < H) {
            const int64_t left = (H - 1 - (y0 + ywave)) / YSTEP + 1;
            nrows = left < RPW ? (int)left : RPW;
        }
#pragma unroll 1
        for (int rr = 0; rr < nrows; ++rr) do_row(rr, rr + 1 < nrows);
    }
}

// ---- trusted-gray gradVertical --------------------------------------------------------------------
// gV[fy] = sum_c g[c] * T[c,fy]  (kernel.cu:77-112).  With identical input channels T[c,fy] is the same for every c:
// it is computed ONCE (one third of the MFMAs) and combined with the three gradient channels in the generic kernel's
// order, s = fma(g2, T, fma(g1, T, fma(g0, T, 0))) -- the same T bits (same MFMA sequence on the same data) and the
// same FMA chain, so the result is bit-identical to the generic path (not the cheaper (g0+g1+g2)*T, which rounds
// differently).  Streams H in (coalesced taps skewed in registers, as the forward gray kernel) and gV out
// (51 row segments per pixel row through one buffer resource, one scalar add per tap).
template <int WAVES, int RPW, int WPE, bool PFH, int RING, bool BF = false>
__global__ __launch_bounds__(WAVES * 64, WPE) void sepconv_gray_gradv_mfma(
    const float* __restrict__ in, const float* __restrict__ gout, const float* __restrict__ hor,
    float* __restrict__ gv, TileArgs args, const int* __restrict__ gray_flag)
{
    static_assert(!PFH || (RPW % 2) == 0, "row pairs");
    if (gray_flag && *gray_flag == 0) return;   // not identical: the generic build owns this call
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F;
    constexpr int RS = rm_pitch(1);
    constexpr int NG = 2;
    constexpr int D = RING - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, sub = lane & 3;
    const bool xok = (x0 + lane) < W;
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const int64_t yfirst = (y0 + wave < H) ? (y0 + wave) : (H - 1);

    const uint32_t plane4 = (uint32_t)plane * 4u;
    const uint32_t img_bytes = (uint32_t)F * plane4;                    // < 4 GiB (launcher)
    const rsrc_t rh = coef_rsrc_c<BF>(hor, (b * F) * plane, img_bytes);
    const rsrc_t rgv = coef_rsrc(gv + (b * F) * plane, img_bytes);
    const float* g_b = gout + (b * 3) * plane + x0;                     // uniform: channel 0 of image b

    float hs[KSTEPS], hn[PFH ? KSTEPS : 1];
    load_taps_buf<BF>(hs, rh, (uint32_t)(yfirst * W + x0) * 4u, plane4, xoff);

    // round 4: the tile by LDS-DMA, as in the forward kernel (one memory latency per tile instead of dependent register batches)
#if SSTEM_GRAY_DMA
    stage_gray_tile_dma<WAVES * 64, ROWS, RS, false>(lds, in + (b * 3) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
#else
    stage_gray_tile<WAVES * 64, ROWS, RS, false>(lds, in + (b * 3) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
#endif
    __syncthreads();

    auto do_row = [&](float (&hc)[KSTEPS], float (&hx)[PFH ? KSTEPS : 1], const int rr, const bool more) __attribute__((always_inline)) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        const uint32_t pn = more ? plane4 : 0u;
        const uint32_t rowoff = (uint32_t)(y * W + x0) * 4u;             // (tap 0, this row, x0)
        const uint32_t nextoff = (uint32_t)((more ? y + WAVES : y) * W + x0) * 4u;
        const float* gp = g_b + y * W;
        pin_uniform(gp);
        float gch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { gch[c] = ldg(gp, xoff); gp += plane; pin_uniform(gp); }
        uint32_t srun = rowoff;                                          // running store offset: tap fy of this row
        pin_s(srun);

        skew_taps_in_place(hc, sub);
        const float* arow = lds + (yl + sub) * RS + blk * 4;
        f32x4 ar[RING][NG];
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g)
                ar[q][g] = *reinterpret_cast<const f32x4*>(arow + g * 4 * RS + q * 4);
#pragma unroll
        for (int fg = 0; fg < 6; ++fg) {
            f32x4 acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* abase = arow + fg * (NG * 4) * RS;
            const float* anext = arow + (fg + 1) * (NG * 4) * RS;
            const int gstep = (fg == 5) ? 0 : 4 * RS;
            if constexpr (PFH) load_taps_buf<BF>(hx, rh, nextoff, pn, xoff, 11 * fg, (11 * fg + 11 < F) ? 11 * fg + 11 : F);
#pragma unroll
            for (int tq = 0; tq < 14; ++tq) {
                const int cc = fg * 14 + tq;
                if (tq + D < 14) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(abase + g * 4 * RS + (tq + D) * 4);
                } else {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(anext + g * gstep + (tq + D - 14) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tq * 4 + e;
                    if (t < KSTEPS) {
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][g][e], hc[t], acc[g], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (xok) {     // lanes beyond the image edge store nothing; the running offset inside is a local (it stays
                uint32_t so = srun;   // wave-uniform in here and dies here: srun itself must not change under a divergent branch)
                pin_s(so);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                        // fy = 8fg + 4g + i, ascending
                        float sacc = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sacc = fmaf(gch[c], acc[g][i], sacc);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgv, (int)xoff, (int)so, 0);
                        so += plane4;
                        pin_s(so);
                    }
            }
            srun += 8u * plane4;
            pin_s(srun);
        }
        {   // tile 12: fy = 48, 49, 50
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* abase = arow + 48 * RS;
#pragma unroll
            for (int tq = 0; tq < 14; ++tq) {
                const int cc = 6 * 14 + tq;
                if (tq + D < 14) ar[(cc + D) % RING][0] = *reinterpret_cast<const f32x4*>(abase + (tq + D) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tq * 4 + e;
                    if (t < KSTEPS) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][0][e], hc[t], acc0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (xok) {
                uint32_t so = srun;
                pin_s(so);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float sacc = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) sacc = fmaf(gch[c], acc0[i], sacc);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgv, (int)xoff, (int)so, 0);
                    so += plane4;
                    pin_s(so);
                }
            }
        }
        if constexpr (!PFH) load_taps_buf<BF>(hc, rh, nextoff, pn, xoff);
    };

    int nrows = 0;
    if (y0 + wave < H) {
        const int64_t left = (H - 1 - (y0 + wave)) / WAVES + 1;
        nrows = left < RPW ? (int)left : RPW;
    }
    if constexpr (PFH) {
#pragma unroll 1
        for (int rr = 0; rr + 1 < nrows; rr += 2) {
            do_row(hs, hn, rr, true);
            do_row(hn, hs, rr + 1, rr + 2 < nrows);
        }
        if (nrows & 1) do_row(hs, hn, nrows - 1, false);
    } else {
        float dummy[1];
#pragma unroll 1
        for (int rr = 0; rr < nrows; ++rr) do_row(hs, dummy, rr, rr + 1 < nrows);
    }
}

// ---- gradHorizontal: column-major LDS image ---------------------------------------------------
// dword index of tile element (c, col, r):  (c*TCOLS + col)*PITCH_T + r, PITCH_T = 4*odd so the
// ds_read_b128 of 64 consecutive columns (same 4-row chunk) is conflict-free.
constexpr int TCOLS = 120;   // 64 + 50 halo, + t-tiles reach col 4*13+3+63 = 118
constexpr int KSTEPS_T = 56; // 14 aligned 4-row chunks cover 51 taps at any row phase

template <int CH, int WAVES, int RPW, bool COALESCE>
__global__ __launch_bounds__(WAVES * 64) void sepconv_gradh_mfma(
    const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ ver,
    float* __restrict__ gh, TileArgs args, const int* __restrict__ gray_flag)
{
    if (gray_flag && *gray_flag != 0) return;               // identical channels: sepconv_gray_gradh_mfma owns this call
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;                         // aligned chunks start up to 3 rows early
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;        // multiple of 4 dwords, (PITCH_T/4) odd
    constexpr int CSTRIDE = TCOLS * PITCH_T;
    static_assert(((PITCH_T / 4) & 1) == 1 && PITCH_T >= ROWS, "pitch");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W, C = args.C;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int64_t y0 = ty * TR, x0 = tx * 64;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & 3;
    const int64_t x = x0 + lane;
    const bool xok = x < W;
    const bool ld_ok = xok && !(args.dbg & 2);
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const float* ver_b = ver + (b * F) * plane + x0;                 // uniform bases
    const float* g_b = g + (b * C + args.c0) * plane + x0;

    // K runs over 4-row aligned chunks starting at row k0 = yl & ~3; the vertical coefficient of
    // LDS row k0 + k is V[k - (yl&3)], zero outside [0,51).
    float vs[KSTEPS_T];
    {
        const int64_t yf = (y0 + wave < H) ? (y0 + wave) : (H - 1);
        const int sh = wave & 3;   // wave-uniform skew
        load_skewed<KSTEPS_T>(vs, ver_b + yf * W, plane, xoff, sh, ld_ok);
    }

    // stage the tile transposed: thread -> column (coalesced global read along x)
    if (!(args.dbg & 1)) {
        const int col = threadIdx.x & 127;
        const int rsub = threadIdx.x >> 7;
        constexpr int RSTEP = (WAVES * 64) / 128;
        if (col < TCOLS) {
            const bool col_ok = x0 + col < Win;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float* src = in + ((b * C + (args.c0 + c)) * Hin + y0) * Win + x0 + col;
                float* dst = lds + (c * TCOLS + col) * PITCH_T;
#pragma unroll 4
                for (int r = rsub; r < ROWS; r += RSTEP) {
                    float v = 0.f;
                    if (col_ok && (y0 + r < Hin)) v = src[(int64_t)r * Win];
                    dst[r] = v;
                }
            }
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        if (y >= H) break;

        constexpr bool PF = (WAVES <= 8);
        float vn[PF ? KSTEPS_T : 1];
        const bool more = (rr + 1 < RPW) && (y + WAVES < H);
        if constexpr (PF) {
            const int64_t yn = more ? (y + WAVES) : y;     // see the row-major kernel: unconditional, stride 0 on the last row
            load_skewed<KSTEPS_T>(vn, ver_b + yn * W, more ? plane : 0, xoff, (yl + WAVES) & 3, ld_ok && more);
        }
        const int k0 = yl & ~3;
        float gch[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            gch[c] = xok ? ldg(g_b + (int64_t)c * plane + y * W, xoff) : 0.f;

        // A operand: lane (blk, i) <-> tile column lane + 4*tt, rows k0 + 4*kq .. +3
        const float* abase = lds + lane * PITCH_T + k0;
        f32x4 a_cur[CH], a_nxt[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) a_cur[c] = *reinterpret_cast<const f32x4*>(abase + c * CSTRIDE);

        const int ntt = (args.dbg & 4) ? 1 : 14;
        float carry[3] = {0.f, 0.f, 0.f};    // COALESCE: entries t = 4tt-3 .. 4tt-1 of the previous tile
#pragma unroll 1
        for (int tt = 0; tt < ntt; ++tt) {
            f32x4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* acol = abase + tt * 4 * PITCH_T;
            const float* anext = abase + ((tt == 13) ? 0 : (tt + 1) * 4 * PITCH_T);
#pragma unroll
            for (int kq = 0; kq < 14; ++kq) {
                if (kq < 13) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        a_nxt[c] = *reinterpret_cast<const f32x4*>(acol + c * CSTRIDE + (kq + 1) * 4);
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        a_nxt[c] = *reinterpret_cast<const f32x4*>(anext + c * CSTRIDE);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[c][e], vs[kq * 4 + e], acc[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CH; ++c) a_cur[c] = a_nxt[c];
            }
            // acc[c][i] = G_c[t = 4tt+i ; my pixel j=sub];  gH[fx = t - j]
            if constexpr (COALESCE) {
                // Re-sort in registers so that every store is one whole 256-B row segment of ONE tap plane (measured on
                // the gray kernel: four-plane 64-B pieces cost 20-25 %): plane f takes entry t = f + j, a 4-way select
                // over a window of this tile's four entries and the last three of the previous tile.  Same values.
                float w[7];
#pragma unroll
                for (int u = 0; u < 3; ++u) w[u] = carry[u];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float sc = 0.f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) sc = fmaf(gch[c], acc[c][i], sc);
                    w[3 + i] = sc;
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) carry[u] = w[4 + u];
                if (xok) {
                    const bool m1 = sub >= 1, m2 = sub >= 2, m3 = sub == 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = tt * 4 - 3 + q;                    // wave-uniform
                        if (f >= 0 && f < F) {
                            const float val = m3 ? w[q + 3] : (m2 ? w[q + 2] : (m1 ? w[q + 1] : w[q]));
                            gfloat* dst = stg_ptr(gh + ((b * F + f) * H + y) * W + x0, xoff);
                            if (args.c0 == 0) *dst = val; else *dst += val;
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int fx = tt * 4 + i - sub;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) s = fmaf(gch[c], acc[c][i], s);
                if (xok && fx >= 0 && fx < F) {
                    // plane fx = 4tt+i-sub is lane-dependent: uniform base of plane (4tt+i-3) + per-lane
                    // ((3-sub)*plane + lane)*4
                    gfloat* dst = stg_ptr(gh + ((b * F + (tt * 4 + i - 3)) * H + y) * W + x0,
                                         (uint32_t)((3 - sub) * plane) * 4u + xoff);
                    if (args.c0 == 0) *dst = s; else *dst += s;
                }
            }
        }
        if constexpr (PF) {
#pragma unroll
            for (int k = 0; k < KSTEPS_T; ++k) vs[k] = vn[k];
        } else {
            if (more) load_skewed<KSTEPS_T>(vs, ver_b + (y + WAVES) * W, plane, xoff, (yl + WAVES) & 3, ld_ok);
        }
    }
}

// ---- trusted-gray gradHorizontal -----------------------------------------------------------------
// gH[fx] = sum_c g[c] * G_c[fx],  G_c[fx;p] = sum_fy V[fy;p] * in[c, y+fy, x+fx]  (kernel.cu:115-150).  With identical
// input channels G_c is the same for every c: computed once on channel 0 with the generic kernel's MFMA sequence
// (same aligned 4-row chunks, same order) and combined with the three gradient channels in the generic FMA order
// => bit-identical to sepconv_gradh_mfma<3>.  Two tiles (8 values of t) run as concurrent accumulator chains.
// The B operand (the pixel's vertical taps shifted by the wave-uniform row phase) comes through a buffer resource
// with a running scalar offset; every lane of a request reads the same tap plane (coalesced).  Lane j holds, for
// entry t, the result of tap fx = t - j: COALESCE = false stores it as it stands (64-B pieces of four neighbouring
// tap planes per store); COALESCE = true re-sorts in registers first (plane f takes entry f + j: a 4-way select over
// a sliding window that carries three entries from tile pair to tile pair) so that every store is one whole 256-B row
// segment of one plane.  Selects only: same bits either way.
template <int THREADS, int ROWS, int PITCH_T, int BATCH = 11>
__device__ __forceinline__ void stage_gray_tile_colmajor(float* lds, const float* __restrict__ img, int Hs, int Ws, int y0, int x0)
{
    // channel 0 of the padded image -> lds[col * PITCH_T + row]; thread -> column (coalesced global reads along x)
    const int col = threadIdx.x & 127;
    const int rsub = threadIdx.x >> 7;
    constexpr int RSTEP = THREADS / 128;
    constexpr int NPASS = (ROWS + RSTEP - 1) / RSTEP;
    if (col >= TCOLS) return;
    int xs = x0 + col;
    const bool col_ok = xs < Ws;
    xs = xs > Ws - 1 ? Ws - 1 : xs;
    float* dst = lds + col * PITCH_T;
#pragma unroll 1
    for (int k0 = 0; k0 < NPASS; k0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int r = rsub + (k0 + k) * RSTEP;
            int ys = y0 + r;
            const bool ok = col_ok && ys < Hs;
            ys = ys > Hs - 1 ? Hs - 1 : ys;
            const float t = ldg(img, ((uint32_t)ys * (uint32_t)Ws + (uint32_t)xs) * 4u);
            v[k] = ok ? t : 0.f;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int r = rsub + (k0 + k) * RSTEP;
            if (r < ROWS) dst[r] = v[k];
        }
    }
}

// vs[k] = V[k - sh] of the lane's pixel for k in [k0, k1), 0 outside [0,51); sh = row phase (wave-uniform, 0..3).
// rowoff = byte offset of (tap 0, row, x0); pstride = plane bytes or 0 (hot re-read, results unused).  The running
// offset points at tap clamp(k - sh, 0, 50): entries outside the band re-read a neighbouring tap and are zeroed.
template <bool BF = false>
__device__ __forceinline__ void load_phase_taps_buf(float (&dst)[KSTEPS_T], rsrc_t r, uint32_t rowoff, uint32_t pstride,
                                                    uint32_t xoff, int sh, const int k0 = 0, const int k1 = KSTEPS_T)
{
    int tap0 = k0 - sh;                                  // scalar
    tap0 = tap0 < 0 ? 0 : (tap0 > F - 1 ? F - 1 : tap0);
    uint32_t soff = rowoff + (uint32_t)tap0 * pstride;
    pin_s(soff);
#pragma unroll
    for (int k = 0; k < KSTEPS_T; ++k) {
        if (k < k0 || k >= k1) continue;
        const int tap = k - sh;                          // scalar
        const bool valid = tap >= 0 && tap < F;
        const float v = bldc<BF>(r, xoff, soff);
        dst[k] = valid ? v : 0.f;
        soff += (tap >= 0 && tap < F - 1) ? pstride : 0u;   // advance while the next entry's tap is a new valid one
        pin_s(soff);
    }
}

template <int WAVES, int RPW, int WPE, bool PFH, int RING, bool COALESCE, bool BF = false>
__global__ __launch_bounds__(WAVES * 64, WPE) void sepconv_gray_gradh_mfma(
    const float* __restrict__ in, const float* __restrict__ gout, const float* __restrict__ ver,
    float* __restrict__ gh, TileArgs args, const int* __restrict__ gray_flag)
{
    static_assert(!PFH || (RPW % 2) == 0, "row pairs");
    static_assert((WAVES % 4) == 0, "a wave keeps its row phase from row to row");
    if (gray_flag && *gray_flag == 0) return;   // not identical: the generic build owns this call
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    static_assert(((PITCH_T / 4) & 1) == 1 && PITCH_T >= ROWS, "pitch");
    constexpr int NG = 2;
    constexpr int D = RING - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 3;
    const bool xok = (x0 + lane) < W;
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const int64_t yfirst = (y0 + wave < H) ? (y0 + wave) : (H - 1);
    const int sh = wave & 3;                    // row phase of every row of this wave (TR and WAVES are multiples of 4)

    const uint32_t plane4 = (uint32_t)plane * 4u;
    const uint32_t img_bytes = (uint32_t)F * plane4;                    // < 4 GiB (launcher)
    const rsrc_t rv = coef_rsrc_c<BF>(ver, (b * F) * plane, img_bytes);
    const rsrc_t rgh = coef_rsrc(gh + (b * F) * plane, img_bytes);
    const float* g_b = gout + (b * 3) * plane + x0;
    const uint32_t lane_plane = (uint32_t)(3 - sub) * plane4 + xoff;    // store voffset of entries t >= 3: tap (t-3) + (3-sub)

    float vs[KSTEPS_T], vn[PFH ? KSTEPS_T : 1];
    load_phase_taps_buf<BF>(vs, rv, (uint32_t)(yfirst * W + x0) * 4u, plane4, xoff, sh);

    stage_gray_tile_colmajor<WAVES * 64, ROWS, PITCH_T>(lds, in + (b * 3) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
    __syncthreads();

    auto do_row = [&](float (&vc)[KSTEPS_T], float (&vx)[PFH ? KSTEPS_T : 1], const int rr, const bool more) __attribute__((always_inline)) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        const uint32_t pn = more ? plane4 : 0u;
        const uint32_t rowoff = (uint32_t)(y * W + x0) * 4u;
        const uint32_t nextoff = (uint32_t)((more ? y + WAVES : y) * W + x0) * 4u;
        const float* gp = g_b + y * W;
        pin_uniform(gp);
        float gch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { gch[c] = ldg(gp, xoff); gp += plane; pin_uniform(gp); }
        uint32_t srun = rowoff;                  // running store offset: tap max(t - 3, 0) of this row
        pin_s(srun);

        const int k0 = yl & ~3;
        const float* abase = lds + lane * PITCH_T + k0;   // lane <-> tile column lane + 4*tt, rows k0 + 4*kq .. +3
        f32x4 ar[RING][NG];
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g)
                ar[q][g] = *reinterpret_cast<const f32x4*>(abase + g * 4 * PITCH_T + q * 4);
        float carry[3] = {0.f, 0.f, 0.f};        // COALESCE: entries t = 8p-3 .. 8p-1 of the previous tile pair
#pragma unroll
        for (int p = 0; p < 7; ++p) {                                    // tiles tt = 2p, 2p + 1
            f32x4 acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* acol = abase + p * (NG * 4) * PITCH_T;
            const float* anext = abase + ((p == 6) ? 0 : (p + 1) * (NG * 4) * PITCH_T);   // p == 6: valid address, unused
            if constexpr (PFH) load_phase_taps_buf<BF>(vx, rv, nextoff, pn, xoff, sh, 8 * p, 8 * p + 8);
#pragma unroll
            for (int kq = 0; kq < 14; ++kq) {
                const int cc = p * 14 + kq;
                if (kq + D < 14) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(acol + g * 4 * PITCH_T + (kq + D) * 4);
                } else {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        ar[(cc + D) % RING][g] = *reinterpret_cast<const f32x4*>(anext + g * 4 * PITCH_T + (kq + D - 14) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[cc % RING][g][e], vc[kq * 4 + e], acc[g], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // acc[g][i] = G[t = 8p + 4g + i ; my pixel j = sub];  gH[fx = t - j]
            if constexpr (COALESCE) {
                float w[11];                     // entries t = 8p - 3 + u
#pragma unroll
                for (int u = 0; u < 3; ++u) w[u] = carry[u];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float sacc = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sacc = fmaf(gch[c], acc[g][i], sacc);
                        w[3 + g * 4 + i] = sacc;
                    }
#pragma unroll
                for (int u = 0; u < 3; ++u) carry[u] = w[8 + u];
                if (xok) {
                    uint32_t so = srun;          // running offset of plane f (local: uniform inside the divergent region)
                    pin_s(so);
                    const bool m1 = sub >= 1, m2 = sub >= 2, m3 = sub == 3;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int f = p * 8 - 3 + q;                     // plane f takes entry t = f + j = w[q + j]
                        if (f < 0 || f >= F) continue;
                        const float val = m3 ? w[q + 3] : (m2 ? w[q + 2] : (m1 ? w[q + 1] : w[q]));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rgh, (int)xoff, (int)so, 0);
                        so += plane4;
                        pin_s(so);
                    }
                }
            } else
            if (xok) {
                uint32_t so = srun;              // local: stays wave-uniform inside the divergent region (see gradVertical)
                pin_s(so);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = p * 8 + g * 4 + i;
                        if (t >= F + 3) continue;                        // fx = t - j >= 51 for every j
                        float sacc = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sacc = fmaf(gch[c], acc[g][i], sacc);
                        if (t >= 3 && t < F) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgh, (int)lane_plane, (int)so, 0);
                        } else if (t < 3) {                              // uniform part = tap 0; lanes with j <= t store tap t - j
                            if (sub <= t)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgh,
                                                                      (int)((uint32_t)(t - sub) * plane4 + xoff), (int)so, 0);
                        } else {                                         // t = 51..53: lanes with t - j <= 50
                            if (t - sub < F)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sacc), rgh, (int)lane_plane, (int)so, 0);
                        }
                        if (t >= 3) { so += plane4; pin_s(so); }
                    }
            }
            srun += (uint32_t)((p == 0) ? 5 : 8) * plane4;               // t = 3..7 advance in the first pair, all 8 afterwards
            pin_s(srun);
        }
        if constexpr (!PFH) load_phase_taps_buf<BF>(vc, rv, nextoff, pn, xoff, sh);
    };

    int nrows = 0;
    if (y0 + wave < H) {
        const int64_t left = (H - 1 - (y0 + wave)) / WAVES + 1;
        nrows = left < RPW ? (int)left : RPW;
    }
    if constexpr (PFH) {
#pragma unroll 1
        for (int rr = 0; rr + 1 < nrows; rr += 2) {
            do_row(vs, vn, rr, true);
            do_row(vn, vs, rr + 1, rr + 2 < nrows);
        }
        if (nrows & 1) do_row(vs, vn, nrows - 1, false);
    } else {
        float dummy[1];
#pragma unroll 1
        for (int rr = 0; rr < nrows; ++rr) do_row(vs, dummy, rr, rr + 1 < nrows);
    }
}

// ---- gradHorizontal for three independent channels on the streaming structure (round 4) ----------------------------------------
// sepconv_gradh_mfma<3, 16, 2> (round 1) reloads its 56 phase-shifted vertical taps at every row end through per-lane 64-bit
// addresses and runs 4 waves per SIMD at 128 registers: 2.3 ms per C2 call where its 308 M MFMAs need 1.25 ms.  Here, what the
// trusted-gray gradient kernel does, for CH = 3: 8 waves x 4 rows at 2 waves per SIMD, the taps through one buffer resource with a
// running scalar offset, the NEXT row's 56 taps requested into a second register set while the current row's 2352 MFMAs run (the two
// sets swap roles from row to row), results re-sorted in registers into whole 256-byte row-segment stores.  Same MFMA sequence per
// (tile, channel) -- k ascending over the 14 aligned 4-row chunks --, same fma chain over the three gradient channels, same
// re-sort: bit-identical to sepconv_gradh_mfma<3, ..., true> (tests/test_sepconv_gpu.py).
template <int WAVES, int RPW>
__global__ __launch_bounds__(WAVES * 64, 1) void sepconv_rgb_gradh_stream_mfma(
    const float* __restrict__ in, const float* __restrict__ gout, const float* __restrict__ ver,
    float* __restrict__ gh, TileArgs args, const int* __restrict__ gray_flag)
{
    static_assert((RPW % 2) == 0, "row pairs");
    static_assert((WAVES % 4) == 0, "a wave keeps its row phase from row to row");
    if (gray_flag && *gray_flag != 0) return;   // identical channels: sepconv_gray_gradh_mfma owns this call
    constexpr int CH = 3;
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    constexpr int CSTRIDE = TCOLS * PITCH_T;
    static_assert(((PITCH_T / 4) & 1) == 1 && PITCH_T >= ROWS, "pitch");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    int64_t b, ty, tx;
    decode_block(args, b, ty, tx);
    const int64_t H = args.H, W = args.W;
    const int64_t Hin = H + F - 1, Win = W + F - 1;
    const int64_t plane = H * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t y0 = ty * TR, x0 = tx * 64;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 3;
    const bool xok = (x0 + lane) < W;
    const uint32_t xoff = (uint32_t)(xok ? lane : 0) * 4u;
    const int64_t yfirst = (y0 + wave < H) ? (y0 + wave) : (H - 1);
    const int sh = wave & 3;                    // row phase of every row of this wave

    const uint32_t plane4 = (uint32_t)plane * 4u;
    const uint32_t img_bytes = (uint32_t)F * plane4;                    // < 4 GiB (launcher)
    const rsrc_t rv = coef_rsrc(ver + (b * F) * plane, img_bytes);
    const rsrc_t rgh = coef_rsrc(gh + (b * F) * plane, img_bytes);
    const float* g_b = gout + (b * CH) * plane + x0;

    float vs[KSTEPS_T], vn[KSTEPS_T];
    load_phase_taps_buf(vs, rv, (uint32_t)(yfirst * W + x0) * 4u, plane4, xoff, sh);

#pragma unroll
    for (int c = 0; c < CH; ++c)
        stage_gray_tile_colmajor<WAVES * 64, ROWS, PITCH_T>(lds + c * CSTRIDE, in + (b * CH + c) * Hin * Win, (int)Hin, (int)Win, (int)y0, (int)x0);
    __syncthreads();

    auto do_row = [&](float (&vc)[KSTEPS_T], float (&vx)[KSTEPS_T], const int rr, const bool more) __attribute__((always_inline)) {
        const int yl = wave + rr * WAVES;
        const int64_t y = y0 + yl;
        const uint32_t pn = more ? plane4 : 0u;
        const uint32_t rowoff = (uint32_t)(y * W + x0) * 4u;
        const uint32_t nextoff = (uint32_t)((more ? y + WAVES : y) * W + x0) * 4u;
        const float* gp = g_b + y * W;
        pin_uniform(gp);
        float gch[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { gch[c] = ldg(gp, xoff); gp += plane; pin_uniform(gp); }
        uint32_t srun = rowoff;                  // running store offset: plane max(4 tt - 3, 0) of this row
        pin_s(srun);

        const int k0 = yl & ~3;
        const float* abase = lds + lane * PITCH_T + k0;   // lane <-> tile column lane + 4*tt, rows k0 + 4*kq .. +3
        f32x4 a_cur[CH], a_nxt[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) a_cur[c] = *reinterpret_cast<const f32x4*>(abase + c * CSTRIDE);
        float carry[3] = {0.f, 0.f, 0.f};        // entries t = 4 tt - 3 .. 4 tt - 1 of the previous tile
#pragma unroll
        for (int tt = 0; tt < 14; ++tt) {
            f32x4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* acol = abase + tt * 4 * PITCH_T;
            const float* anext = abase + ((tt == 13) ? 0 : (tt + 1) * 4 * PITCH_T);
            load_phase_taps_buf(vx, rv, nextoff, pn, xoff, sh, 4 * tt, 4 * tt + 4);       // the next row's taps, four per tile
#pragma unroll
            for (int kq = 0; kq < 14; ++kq) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a_nxt[c] = *reinterpret_cast<const f32x4*>((kq < 13 ? acol + (kq + 1) * 4 : anext) + c * CSTRIDE);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[c][e], vc[kq * 4 + e], acc[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CH; ++c) a_cur[c] = a_nxt[c];
            }
            // acc[c][i] = G_c[t = 4 tt + i ; my pixel j = sub];  gH[fx = t - j]: plane f takes entry t = f + j (the generic kernel's re-sort)
            float w[7];
#pragma unroll
            for (int u = 0; u < 3; ++u) w[u] = carry[u];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) sc = fmaf(gch[c], acc[c][i], sc);
                w[3 + i] = sc;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) carry[u] = w[4 + u];
            if (xok) {
                uint32_t so = srun;              // local: stays wave-uniform inside the divergent region
                pin_s(so);
                const bool m1 = sub >= 1, m2 = sub >= 2, m3 = sub == 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = tt * 4 - 3 + q;
                    if (f < 0 || f >= F) continue;
                    const float val = m3 ? w[q + 3] : (m2 ? w[q + 2] : (m1 ? w[q + 1] : w[q]));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rgh, (int)xoff, (int)so, 0);
                    so += plane4;
                    pin_s(so);
                }
            }
            srun += (uint32_t)((tt == 0) ? 1 : 4) * plane4;              // tile 0 stores plane 0 only (f = -3 .. 0)
            pin_s(srun);
        }
    };

    int nrows = 0;
    if (y0 + wave < H) {
        const int64_t left = (H - 1 - (y0 + wave)) / WAVES + 1;
        nrows = left < RPW ? (int)left : RPW;
    }
#pragma unroll 1
    for (int rr = 0; rr + 1 < nrows; rr += 2) {
        do_row(vs, vn, rr, true);
        do_row(vn, vs, rr + 1, rr + 2 < nrows);
    }
    if (nrows & 1) do_row(vs, vn, nrows - 1, false);
}

// ---- device-side dispatch between the generic and the trusted-gray build --------------------------
// detect_identical_channels clears *flag when any element of channel 1 or 2 differs (bitwise) from channel 0.
// The flag words live in the code object (no allocation by the library), one array per device.  A call takes the
// slot of ITS STREAM: work on one stream is ordered, so the flag fill of a stream's next call cannot overtake the
// kernels of its previous one, and two streams never share a slot -- no aliasing however many calls are in flight.
// Calls issued while a stream is being captured into a graph take a slot of their own each (a replayed graph may
// run on any stream).  When a device runs out of slots (more than GRAY_STREAM_SLOTS live streams, or more than
// GRAY_CAPTURE_SLOTS captured calls) the call is served by the generic build alone (exact, per-tile vote; slower).
constexpr int GRAY_STREAM_SLOTS = 1024;
constexpr int GRAY_CAPTURE_SLOTS = 3072;
constexpr int MAX_DEVICES = 64;
__device__ int g_gray_flags[GRAY_STREAM_SLOTS + GRAY_CAPTURE_SLOTS];

__global__ __launch_bounds__(256) void detect_identical_channels(const float* __restrict__ a, const float* __restrict__ b2,
                                                                 int64_t nimg, int64_t plane_elems, int* flag)
{
    // a (and b2 when not null) are [nimg, 3, plane_elems] tensors
    unsigned diff = 0u;
    const int64_t total = nimg * plane_elems;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / plane_elems, off = i - img * plane_elems;
        const float* p = a + img * 3 * plane_elems + off;
        const unsigned c0 = __float_as_uint(p[0]);
        diff |= (c0 ^ __float_as_uint(p[plane_elems])) | (c0 ^ __float_as_uint(p[2 * plane_elems]));
        if (b2) {
            const float* q = b2 + img * 3 * plane_elems + off;
            const unsigned d0 = __float_as_uint(q[0]);
            diff |= (d0 ^ __float_as_uint(q[plane_elems])) | (d0 ^ __float_as_uint(q[2 * plane_elems]));
        }
    }
    if (__any(diff != 0u) && (threadIdx.x & 63) == 0) *flag = 0;    // plain store of the same value from several waves
}

struct GrayFlagState {            // one per device
    std::mutex m;
    int* base = nullptr;
    std::unordered_map<hipStream_t, int> slot_of_stream;
    int captured = 0;
};

static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return -1;
    return d;
}

// Returns the flag word of this call (filled with 1 = "identical until proven otherwise" on the stream), or nullptr with
// e == hipSuccess when no slot is free (the caller then launches the generic build alone).
static int* next_gray_flag(hipStream_t s, hipError_t& e)
{
    static GrayFlagState states[MAX_DEVICES];
    e = hipSuccess;
    const int dev = current_device();
    if (dev < 0) return nullptr;
    GrayFlagState& st = states[dev];
    int idx = -1;
    {
        std::lock_guard<std::mutex> lock(st.m);
        if (!st.base) {
            e = hipGetSymbolAddress(reinterpret_cast<void**>(&st.base), HIP_SYMBOL(g_gray_flags));   // the current device's copy
            if (e != hipSuccess) { st.base = nullptr; return nullptr; }
        }
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) {
            if (st.captured < GRAY_CAPTURE_SLOTS) idx = GRAY_STREAM_SLOTS + st.captured++;
        } else {
            auto it = st.slot_of_stream.find(s);
            if (it != st.slot_of_stream.end()) idx = it->second;
            else if ((int)st.slot_of_stream.size() < GRAY_STREAM_SLOTS) {
                idx = (int)st.slot_of_stream.size();
                st.slot_of_stream.emplace(s, idx);
            }
        }
    }
    if (idx < 0) return nullptr;
    int* slot = st.base + idx;
    e = hipMemsetAsync(slot, 1, sizeof(int), s);    // non-zero = "identical until proven otherwise"
    return e == hipSuccess ? slot : nullptr;
}

static hipError_t launch_detect(const float* a, const float* b2, int64_t nimg, int64_t plane_elems, int* flag,
                                hipStream_t s)
{
    int64_t g = (nimg * plane_elems + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(detect_identical_channels, dim3((unsigned)g), dim3(256), 0, s, a, b2, nimg, plane_elems, flag);
    return hipGetLastError();
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
    hipLaunchKernelGGL(sepconv_fwd_direct<false>, dim3(grid_1d(B * H * W, 256)), dim3(256), 0, s,
                       in, ver, hor, out, B, C, H, W, filt, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_bwd_direct(const float* g, const float* in, const float* ver, const float* hor,
                             float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                             int filt, hipStream_t s)
{
    const int grid = grid_1d(B * filt * H * W, 256);
    hipLaunchKernelGGL((sepconv_grad_direct<true, false>), dim3(grid), dim3(256), 0, s,
                       g, in, hor, gv, B, C, H, W, filt, (const int*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sepconv_grad_direct<false, false>), dim3(grid), dim3(256), 0, s,
                       g, in, ver, gh, B, C, H, W, filt, (const int*)nullptr);
    return hipGetLastError();
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair: `done` is the calling launcher
// instantiation's own bit set of devices that already have it (one static per kernel instantiation).
template <typename K>
static hipError_t set_lds(K kernel, size_t bytes, std::atomic<uint64_t>& done)
{
    const int dev = current_device();
    if (dev < 0) return hipErrorInvalidDevice;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// Tile shapes: WAVES waves x RPW rows per wave.  0: 8x4, 1: 12x3, 2: 12x2, 3: 16x2 (default), 4: 16x3 -- the
// developer knob SSTEM_TILE picks one for A/B runs.
static int tile_variant()
{
    static const int cached = [] {
        const char* e = getenv("SSTEM_TILE");
        const int v = e ? atoi(e) : 3;   // default: 16 waves x 2 rows (fastest measured on MI355X)
        return (v >= 0 && v <= 4) ? v : 3;
    }();
    return cached;
}
static int tile_rows(int variant) { return variant == 0 ? 32 : (variant == 1 ? 36 : (variant == 2 ? 24 : (variant == 3 ? 32 : 48))); }

template <int MODE, int CH, int WAVES, int RPW>
static hipError_t launch_rowmajor_v(const float* in, const float* vg, const float* hor, float* out,
                                    const TileArgs& a, hipStream_t s,
                                    FusedArgs fa = FusedArgs{nullptr, nullptr, nullptr, nullptr})
{
    constexpr int TR = WAVES * RPW;
    constexpr int CHL = CH;
    constexpr size_t lds_bytes = (size_t)CHL * (TR + F) * rm_pitch_tile(CHL, WAVES, RPW) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_rowmajor_mfma<MODE, CH, WAVES, RPW>;
    static std::atomic<uint64_t> lds_set{0};                // per instantiation: devices whose attribute is set
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, vg, hor, out, a, fa);
    return hipGetLastError();
}

template <int MODE, int CH>
static hipError_t launch_rowmajor(const float* in, const float* vg, const float* hor, float* out,
                                  const TileArgs& a, hipStream_t s)
{
    switch (tile_variant()) {
        case 1: return launch_rowmajor_v<MODE, CH, 12, 3>(in, vg, hor, out, a, s);
        case 2: return launch_rowmajor_v<MODE, CH, 12, 2>(in, vg, hor, out, a, s);
        case 3: return launch_rowmajor_v<MODE, CH, 16, 2>(in, vg, hor, out, a, s);
        case 4: return launch_rowmajor_v<MODE, CH, 16, 3>(in, vg, hor, out, a, s);
        default: return launch_rowmajor_v<MODE, CH, 8, 4>(in, vg, hor, out, a, s);
    }
}

// SSTEM_GH_COALESCE=0 keeps the four-plane store pieces (A/B runs); default: results re-sorted into whole-row stores
static bool gradh_coalesce()
{
    static const bool on = [] { const char* e = getenv("SSTEM_GH_COALESCE"); return !(e && atoi(e) == 0); }();
    return on;
}

template <int CH, int WAVES, int RPW, bool COALESCE>
static hipError_t launch_gradh_vc(const float* in, const float* g, const float* ver, float* gh,
                                  const TileArgs& a, hipStream_t s, const int* flag)
{
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    constexpr size_t lds_bytes = (size_t)CH * TCOLS * PITCH_T * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_gradh_mfma<CH, WAVES, RPW, COALESCE>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, g, ver, gh, a, flag);
    return hipGetLastError();
}

template <int CH, int WAVES, int RPW>
static hipError_t launch_gradh_v(const float* in, const float* g, const float* ver, float* gh,
                                 const TileArgs& a, hipStream_t s, const int* flag = nullptr)
{
    return gradh_coalesce() ? launch_gradh_vc<CH, WAVES, RPW, true>(in, g, ver, gh, a, s, flag)
                            : launch_gradh_vc<CH, WAVES, RPW, false>(in, g, ver, gh, a, s, flag);
}

// three independent channels: the streaming form (8 waves x 4 rows, 2 waves per SIMD) under the same switch and limits as the forward's
static hipError_t launch_rgb_gradh_stream(const float* in, const float* g, const float* ver, float* gh, TileArgs a, hipStream_t s,
                                          const int* flag)
{
    constexpr int WAVES = 8, RPW = 4, TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    constexpr size_t lds_bytes = (size_t)3 * TCOLS * PITCH_T * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_rgb_gradh_stream_mfma<WAVES, RPW>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, g, ver, gh, a, flag);
    return hipGetLastError();
}

template <int CH>
static hipError_t launch_gradh(const float* in, const float* g, const float* ver, float* gh,
                               const TileArgs& a, hipStream_t s)
{
    switch (tile_variant()) {
        case 1: return launch_gradh_v<CH, 12, 3>(in, g, ver, gh, a, s);
        case 2: return launch_gradh_v<CH, 12, 2>(in, g, ver, gh, a, s);
        case 3: return launch_gradh_v<CH, 16, 2>(in, g, ver, gh, a, s);
        case 4: return launch_gradh_v<CH, 16, 3>(in, g, ver, gh, a, s);
        default: return launch_gradh_v<CH, 8, 4>(in, g, ver, gh, a, s);
    }
}

static TileArgs make_args(int64_t B, int64_t C, int64_t H, int64_t W)
{
    TileArgs a;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.tiles_x = (W + 63) / 64;
    const int tr = tile_rows(tile_variant());
    a.tiles_y = (H + tr - 1) / tr;
    a.c0 = 0;
    a.in_planes = 3;
    static const int dbg = [] { const char* d = getenv("SSTEM_DEBUG_FLAGS"); return d ? atoi(d) : 0; }();
    a.dbg = dbg;                                    // developer ablations only (see TileArgs::dbg)
    return a;
}

bool mfma_grid_ok(int64_t B, int64_t H, int64_t W)
{
    TileArgs a = make_args(B, 1, H, W);
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    return nwg > 0 && nwg <= 0x7fffffffLL;
}

// Trusted-gray launch.  Developer knob SSTEM_GRAY_SHAPE picks the workgroup shape / prefetch scheme for A/B runs:
//   0: 4 waves x 8 rows, 3 waves per SIMD, B operand re-requested at the row end
//   1: 4 waves x 8 rows, 2 waves per SIMD, B operand of the next row prefetched into a second register set
//   2: 4 waves x 16 rows, otherwise as 1      3: as 2 with a 2-deep A ring
//   6 / 7 / 8: 2 x 8, 4 x 4, 2 x 4 rows at 3 waves per SIMD (small grids; 7 is the default below 512 workgroups)
template <int MODE, int WAVES, int RPW, int WPE, bool PFH, int RING, bool BLK = false, bool BF = false>
static hipError_t launch_gray_v(const float* in, const float* ver, const float* hor, float* out, TileArgs a,
                                hipStream_t s, const FusedArgs& fa)
{
    constexpr int TR = WAVES * RPW;
    constexpr size_t lds_bytes = (size_t)(TR + F) * rm_pitch(1) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_gray_mfma<MODE, WAVES, RPW, WPE, PFH, RING, BLK, BF>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, ver, hor, out, a, fa);
    return hipGetLastError();
}

template <int MODE, bool BLK = false>
static hipError_t launch_gray(const float* in, const float* vg, const float* hor, float* out, TileArgs a,
                              hipStream_t s, const FusedArgs& fa)
{
    static const int forced = [] { const char* e = getenv("SSTEM_GRAY_SHAPE"); return e ? atoi(e) : -1; }();
    // default: the tall 2-waves-per-SIMD shape (fastest measured at C2) when its 64-row tiles still give every CU
    // several workgroups, otherwise the 32-row shape (small images: 256x256 has only 4 x 4 tall tiles per image)
    int shape = forced;
    if (shape < 0) {
        shape = (a.B * a.tiles_x * ((a.H + 63) / 64) >= 1024) ? 3 : 0;
        // fewer than 512 of the 32-row tiles (8 x 256 x 256: 256): 16-row tiles, 4 rows per wave -- two workgroups per CU keep twice the
        // coefficient rows in flight (0.111 -> 0.105 ms on the 8 x 256 x 256 apply; the 51 extra halo rows are 2 % of the bytes)
        if (shape == 0 && a.B * a.tiles_x * ((a.H + 31) / 32) < 512) shape = 7;
    }
    // blocked coefficients with the LDS-DMA staging: the 32-row, 3-waves-per-SIMD shape is ahead of the tall one at C2 (1.349 vs 1.364 ms,
    // same box, profiles/r03) -- the tile's bigger halo share no longer costs staging time, and three waves hide more latency than the
    // B-operand prefetch of two
    if (BLK && forced < 0 && shape == 3) shape = 0;
    if constexpr (BLK) {   // blocked coefficients: the three shapes the default rule picks (the developer shapes stay NCHW-only)
        switch (shape) {
            case 3: return launch_gray_v<MODE, 4, 16, 2, true, 2, true>(in, vg, hor, out, a, s, fa);
            case 7: return launch_gray_v<MODE, 4, 4, 3, false, 2, true>(in, vg, hor, out, a, s, fa);
            default: return launch_gray_v<MODE, 4, 8, 3, false, 2, true>(in, vg, hor, out, a, s, fa);
        }
    }
    switch (shape) {
        case 1: return launch_gray_v<MODE, 4, 8, 2, true, 3>(in, vg, hor, out, a, s, fa);
        case 2: return launch_gray_v<MODE, 4, 16, 2, true, 3>(in, vg, hor, out, a, s, fa);
        case 3: return launch_gray_v<MODE, 4, 16, 2, true, 2>(in, vg, hor, out, a, s, fa);
        case 4: return launch_gray_v<MODE, 4, 8, 2, false, 3>(in, vg, hor, out, a, s, fa);
        case 5: return launch_gray_v<MODE, 4, 16, 2, false, 3>(in, vg, hor, out, a, s, fa);
        case 6: return launch_gray_v<MODE, 2, 8, 3, false, 2>(in, vg, hor, out, a, s, fa);
        case 7: return launch_gray_v<MODE, 4, 4, 3, false, 2>(in, vg, hor, out, a, s, fa);
        case 8: return launch_gray_v<MODE, 2, 4, 3, false, 2>(in, vg, hor, out, a, s, fa);
        default: return launch_gray_v<MODE, 4, 8, 3, false, 2>(in, vg, hor, out, a, s, fa);
    }
}

// Three independent channels: the streaming kernel (8 waves x 4 rows, LDS-DMA staging) unless SSTEM_RGB_STREAM=0 (A/B runs: the
// round-1 kernel sepconv_rowmajor_mfma) or the image's coefficient planes do not fit a 32-bit buffer resource.
static bool rgb_stream_enabled(int64_t H, int64_t W)
{
    static const bool on = [] { const char* e = getenv("SSTEM_RGB_STREAM"); return !(e && atoi(e) == 0); }();
    return on && (uint64_t)F * (uint64_t)H * (uint64_t)W * 4u < (1ull << 32) && (uint64_t)3 * (uint64_t)(H + F) * (uint64_t)(W + F) * 4u < (1ull << 32);
}

template <int MODE>
static hipError_t launch_rgb_stream(const float* in, const float* ver, const float* hor, float* out, TileArgs a, hipStream_t s,
                                    const FusedArgs& fa)
{
    constexpr int WAVES = 8, RPW = 4, TR = WAVES * RPW;
    constexpr size_t lds_bytes = (size_t)(TR + F) * 3 * rm_pitch(3) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_rgb_stream_mfma<MODE, WAVES, RPW>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, ver, hor, out, a, fa);
    return hipGetLastError();
}

// SSTEM_GRAY_KERNEL=0 disables the trusted-gray build + device dispatch (A/B runs); the in-kernel per-tile vote stays.
static bool gray_dispatch_enabled(int64_t H, int64_t W)
{
    static const bool on = [] { const char* e = getenv("SSTEM_GRAY_KERNEL"); return !(e && atoi(e) == 0); }();
    // the trusted-gray kernel addresses one image's 51 coefficient planes through a 32-bit buffer resource
    return on && (uint64_t)F * (uint64_t)H * (uint64_t)W * 4u < (1ull << 32);
}

hipError_t launch_fwd_mfma(const float* in, const float* ver, const float* hor, float* out,
                           int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s)
{
    TileArgs a = make_args(B, C, H, W);
    hipError_t e = hipSuccess;
    int* flag = nullptr;
    if (C == 3 && gray_dispatch_enabled(H, W) && tile_rows(tile_variant()) == 32) {
        flag = next_gray_flag(s, e);          // nullptr without an error: no free slot, the generic build serves the call
        if (e != hipSuccess) return e;
    }
    if (flag) {
        // detect -> generic build (returns at once on gray input) -> trusted-gray build (returns at once otherwise)
        e = launch_detect(in, nullptr, B, (H + F - 1) * (W + F - 1), flag, s);
        if (e != hipSuccess) return e;
        const FusedArgs fa{nullptr, nullptr, nullptr, flag};
        if (rgb_stream_enabled(H, W)) e = launch_rgb_stream<0>(in, ver, hor, out, a, s, fa);
        else switch (tile_variant()) {
            case 0: e = launch_rowmajor_v<0, 3, 8, 4>(in, ver, hor, out, a, s, fa); break;
            default: e = launch_rowmajor_v<0, 3, 16, 2>(in, ver, hor, out, a, s, fa); break;
        }
        if (e != hipSuccess) return e;
        return launch_gray<0>(in, ver, hor, out, a, s, fa);
    }
    if (C == 3 && rgb_stream_enabled(H, W))
        return launch_rgb_stream<0>(in, ver, hor, out, a, s, FusedArgs{nullptr, nullptr, nullptr, nullptr});
    for (int64_t c0 = 0; c0 < C && e == hipSuccess; c0 += 3) {
        a.c0 = (int)c0;
        const int64_t ch = (C - c0) < 3 ? (C - c0) : 3;
        if (ch == 3) e = launch_rowmajor<0, 3>(in, ver, hor, out, a, s);
        else if (ch == 2) e = launch_rowmajor<0, 2>(in, ver, hor, out, a, s);
        else e = launch_rowmajor<0, 1>(in, ver, hor, out, a, s);
    }
    return e;
}

hipError_t launch_interp_fused(const float* i1, const float* i2, const float* k1v, const float* k1h,
                               const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                               hipStream_t s)
{
    // y = sepconv(pad(i2), k2v, k2h) + sepconv(pad(i1), k1v, k1h): phase 0 = image 2, phase 1 = image 1
    TileArgs a = make_args(B, 3, H, W);
    FusedArgs fa{i1, k1v, k1h, nullptr};
    if (gray_dispatch_enabled(H, W)) {
        hipError_t e = hipSuccess;
        int* flag = next_gray_flag(s, e);
        if (e != hipSuccess) return e;
        if (flag) {
            e = launch_detect(i1, i2, B, H * W, flag, s);
            if (e != hipSuccess) return e;
            fa.gray_flag = flag;
            e = launch_gray<2>(i2, k2v, k2h, out, a, s, fa);
            if (e != hipSuccess) return e;
        }
    }
    // measured on MI355X: the 8-wave shape (next-row coefficient prefetch, 256-register budget) wins the fused
    // launch on grayscale frames (2.06 vs 2.35 ms) and ties on independent channels; SSTEM_FUSED_TILE overrides
    static const int fv = [] { const char* e = getenv("SSTEM_FUSED_TILE"); return e ? atoi(e) : 0; }();
    if (fv == 0 && rgb_stream_enabled(H, W)) return launch_rgb_stream<2>(i2, k2v, k2h, out, a, s, fa);
    if (fv == 1) { a.tiles_y = (H + 35) / 36; return launch_rowmajor_v<2, 3, 12, 3>(i2, k2v, k2h, out, a, s, fa); }
    if (fv == 0) { a.tiles_y = (H + 31) / 32; return launch_rowmajor_v<2, 3, 8, 4>(i2, k2v, k2h, out, a, s, fa); }
    a.tiles_y = (H + 31) / 32;
    return launch_rowmajor_v<2, 3, 16, 2>(i2, k2v, k2h, out, a, s, fa);
}

// The 16x16x4 formulation of the fused apply on grayscale planes (sepconv_gray16_mfma): OPT-IN (SSTEM_GRAY16=1, read at every call so
// that a test can switch it).  Measured at C2 (profiles/r05): 1.36 ms against the 4x4x1 kernel's 1.32 on the same box -- it runs at
// 2.08 GHz instead of 1.52 under the same 1400 W cap (what the micro-benchmark predicted) with the matrix pipe 63 % busy, but holds
// 128 coefficient registers per lane (two waves per SIMD) and refills them in two bursts per item, a shorter prefetch distance than the
// 4x4x1 kernel's row-ahead refills: the stream it sustains is the same.  DESIGN 4.5 has the ablation table and what would be next.
static int gray16_mode()          // 0: the 4x4x1 kernel; 1: four column groups per wave; 2: one pair per wave, two register sets
{
    const char* e = getenv("SSTEM_GRAY16");
    return e ? atoi(e) : 0;
}
static bool gray16_enabled() { const int m = gray16_mode(); return m >= 1 && m <= 3; }

template <int WAVES, int RPW, bool BLK, int VAR = 0>
static hipError_t launch_gray16_v(const float* in, const float* ver, const float* hor, float* out, TileArgs a, hipStream_t s, const FusedArgs& fa)
{
    constexpr int TR = WAVES * RPW;
    constexpr size_t lds_bytes = (size_t)2 * (TR + F) * G16_RS * sizeof(float);
    static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU");
    auto k = sepconv_gray16_mfma<WAVES, RPW, BLK, VAR>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, ver, hor, out, a, fa);
    return hipGetLastError();
}

template <int RPW, bool BLK, bool SKEW_AHEAD = false>
static hipError_t launch_gray16p_v(const float* in, const float* ver, const float* hor, float* out, TileArgs a, hipStream_t s, const FusedArgs& fa)
{
    constexpr int TR = 2 * RPW;
    constexpr size_t lds_bytes = (size_t)2 * (TR + F) * G16_RS * sizeof(float);
    static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU");
    auto k = sepconv_gray16p_mfma<RPW, BLK, SKEW_AHEAD>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(256), lds_bytes, s, in, ver, hor, out, a, fa);
    return hipGetLastError();
}

template <bool BLK>
static hipError_t launch_gray16(const float* in, const float* ver, const float* hor, float* out, const TileArgs& a, hipStream_t s, const FusedArgs& fa)
{
    if (gray16_mode() == 3) {                 // ... with the next item's B operand skewed under the current item's MFMAs
        if (a.B * a.tiles_x * ((a.H + 23) / 24) < 512) return launch_gray16p_v<4, BLK, true>(in, ver, hor, out, a, s, fa);
        return launch_gray16p_v<12, BLK, true>(in, ver, hor, out, a, s, fa);
    }
    if (gray16_mode() == 2) {                 // the ping-pong form: 24-row tiles (12 rows per wave), 8-row tiles for small grids
        if (a.B * a.tiles_x * ((a.H + 23) / 24) < 512) return launch_gray16p_v<4, BLK>(in, ver, hor, out, a, s, fa);
        return launch_gray16p_v<12, BLK>(in, ver, hor, out, a, s, fa);
    }
    // 24-row tiles (6 rows per wave); 16-row tiles when those give fewer than two workgroups per CU
    static const int forced = [] { const char* e = getenv("SSTEM_GRAY16_ROWS"); return e ? atoi(e) : 0; }();
    const bool small = forced ? forced == 16 : a.B * a.tiles_x * ((a.H + 23) / 24) < 512;
    if (small) return launch_gray16_v<4, 4, BLK>(in, ver, hor, out, a, s, fa);
#if SSTEM_GRAY16_ABLATE      // developer builds only (tools/build_ablate.sh x "-DSSTEM_GRAY16_ABLATE=1"): timing-only variants, wrong results above 1
    static const int var = [] { const char* e = getenv("SSTEM_GRAY16_VAR"); return e ? atoi(e) : 0; }();
    if constexpr (BLK) {
        switch (var) {
            case 1: return launch_gray16_v<4, 6, BLK, 1>(in, ver, hor, out, a, s, fa);
            case 2: return launch_gray16_v<4, 6, BLK, 2>(in, ver, hor, out, a, s, fa);
            case 4: return launch_gray16_v<4, 6, BLK, 4>(in, ver, hor, out, a, s, fa);
            case 8: return launch_gray16_v<4, 6, BLK, 8>(in, ver, hor, out, a, s, fa);
            case 16: return launch_gray16_v<4, 6, BLK, 16>(in, ver, hor, out, a, s, fa);
            case 6: return launch_gray16_v<4, 6, BLK, 6>(in, ver, hor, out, a, s, fa);
            case 22: return launch_gray16_v<4, 6, BLK, 22>(in, ver, hor, out, a, s, fa);
            default: break;
        }
    }
#endif
    return launch_gray16_v<4, 6, BLK>(in, ver, hor, out, a, s, fa);
}

// Single-plane spelling of the fused interpolation apply: the caller KNOWS its two frames are grayscale (it built the x3
// replication itself: inference_singleImage.py:55-61, test_fusion.py:105-106) and hands over the planes [B,1,H,W].  The
// trusted-gray kernel is launched directly: no channel comparison, no flag, no second build.  Same MFMA sequence on the
// same plane => the same bits as launch_interp_fused on the replicated frames.
bool interp_fused_gray_ok(int64_t H, int64_t W)
{
    return (uint64_t)F * (uint64_t)H * (uint64_t)W * 4u < (1ull << 32);   // one image's 51 planes behind a 32-bit buffer resource
}

hipError_t launch_interp_fused_gray(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                    const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                    hipStream_t s, uint8_t* out_u8)
{
    if (!interp_fused_gray_ok(H, W)) return hipErrorInvalidValue;
    TileArgs a = make_args(B, 3, H, W);
    a.in_planes = 1;
    const FusedArgs fa{g1, k1v, k1h, nullptr, out_u8};
    if (gray16_enabled() && !out_u8) return launch_gray16<false>(g2, k2v, k2h, out, a, s, fa);
    return launch_gray<2>(g2, k2v, k2h, out, a, s, fa);
}

// Blocked coefficients (include/sstem_sepconv.h): [B][H][tiles_x][51][64] fp32, tiles_x = ceil(W / 64); pixels beyond W are padding.
int64_t coef_blocked_floats(int64_t B, int64_t H, int64_t W) { return B * H * ((W + 63) / 64) * (int64_t)(F * 64); }

bool interp_fused_gray_blocked_ok(int64_t H, int64_t W)
{
    return (uint64_t)H * (uint64_t)((W + 63) / 64) * (uint64_t)(F * 256) < (1ull << 32);   // one image behind a 32-bit buffer resource
}

__global__ __launch_bounds__(256) void coef_nchw_to_blocked(const float* __restrict__ src, float* __restrict__ dst,
                                                            int64_t total, int H, int W, int tiles_x)
{
    // one thread per element of dst: (b, y, tx, f, l) <- src[b, f, y, tx*64 + l] (0 beyond W); reads and writes are 256-B runs
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        int64_t r = i >> 6;
        const int f = (int)(r % F); r /= F;
        const int tx = (int)(r % tiles_x); r /= tiles_x;
        const int y = (int)(r % H);
        const int64_t b = r / H;
        const int x = tx * 64 + l;
        dst[i] = x < W ? src[((b * F + f) * H + y) * (int64_t)W + x] : 0.f;
    }
}

hipError_t launch_coef_to_blocked(const float* src, float* dst, int64_t B, int64_t H, int64_t W, hipStream_t s)
{
    const int64_t total = coef_blocked_floats(B, H, W);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(coef_nchw_to_blocked, dim3(grid_1d(total, 256)), dim3(256), 0, s, src, dst, total, (int)H, (int)W,
                       (int)((W + 63) / 64));
    return hipGetLastError();
}

hipError_t launch_interp_fused_gray_blocked(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                            const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                            hipStream_t s, uint8_t* out_u8)
{
    if (!interp_fused_gray_blocked_ok(H, W)) return hipErrorInvalidValue;
    TileArgs a = make_args(B, 3, H, W);
    a.in_planes = 1;
    const FusedArgs fa{g1, k1v, k1h, nullptr, out_u8};
    if (out_u8) return launch_gray<2, true>(g2, k2v, k2h, out, a, s, fa);      // (the 16x16x4 forms have no uint8 store)
    if (gray16_enabled()) return launch_gray16<true>(g2, k2v, k2h, out, a, s, fa);
    return launch_gray<2, true>(g2, k2v, k2h, out, a, s, fa);
}

// Trusted-gray gradVertical launch; SSTEM_GRAY_GV_SHAPE: 0 = 4 waves x 8 rows (3 waves/SIMD), 1 = 4 x 16 with the
// B-operand prefetch (2 waves/SIMD); default as launch_gray.
template <int WAVES, int RPW, int WPE, bool PFH, int RING, bool BF = false>
static hipError_t launch_gray_gradv_v(const float* in, const float* g, const float* hor, float* gv, TileArgs a,
                                      hipStream_t s, const int* flag)
{
    constexpr int TR = WAVES * RPW;
    constexpr size_t lds_bytes = (size_t)(TR + F) * rm_pitch(1) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_gray_gradv_mfma<WAVES, RPW, WPE, PFH, RING, BF>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, g, hor, gv, a, flag);
    return hipGetLastError();
}

static hipError_t launch_gray_gradv(const float* in, const float* g, const float* hor, float* gv, const TileArgs& a,
                                    hipStream_t s, const int* flag)
{
    static const int forced = [] { const char* e = getenv("SSTEM_GRAY_GV_SHAPE"); return e ? atoi(e) : -1; }();
    int shape = forced;
    if (shape < 0) shape = (a.B * a.tiles_x * ((a.H + 63) / 64) >= 1024) ? 1 : 0;
    if (shape == 1) return launch_gray_gradv_v<4, 16, 2, true, 2>(in, g, hor, gv, a, s, flag);
    return launch_gray_gradv_v<4, 8, 3, false, 2>(in, g, hor, gv, a, s, flag);
}

// Trusted-gray gradHorizontal launch; SSTEM_GRAY_GH_SHAPE: 0 = 4 waves x 8 rows (3 waves/SIMD), 1 = 4 x 16 with the
// B-operand prefetch (2 waves/SIMD); 2, 3 = the same two with the results re-sorted in registers into whole-row stores.
template <int WAVES, int RPW, int WPE, bool PFH, int RING, bool COALESCE, bool BF = false>
static hipError_t launch_gray_gradh_v(const float* in, const float* g, const float* ver, float* gh, TileArgs a,
                                      hipStream_t s, const int* flag)
{
    constexpr int TR = WAVES * RPW;
    constexpr int ROWS = TR + F + 4;
    constexpr int PITCH_T = ((ROWS + 3) / 4 * 4) | 4;
    constexpr size_t lds_bytes = (size_t)TCOLS * PITCH_T * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    auto k = sepconv_gray_gradh_mfma<WAVES, RPW, WPE, PFH, RING, COALESCE, BF>;
    static std::atomic<uint64_t> lds_set{0};
    const hipError_t attr = set_lds(k, lds_bytes, lds_set);
    if (attr != hipSuccess) return attr;
    a.tiles_y = (a.H + TR - 1) / TR;
    const int64_t nwg = a.B * a.tiles_y * a.tiles_x;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WAVES * 64), lds_bytes, s, in, g, ver, gh, a, flag);
    return hipGetLastError();
}

static hipError_t launch_gray_gradh(const float* in, const float* g, const float* ver, float* gh, const TileArgs& a,
                                    hipStream_t s, const int* flag)
{
    static const int forced = [] { const char* e = getenv("SSTEM_GRAY_GH_SHAPE"); return e ? atoi(e) : -1; }();
    // measured at C2 on one box (profiles/r01/r_backward_kernels_same_box.txt): whole-row stores 829 us (2) / 814 us (3)
    // against 1029 us (0) / 1094 us (1) for the four-plane pieces; 0 and 1 stay as A/B knobs only
    int shape = forced;
    if (shape < 0) shape = (a.B * a.tiles_x * ((a.H + 63) / 64) >= 1024) ? 3 : 2;
    switch (shape) {
        case 1: return launch_gray_gradh_v<4, 16, 2, true, 2, false>(in, g, ver, gh, a, s, flag);
        case 2: return launch_gray_gradh_v<4, 8, 3, false, 2, true>(in, g, ver, gh, a, s, flag);
        case 3: return launch_gray_gradh_v<4, 16, 2, true, 2, true>(in, g, ver, gh, a, s, flag);
        default: return launch_gray_gradh_v<4, 8, 3, false, 2, false>(in, g, ver, gh, a, s, flag);
    }
}

hipError_t launch_bwd_mfma(const float* g, const float* in, const float* ver, const float* hor,
                           float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                           hipStream_t s)
{
    // C <= 3 (checked by the caller): a single channel chunk, c0 == 0.
    TileArgs a = make_args(B, C, H, W);
    hipError_t e = hipSuccess;
    int* flag = nullptr;
    if (C == 3 && gray_dispatch_enabled(H, W) && tile_rows(tile_variant()) == 32) {
        flag = next_gray_flag(s, e);
        if (e != hipSuccess) return e;
    }
    if (flag) {
        // gradVertical: detect -> generic build (returns at once on gray input) -> gray build (returns at once otherwise)
        e = launch_detect(in, nullptr, B, (H + F - 1) * (W + F - 1), flag, s);
        if (e != hipSuccess) return e;
        const FusedArgs fa{nullptr, nullptr, nullptr, flag};
        if (rgb_stream_enabled(H, W)) e = launch_rgb_stream<1>(in, g, hor, gv, a, s, fa);       // round 4: the streaming kernel's gradVertical
        else if (tile_variant() == 0) e = launch_rowmajor_v<1, 3, 8, 4>(in, g, hor, gv, a, s, fa);
        else e = launch_rowmajor_v<1, 3, 16, 2>(in, g, hor, gv, a, s, fa);
        if (e != hipSuccess) return e;
        e = launch_gray_gradv(in, g, hor, gv, a, s, flag);
        if (e != hipSuccess) return e;
        // gradHorizontal: same flag, same pair of launches
        if (rgb_stream_enabled(H, W) && gradh_coalesce()) e = launch_rgb_gradh_stream(in, g, ver, gh, a, s, flag);       // round 4
        else if (tile_variant() == 0) e = launch_gradh_v<3, 8, 4>(in, g, ver, gh, a, s, flag);
        else e = launch_gradh_v<3, 16, 2>(in, g, ver, gh, a, s, flag);
        if (e != hipSuccess) return e;
        return launch_gray_gradh(in, g, ver, gh, a, s, flag);
    }
    if (C == 3 && rgb_stream_enabled(H, W)) e = launch_rgb_stream<1>(in, g, hor, gv, a, s, FusedArgs{nullptr, nullptr, nullptr, nullptr});
    else if (C == 3) e = launch_rowmajor<1, 3>(in, g, hor, gv, a, s);
    else if (C == 2) e = launch_rowmajor<1, 2>(in, g, hor, gv, a, s);
    else e = launch_rowmajor<1, 1>(in, g, hor, gv, a, s);
    if (e != hipSuccess) return e;
    if (C == 3 && rgb_stream_enabled(H, W) && gradh_coalesce()) e = launch_rgb_gradh_stream(in, g, ver, gh, a, s, nullptr);
    else if (C == 3) e = launch_gradh<3>(in, g, ver, gh, a, s);
    else if (C == 2) e = launch_gradh<2>(in, g, ver, gh, a, s);
    else e = launch_gradh<1>(in, g, ver, gh, a, s);
    return e;
}

// ---- bf16 coefficient tensors (include/sstem_sepconv.h, the ..._bf16coef entry points) ----------------------------------------------
// vertical / horizontal are [B,51,H,W] bf16; everything else fp32.  x3-replicated grayscale frames (what every caller of the reference
// feeds) run on the trusted-gray streaming kernels, found by the same device-side dispatch as the fp32 entries; anything else runs on
// the direct kernels (one lane per element: correct for any C, slow -- the streaming three-channel kernels have no bf16 instance).
static bool gray_bf16_ok(int64_t C, int64_t H, int64_t W)
{
    return C == 3 && gray_dispatch_enabled(H, W) && tile_rows(tile_variant()) == 32;
}

hipError_t launch_fwd_bf16coef(const float* in, const uint16_t* ver, const uint16_t* hor, float* out,
                               int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s)
{
    const float* v = reinterpret_cast<const float*>(ver);
    const float* h = reinterpret_cast<const float*>(hor);
    hipError_t e = hipSuccess;
    int* flag = nullptr;
    if (gray_bf16_ok(C, H, W)) {
        flag = next_gray_flag(s, e);
        if (e != hipSuccess) return e;
    }
    if (flag) {
        e = launch_detect(in, nullptr, B, (H + F - 1) * (W + F - 1), flag, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sepconv_fwd_direct<true>, dim3(grid_1d(B * H * W, 256)), dim3(256), 0, s, in, v, h, out, B, C, H, W, F, (const int*)flag);
    e = hipGetLastError();
    if (e != hipSuccess || !flag) return e;
    TileArgs a = make_args(B, C, H, W);
    const FusedArgs fa{nullptr, nullptr, nullptr, flag};
    if (a.B * a.tiles_x * ((a.H + 31) / 32) < 512) return launch_gray_v<0, 4, 4, 3, false, 2, false, true>(in, v, h, out, a, s, fa);
    return launch_gray_v<0, 4, 8, 3, false, 2, false, true>(in, v, h, out, a, s, fa);
}

hipError_t launch_bwd_bf16coef(const float* g, const float* in, const uint16_t* ver, const uint16_t* hor,
                               float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s)
{
    const float* v = reinterpret_cast<const float*>(ver);
    const float* h = reinterpret_cast<const float*>(hor);
    hipError_t e = hipSuccess;
    int* flag = nullptr;
    if (gray_bf16_ok(C, H, W)) {
        flag = next_gray_flag(s, e);
        if (e != hipSuccess) return e;
    }
    if (flag) {
        e = launch_detect(in, nullptr, B, (H + F - 1) * (W + F - 1), flag, s);
        if (e != hipSuccess) return e;
    }
    const int grid = grid_1d(B * F * H * W, 256);
    hipLaunchKernelGGL((sepconv_grad_direct<true, true>), dim3(grid), dim3(256), 0, s, g, in, h, gv, B, C, H, W, F, (const int*)flag);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sepconv_grad_direct<false, true>), dim3(grid), dim3(256), 0, s, g, in, v, gh, B, C, H, W, F, (const int*)flag);
    e = hipGetLastError();
    if (e != hipSuccess || !flag) return e;
    TileArgs a = make_args(B, C, H, W);
    e = launch_gray_gradv_v<4, 8, 3, false, 2, true>(in, g, h, gv, a, s, flag);
    if (e != hipSuccess) return e;
    return launch_gray_gradh_v<4, 8, 3, false, 2, true, true>(in, g, v, gh, a, s, flag);
}

bool interp_fused_gray_bf16coef_ok(int64_t H, int64_t W) { return interp_fused_gray_ok(H, W); }

hipError_t launch_interp_fused_gray_bf16coef(const float* g1, const float* g2, const uint16_t* k1v, const uint16_t* k1h,
                                             const uint16_t* k2v, const uint16_t* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                             hipStream_t s)
{
    if (!interp_fused_gray_bf16coef_ok(H, W)) return hipErrorInvalidValue;
    TileArgs a = make_args(B, 3, H, W);
    a.in_planes = 1;
    const FusedArgs fa{g1, reinterpret_cast<const float*>(k1v), reinterpret_cast<const float*>(k1h), nullptr};
    const float* v2 = reinterpret_cast<const float*>(k2v);
    const float* h2 = reinterpret_cast<const float*>(k2h);
    if (a.B * a.tiles_x * ((a.H + 31) / 32) < 512) return launch_gray_v<2, 4, 4, 3, false, 2, false, true>(g2, v2, h2, out, a, s, fa);
    return launch_gray_v<2, 4, 8, 3, false, 2, false, true>(g2, v2, h2, out, a, s, fa);
}

}  // namespace sstem
