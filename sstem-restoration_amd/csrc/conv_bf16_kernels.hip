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
//       ds_read_b128 at bin_off(row, pixel, h): 16 consecutive 16-byte slots per lane group (conflict-free: BIN_PITCH below), with the tap as an immediate
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
// the tile's LDS image: [channel half (8 channels = 16 B)][tile row][column] in 16-byte slots, rows BIN_PITCH = 35 slots apart (odd) -- the
// conflict-free image of conv_split_kernels.hip (round 4: the former [pixel][16 channels] image spent 2/3 of the LDS cycles in bank conflicts)
constexpr int BIN_PITCH = BIN_PW | 1;
constexpr int BIN_HOFF = BIN_R * BIN_PITCH * 16;      // bytes from channel half 0 to half 1
constexpr int BIN_BYTES = 2 * BIN_HOFF;               // 11200 B per buffer
__device__ __forceinline__ constexpr int bin_off(int row, int col, int half) { return (row * BIN_PITCH + col) * 16 + half * BIN_HOFF; }
constexpr uint32_t OOB = 0x80000000u;         // per-lane offset of a padding lane: beyond any buffer this kernel accepts

// "wave-uniform 64-bit base (SGPR pair) + one 32-bit per-lane byte offset" stores: the saddr form, no per-lane 64-bit addresses
typedef __attribute__((address_space(1))) float gfloat_t;
template <typename T>
__device__ __forceinline__ void pin_uniform_ptr(T*& p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ void store_lane(float* ubase, uint32_t lane_byte_off, float v)
{
    *reinterpret_cast<gfloat_t*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off) = v;
}

typedef __attribute__((address_space(1))) __bf16 gbf16_t;
typedef __attribute__((address_space(1))) uint8_t gbyte_bf_t;
__device__ __forceinline__ void store_lane_b16(__bf16* ubase, uint32_t lane_byte_off, float v)
{
    *reinterpret_cast<gbf16_t*>(reinterpret_cast<uint64_t>(ubase) + lane_byte_off) = (__bf16)v;
}

__device__ __forceinline__ void pin_sgpr(uint32_t& v) { asm volatile("" : "+s"(v)); }

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

// Both packings of one layer's weights in one launch (see pack_weights_3x3_both in conv_kernels.hip)
__global__ void pack_weights_3x3_bf16_both(const float* __restrict__ w, __bf16* __restrict__ wp_f, __bf16* __restrict__ wp_t, int Cin,
                                           int Cout, int CO_f, int nchunks_f, int64_t n_fwd, int CO_t, int nchunks_t, int64_t n_t)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const bool t = i >= n_fwd;
        const int64_t idx = t ? i - n_fwd : i;
        const int CO = t ? CO_t : CO_f, nchunks = t ? nchunks_t : nchunks_f;
        const int cin = t ? Cout : Cin, cout = t ? Cin : Cout;
        const int cl = idx % BKC;
        int64_t r = idx / BKC;
        const int col = r % CO; r /= CO;
        const int tap = r % 9; r /= 9;
        const int chunk = r % nchunks;
        const int cb = r / nchunks;
        const int ci = chunk * BKC + cl, co = cb * CO + col;
        float v = 0.f;
        if (ci < cin && co < cout) v = t ? w[((int64_t)ci * cout + co) * 9 + (8 - tap)] : w[((int64_t)co * cin + ci) * 9 + tap];
        (t ? wp_t : wp_f)[idx] = (__bf16)v;
    }
}

// Both packings of MANY layers in one launch (see pack_weights_3x3_group in conv_kernels.hip; same table layout, entries from
// pack_group_entry_bf16)
__global__ __launch_bounds__(256) void pack_weights_3x3_bf16_group(const int64_t* __restrict__ table, int n_entries)
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
    const bool t = i >= n_fwd;
    const int64_t idx = t ? i - n_fwd : i;
    const int CO = (int)(t ? en[9] : en[5]), nchunks = (int)(t ? en[10] : en[6]);
    const int cin = t ? Cout : Cin, cout = t ? Cin : Cout;
    const int cl = idx % BKC;
    int64_t r = idx / BKC;
    const int col = r % CO; r /= CO;
    const int tap = r % 9; r /= 9;
    const int chunk = r % nchunks;
    const int cb = r / nchunks;
    const int ci = chunk * BKC + cl, co = cb * CO + col;
    float v = 0.f;
    if (ci < cin && co < cout) v = t ? w[((int64_t)ci * cout + co) * 9 + (8 - tap)] : w[((int64_t)co * cin + ci) * 9 + tap];
    (t ? wp_t : wp_f)[idx] = (__bf16)v;
}

// INB / OUTB (16-byte staging only): the input / output TENSOR is bf16 NCHW instead of fp32 -- what the convolutions inside one
// Conv-ReLU-Conv block exchange under the bf16 id when no backward can follow.  Numerically free: the consumer rounds the same
// fp32 value to bf16 with the same instruction.
// MASKED (16-byte staging, fp32 input): in_mask (nullable, [N,Cin,H,W] bytes) zeroes the input elements whose byte is 0 while they are
// staged -- the data gradient of a Conv+ReLU layer without a select pass over the incoming gradient; out_mask (nullable, [N,Cout,H,W]
// bytes) receives (stored output > 0), the mask the forward launch writes instead of a compare pass (sstem_conv3x3_forward_bf16io_masked).
template <int WCO, int WR, int WPE, bool VEC, bool INB = false, bool OUTB = false, bool MASKED = false>
__global__ __launch_bounds__(256, WPE) void conv3x3_bf16_mfma(
    const void* __restrict__ in_v, const __bf16* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, void* __restrict__ out_v,
    int N, int Cin, int H, int W, int Cout, int nchunks, int ncb, int act, float slope, int ksplit, float* __restrict__ slab,
    int xcd_remap, const uint8_t* __restrict__ in_mask = nullptr, uint8_t* __restrict__ out_mask = nullptr)
{
    static_assert(!MASKED || VEC, "masks: 16-byte staging (in_mask: of an fp32 input; the launcher refuses it with a bf16 one)");
    static_assert(WCO * WR == 4, "four waves");
    static_assert(VEC || !INB, "a bf16 input tensor needs the 16-byte staging path");
    constexpr int CO = 32 * WCO, R = BTH / WR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BIN_BYTES];
    const float* in = static_cast<const float*>(in_v);          // fp32 view (the dword path, and the VEC path when !INB)
    float* out = static_cast<float*>(out_v);
    __bf16* outb = static_cast<__bf16*>(out_v);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int wco = wave % WCO, wr = wave / WCO;
    // XCD-aware tile order.  Workgroups are handed to the 8 XCDs round-robin in launch order (x fastest), and each XCD has its own
    // L2: with the plain order horizontally adjacent tiles -- which share the cache lines of their halo columns -- always sit on
    // different XCDs, and every tile fetched three 128-B lines per input row and channel where 1.25 carry its data (PMC: 1572 MB
    // fetched for a 537 MB input on 8 x 64->64 at 512^2).  Re-mapped, XCD k owns a contiguous run of the linear tile order
    // (x fastest, then y, then image / channel block), so a tile's left and right neighbours run next to it on the same L2.
    // The output-channel blocks of one pixel tile are neighbours in that order too (they read the same input tile: the second one
    // finds it in L2), then x, y, K slice, image.
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
    const int X0 = bx * BTW, Y0 = by * BTH;
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
        lds_off[k] = px < BIN_PX ? bin_off(row, col, half) : -1;
        half_of[k] = half;
    }

    float stg[3][8];
    auto issue_in = [&](int chunk) {
        const int cl_lim = Cin - chunk * BKC;                    // channels left from this chunk on (uniform)
        const uint32_t sbase = (uint32_t)(chunk * BKC) * plane4;
        if (cl_lim >= BKC) {                                     // whole chunk: 24 loads back to back, no branches
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t so = sbase + (uint32_t)(half_of[k] * 8) * plane4;      // one running scalar offset, not 24 live ones
                pin_sgpr(so);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    stg[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, (int)voff[k], (int)so, 0));
                    so += plane4;
                    pin_sgpr(so);
                }
            }
        } else {                                                 // the last chunk of a ragged channel count
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = half_of[k] * 8 + i;
                    float v = 0.f;
                    if (c < cl_lim)
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

    // ---- VEC staging (W % 4 == 0, 16-B aligned input -- every real layer): one lane loads 4 pixels of a row (16 B) from each
    // of the 8 channels of its wave's half, and stores one 16-B pixel slot per pixel.  With one dword per lane the kernel was
    // bound by the rate at which a CU's address unit takes wave-instructions (96 per chunk for 21.8 KB; 32 here).
    //   waves 0, 1 (half = wave & 1): lane = (row 0..7, group 0..7) of the 32 interior columns;
    //   waves 2, 3: lanes 0..15 = rows 8, 9; lanes 16..35 = the two halo columns of the 10 rows, taken from the ALIGNED group that
    //   holds them (left halo x = X0 - 1 = last pixel of group X0 - 4; right halo x = X0 + 32 = first pixel of its group).
    const int vhalf = wave & 1;
    uint32_t vvoff = OOB;
    int vdst[4] = {-1, -1, -1, -1};      // LDS byte offset of the pixel slot of the lane's pixel j, or -1
    {
        int row = -1, xg = 0, first_col = 0, only = -1;                  // only: halo lanes keep a single pixel of the group
        if (wave < 2) {                 // eight consecutive lanes (one ds_write_b128 group) = 4 rows x 2 column groups: all eight slot residues
            const int q = 2 * (lane >> 4) + (lane & 1);
            row = 4 * ((lane >> 3) & 1) + ((lane & 7) >> 1); xg = X0 + 4 * q; first_col = 1 + 4 * q;
        }
        else if (lane < 16) { row = 8 + (lane >> 3); xg = X0 + 4 * (lane & 7); first_col = 1 + 4 * (lane & 7); }
        else if (lane < 36) {
            const int hl = lane - 16; row = hl >> 1;
            if (hl & 1) { xg = X0 + BTW; first_col = BIN_PW - 1; only = 0; }
            else { xg = X0 - 4; first_col = 0 - 3; only = 3; }
        }
        if (row >= 0) {
            const int y = Y0 - 1 + row;
            if (y >= 0 && y < H && xg >= 0 && xg < W) vvoff = (uint32_t)(y * W + xg) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (only < 0 || only == j) vdst[j] = bin_off(row, first_col + j, vhalf);
        }
    }
    // (plain 16-B global loads from a uniform channel base plus the lane's offset, clamped to 0 and masked at the LDS store for
    // padding lanes: clang 19's __builtin_amdgcn_raw_buffer_load_b128 lowers to a ONE-dword load splat over the vector.)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const bool vok = vvoff != OOB;
    const uint32_t vsafe = vok ? vvoff : 0u;
    constexpr int ESZ = INB ? 2 : 4;                                   // bytes per input element
    const uint32_t planeB = (uint32_t)plane * ESZ;
    const uint32_t vsafeB = INB ? vsafe / 2u : vsafe;
    const char* in_n = static_cast<const char*>(in_v) + (int64_t)n * Cin * plane * ESZ;
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    f32x4v stg4[INB ? 1 : 8];
    u32x2v stg2[INB ? 8 : 1];                                         // INB: 4 bf16 pixels of one channel = 8 B
    uint32_t mk4[(MASKED && !INB) ? 8 : 1];                           // MASKED: the mask bytes of the lane's four pixels, per channel
    auto issue_in_v = [&](int chunk) {
        const int cl_lim = Cin - chunk * BKC;
        const char* pc = in_n + (int64_t)(chunk * BKC + vhalf * 8) * planeB;      // uniform
        if constexpr (MASKED && !INB) {
            const uint8_t* pm = in_mask + ((int64_t)n * Cin + chunk * BKC + vhalf * 8) * plane;          // uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool chan = in_mask != nullptr && (cl_lim >= BKC || vhalf * 8 + i < cl_lim);
                uint32_t m = 0x01010101u;
                if (chan) m = *reinterpret_cast<const uint32_t*>(pm + (int64_t)i * plane + (vsafe >> 2));
                mk4[i] = m;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool chan = cl_lim >= BKC || vhalf * 8 + i < cl_lim;    // uniform: the last chunk of a ragged channel count
            if constexpr (INB) {
                u32x2v v = {0u, 0u};
                if (chan) v = *reinterpret_cast<const u32x2v*>(pc + (int64_t)i * planeB + vsafeB);
                stg2[i] = v;
            } else {
                f32x4v v = {0.f, 0.f, 0.f, 0.f};
                if (chan) v = *reinterpret_cast<const f32x4v*>(pc + (int64_t)i * planeB + vsafeB);
                stg4[i] = v;
            }
        }
    };
    // interior tiles (uniform): every lane that stores loaded from inside the image -- no selects (32 per chunk otherwise)
    const bool tile_interior = Y0 >= 1 && Y0 + BTH + 1 <= H && X0 >= 4 && X0 + BTW + 4 <= W;
    auto commit_in_v = [&](int buf) {
        if constexpr (INB) {
            // pixel j of channels 2m, 2m+1 -> one dword: the low (j even) or high (j odd) halves of dword j >> 1 of both channels
            typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
            const bool keep = tile_interior || vok;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4p pk;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t d = __builtin_amdgcn_perm(stg2[2 * m + 1][j >> 1], stg2[2 * m][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
                    pk[m] = keep ? d : 0u;
                }
                if (vdst[j] >= 0) *reinterpret_cast<u32x4p*>(lds + buf * BIN_BYTES + vdst[j]) = pk;
            }
            return;
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = (__bf16)((vok && ((mk4[i] >> (8 * j)) & 0xffu) != 0u) ? stg4[i][j] : 0.f);
                if (vdst[j] >= 0) *reinterpret_cast<bf16x8*>(lds + buf * BIN_BYTES + vdst[j]) = pk;
            }
            return;
        }
        if (tile_interior) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = (__bf16)stg4[i][j];
                if (vdst[j] >= 0) *reinterpret_cast<bf16x8*>(lds + buf * BIN_BYTES + vdst[j]) = pk;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = (__bf16)(vok ? stg4[i][j] : 0.f);
                if (vdst[j] >= 0) *reinterpret_cast<bf16x8*>(lds + buf * BIN_BYTES + vdst[j]) = pk;
            }
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

    const int b_lane = bin_off(wr * R, r, h);
    auto mfmas = [&](const bf16x8 (&a)[9], int buf) {
        const unsigned char* bp = lds + buf * BIN_BYTES + b_lane;
        bf16x8 b[2][3];                                          // fragments of input row ro / ro + 1 (read one row ahead)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) b[0][kx] = *reinterpret_cast<const bf16x8*>(bp + kx * 16);
#pragma unroll
        for (int ro = 0; ro < R + 2; ++ro) {                     // input row of this wave's row group
            if (ro + 1 < R + 2) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    b[(ro + 1) & 1][kx] = *reinterpret_cast<const bf16x8*>(bp + ((ro + 1) * BIN_PITCH + kx) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);                   // keep the reads of row ro + 1 ahead of the MFMAs of row ro
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int rr = ro - ky;
                    if (rr >= 0 && rr < R)
                        acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky * 3 + kx], b[ro & 1][kx], acc[rr], 0, 0, 0);
                }
            }
        }
    };

    bf16x8 a0[9], a1[9];
    if constexpr (VEC) issue_in_v(c_first); else issue_in(c_first);
    load_a(a0, c_first);
    if constexpr (VEC) commit_in_v(0); else commit_in(0);
    __syncthreads();

    auto body = [&](int c, const bf16x8 (&acur)[9], bf16x8 (&anxt)[9]) {
        const bool more = (c + 1 < c_end);
        const int buf = (c - c_first) & 1;
        // acur was loaded a chunk ago: settle it BEFORE this chunk's loads are issued.  (Left to the compiler, the MFMAs
        // below waited with vmcnt(8)..(0) -- for the loads just issued -- on every other chunk.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        if (more) { if constexpr (VEC) issue_in_v(c + 1); else issue_in(c + 1); load_a(anxt, c + 1); }
        mfmas(acur, buf);
        if (more) { if constexpr (VEC) commit_in_v(buf ^ 1); else commit_in(buf ^ 1); }
        __syncthreads();
    };
    for (int c = c_first; c < c_end; c += 2) {
        body(c, a0, a1);
        if (c + 1 < c_end) body(c + 1, a1, a0);
    }

    // ---- epilogue: acc[rr][q] = out[co = cb*CO + wco*32 + (q&3) + 8*(q>>2) + 4*h][y = Y0 + wr*R + rr][x = X0 + r]
    // Tiles that lie wholly inside the image and the channel range take the lean path: one per-lane byte offset, uniform
    // (scalar) bases per (q, row), the per-channel constants loaded up front, the activation resolved once per workgroup.
    // (The generic path below costs ~35 instructions per stored element -- per-element bounds, a switch on the activation, 64-bit
    // address arithmetic --, ~2200 per wave against 144 MFMAs on a 64-channel layer: 21 % of the wave-cycles were instruction
    // issue and the MFMA pipe 14 % busy, profiles/r01/w_conv_bf16.txt.)
    const int x = X0 + r;
    const bool whole = Y0 + BTH <= H && X0 + BTW <= W && (int64_t)Cout * plane * 4 < ((int64_t)1 << 32);
    const bool cpart = cb * CO + CO > Cout;                // uniform: the last, partial channel block (51 of 64 in the kernel heads)
    if (whole) {
        const int co0 = cb * CO + wco * 32;                                            // uniform
        const uint32_t lane_off = (uint32_t)(4 * h) * plane4 + (uint32_t)((Y0 + wr * R) * W + x) * 4u;
        if (ksplit > 1) {
            float* base = slab + (((int64_t)ks * N + n) * Cout + co0) * plane;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bool live = !(cpart && co0 + (q & 3) + 8 * (q >> 2) + 4 * h >= Cout);      // per lane
                float* chp = base + (int64_t)((q & 3) + 8 * (q >> 2)) * plane;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    float* rp = chp + rr * W;
                    pin_uniform_ptr(rp);                                                          // outside the divergent store
                    if (live) store_lane(rp, lane_off, acc[rr][q]);
                }
            }
            return;
        }
        float* base = out + ((int64_t)n * Cout + co0) * plane;
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
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    float v = acc[rr][q] + bs[q];
                    v = actf(v * sc[q] + sh[q]);
                    if constexpr (OUTB) v = (float)(__bf16)v;                                 // the value that is stored (and masked on)
                    if constexpr (MASKED) {
                        if (out_mask) {
                            uint8_t* mp = out_mask + ((int64_t)n * Cout + co0 + (q & 3) + 8 * (q >> 2)) * plane + rr * W;
                            pin_uniform_ptr(mp);
                            if (live) *reinterpret_cast<gbyte_bf_t*>(reinterpret_cast<uint64_t>(mp) + (lane_off >> 2)) = v > 0.f ? 1 : 0;
                        }
                    }
                    if constexpr (OUTB) {
                        __bf16* rpb = outb + ((int64_t)n * Cout + co0 + (q & 3) + 8 * (q >> 2)) * plane + rr * W;
                        pin_uniform_ptr(rpb);
                        if (live) store_lane_b16(rpb, lane_off >> 1, v);
                    } else {
                        float* rp = chp + rr * W;
                        pin_uniform_ptr(rp);                                                  // outside the divergent store
                        if (live) store_lane(rp, lane_off, v);
                    }
                }
            }
        };
        if (act == 1) store_all([](float v) { return v > 0.f ? v : 0.f; });
        else if (act == 2) store_all([slope](float v) { return v > 0.f ? v : v * slope; });
        else store_all([](float v) { return v; });
        return;
    }
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
                v = act_bf(v * sc + sh, act, slope);
                if constexpr (OUTB) v = (float)(__bf16)v;
                if constexpr (MASKED) { if (out_mask) out_mask[((int64_t)n * Cout + co) * plane + (int64_t)y * W + x] = v > 0.f ? 1 : 0; }
                if constexpr (OUTB) outb[((int64_t)n * Cout + co) * plane + (int64_t)y * W + x] = (__bf16)v;
                else out[((int64_t)n * Cout + co) * plane + (int64_t)y * W + x] = v;
            }
        }
    }
}

// Sum of the K slices in ascending order + the fused epilogue (same arithmetic as conv3x3_splitk_epilogue of the fp32 path).
__global__ __launch_bounds__(256) void conv3x3_bf16_splitk_epilogue(
    const float* __restrict__ slab, const float* __restrict__ bias, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int64_t total, int64_t plane, int Cout, int ksplit,
    int act, float slope, uint8_t* __restrict__ out_mask = nullptr)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
        const int co = (int)((i / plane) % Cout);
        v += bias ? bias[co] : 0.f;
        v = act_bf(v * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f), act, slope);
        if (out_mask) out_mask[i] = v > 0.f ? 1 : 0;
        out[i] = v;
    }
}

// ---- 3x3 weight gradient with bf16 operands -----------------------------------------------------------------------------
// gW[co][ci][tap] = sum over pixels of g[co][p] * in[ci][p + tap]: M = co, N = ci, K = pixels (the geometry, the split over
// pixel tiles, the slab layout and the fixed-order reduce kernel are those of conv3x3_wgrad_mfma in conv_kernels.hip; the bias
// gradient is summed from the fp32 values, not from the rounded ones).
// v_mfma_f32_16x16x32_bf16: K = 32 = one whole row of the 2-row x 32-column pixel tile; lane (r = lane & 15, q = lane >> 4)
// supplies pixels 8q .. 8q+7 of its channel, so both tiles sit in LDS as bf16 rows:
//   g_t [2 rows][64 co]  pitch  80 B: 32 pixels + 16 B pad   (80 = 5 x 16: the b128 reads of 16 consecutive channels hit 16 slots)
//   i_t [4 rows][64 ci]  pitch 112 B: tile column cc (x = X0 - 1 + cc) at element 7 + cc, so x = X0 starts a 16-B chunk
// A tap's kx shift is one element = 2 B: the shifted B fragments are assembled in registers from the aligned chunk and its two
// neighbour dwords (five v_alignbit per input row); the ky shift is another LDS row.
// Workgroup = 8 waves = 64 co x 64 ci: wave (wi, wj) owns 32 co x 16 ci = 2 x 9 accumulator tiles of 4 registers (72 VGPRs --
// the 4-wave 32x32 form of the fp32 kernel needs 144 plus 64 registers of loads in flight and spilled here).
constexpr int WG_P = 80, WI_P = 112;
constexpr int WG_BYTES = 2 * 64 * WG_P, WI_BYTES = 4 * 64 * WI_P;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VEC (W % 4 == 0, 16-B aligned tensors -- every real layer): the tiles are staged with 16-B loads (4 pixels of a row per lane;
// the two halo columns of the input tile as single dwords).  With one dword per lane the kernel was bound by the rate at which a
// CU's address unit takes wave-instructions: 256 of them per tile for 51 KB (about 16 B per clock per CU) against 56 here.
// INB (VEC only): the saved input tensor is bf16 NCHW (a block run as one autograd function keeps the tensors between its
// convolutions in bf16): 8-byte loads go to LDS as they are.
// MASKED (VEC): g_mask (nullable, [N,Cout,H,W] bytes): g counts as 0 where the byte is 0 (the ReLU of the layer's output)
template <bool VEC, bool INB = false, bool MASKED = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_bf16_mfma(
    const void* __restrict__ in_v, const float* __restrict__ g, float* __restrict__ slab,
    int N, int Cin, int H, int W, int Cout, int CinP, int CoutP, int ksplit, int tiles_x, int tiles_y,
    float* __restrict__ bias_slab, int run_tiles, const uint8_t* __restrict__ g_mask = nullptr)
{
    static_assert(!MASKED || VEC, "masks: 16-byte staging");
    static_assert(VEC || !INB, "a bf16 input tensor needs the 16-byte staging path");
    // both tiles are double-buffered: tile t goes to buffer t & 1, so the LDS stores of tile t+1 need not wait until every wave has
    // finished reading tile t -- ONE barrier per tile (behind the stores) instead of two, and the waves may drift apart by a phase
    __shared__ __attribute__((aligned(16))) unsigned char g_t2[2 * WG_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char i_t2[2 * WI_BYTES];
    unsigned char* g_t = g_t2;
    unsigned char* i_t = i_t2;
    const float* in = static_cast<const float*>(in_v);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r = lane & 15;
    const int wi = wave >> 2, wj = wave & 3;
    const int nib = CinP / 64;
    // Workgroup -> (channel block, pixel-tile slice).  With run_tiles the slice is a contiguous RUN of tiles (x fastest), so the halo
    // columns and rows of a tile are fetched again by the same workgroup a step later, and the linear workgroup id is re-mapped so
    // that one XCD (= one L2; workgroups are dealt to the 8 XCDs round-robin) owns neighbouring runs.
    uint32_t wgid = blockIdx.x;
    if (run_tiles) {
        const uint32_t total = gridDim.x, k8 = wgid & 7u, q8 = total >> 3, r8 = total & 7u;
        wgid = k8 * q8 + (k8 < r8 ? k8 : r8) + (wgid >> 3);
    }
    const int blk = (int)(wgid / (uint32_t)ksplit), ks = (int)(wgid % (uint32_t)ksplit);
    const int cb = blk / nib, ib = blk % nib;
    const int64_t plane = (int64_t)H * W;
    const int ntiles = N * tiles_y * tiles_x;
    const bool do_bias = (bias_slab != nullptr) && (ib == 0);

    f32x4 acc[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][t][e] = 0.f;

    // flat mapping of a channel's 4 x 34 input tile: e = j*64 + lane (conv3x3_wgrad_mfma), row/column fixed per lane
    constexpr int I_E = 4 * BIN_PW, I_J = 3, CH_W = 8;      // 8 channels of each tile per wave
    int er[I_J], ec[I_J];
#pragma unroll
    for (int j = 0; j < I_J; ++j) { const int e = j * 64 + lane; er[j] = e / BIN_PW; ec[j] = e - er[j] * BIN_PW; }
    float gv[CH_W], ivp[CH_W * I_J], bsum[CH_W];
#pragma unroll
    for (int k = 0; k < CH_W; ++k) bsum[k] = 0.f;

    // run_tiles == 2: the linear tile order runs DOWN a 32-pixel column strip first (then the next strip, then the next image): two
    // consecutive tiles share two of their four input rows, which the same CU has just fetched -- with the x-fastest order the
    // kernel fetched every input row twice from beyond L2 and ran at the streaming limit on it (1156 MB for 537 MB of tensors)
    auto geometry = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        if (run_tiles == 2) {
            const int ty = tile % tiles_y;
            const int r0 = tile / tiles_y;
            n = r0 / tiles_x; X0 = (r0 % tiles_x) * BTW; Y0 = ty * 2;
        } else {
            const int tx = tile % tiles_x;
            const int r0 = tile / tiles_x;
            n = r0 / tiles_y; X0 = tx * BTW; Y0 = (r0 % tiles_y) * 2;
        }
    };
    auto lane_offsets = [&](int X0, int Y0, uint32_t (&off)[I_J], bool (&ok)[I_J]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < I_J; ++j) {
            const int yi = Y0 - 1 + er[j], xi = X0 - 1 + ec[j];
            ok[j] = (j * 64 + lane < I_E) && yi >= 0 && yi < H && xi >= 0 && xi < W;
            off[j] = ok[j] ? (uint32_t)(yi * W + xi) * 4u : 0u;
        }
    };
    auto issue = [&](int tile) __attribute__((always_inline)) {          // loads only (clamped addresses)
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
        const uint32_t poff = (yy < H && xx < W) ? (uint32_t)(yy * W + xx) * 4u : 0u;
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            const int co = cb * 64 + wave + 8 * k;
            const float* base = g + ((int64_t)n * Cout + (co < Cout ? co : 0)) * plane;
            gv[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + poff);
        }
        uint32_t off[I_J]; bool ok[I_J];
        lane_offsets(X0, Y0, off, ok);
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            const int ci = ib * 64 + wave + 8 * k;
            const float* base = in + ((int64_t)n * Cin + (ci < Cin ? ci : 0)) * plane;
#pragma unroll
            for (int j = 0; j < I_J; ++j)
                ivp[k * I_J + j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off[j]);
        }
    };
    auto commit = [&](int tile) __attribute__((always_inline)) {         // masks, rounding, LDS stores of what issue(tile) loaded
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
        const bool pix_ok = yy < H && xx < W;
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            const int c = wave + 8 * k;
            const float v = (pix_ok && cb * 64 + c < Cout) ? gv[k] : 0.f;
            bsum[k] += v;
            *reinterpret_cast<__bf16*>(g_t + ((lane >> 5) * 64 + c) * WG_P + (lane & 31) * 2) = (__bf16)v;
        }
        uint32_t off[I_J]; bool ok[I_J];
        lane_offsets(X0, Y0, off, ok);
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            const int c = wave + 8 * k;
            const bool ch_ok = ib * 64 + c < Cin;
#pragma unroll
            for (int j = 0; j < I_J; ++j)
                if (j * 64 + lane < I_E)
                    *reinterpret_cast<__bf16*>(i_t + (er[j] * 64 + c) * WI_P + (7 + ec[j]) * 2) = (__bf16)((ch_ok && ok[j]) ? ivp[k * I_J + j] : 0.f);
        }
    };

    // ---- VEC staging: items of 4 pixels.  g: 64 channels x 2 rows x 8 groups = 2 items per thread; input interior: 64 x 4 rows x 8
    // = 4 items per thread; input halo columns: 64 x 4 rows x 2 = one dword per thread.  Channel, row and group of an item are
    // fixed per thread; only the tile origin changes.
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const uint32_t plane4 = (uint32_t)plane * 4u;
    uint32_t vg_off[2], vi_off[4], vh_off;
    int vg_lds[2], vi_lds[4], vh_lds;
    bool vg_ch[2], vi_ch[4], vh_ch;
    constexpr uint32_t IE = INB ? 2u : 4u;                       // bytes per input element
    typedef uint32_t u32x2i __attribute__((ext_vector_type(2)));
    f32x4 gq[2], iq[INB ? 1 : 4];
    uint32_t mq[MASKED ? 2 : 1];
    u32x2i iqb[INB ? 4 : 1];
    float hq = 0.f, bsum2[2] = {0.f, 0.f};
    uint16_t hqb = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = tid + 512 * k, ch = item >> 4, row = (item >> 3) & 1, grp = item & 7;
        vg_ch[k] = cb * 64 + ch < Cout;
        vg_off[k] = (uint32_t)ch * plane4 + (uint32_t)(row * W + 4 * grp) * 4u;
        vg_lds[k] = (row * 64 + ch) * WG_P + grp * 8;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int item = tid + 512 * k, ch = item >> 5, row = (item >> 3) & 3, grp = item & 7;
        vi_ch[k] = ib * 64 + ch < Cin;
        vi_off[k] = ((uint32_t)ch * (uint32_t)plane + (uint32_t)(row * W + 4 * grp)) * IE;
        vi_lds[k] = (row * 64 + ch) * WI_P + 16 + grp * 8;
    }
    {
        const int ch = tid >> 3, row = (tid >> 1) & 3, side = tid & 1;
        vh_ch = ib * 64 + ch < Cin;
        vh_off = ((uint32_t)ch * (uint32_t)plane + (uint32_t)(row * W)) * IE;
        vh_lds = (row * 64 + ch) * WI_P + (side ? 40 : 7) * 2;
    }
    auto vec_ok = [&](int X0, int Y0, bool (&gk)[2], bool (&ik)[4], bool& hk, int& hx) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 512 * k, row = (item >> 3) & 1, grp = item & 7;
            gk[k] = vg_ch[k] && Y0 + row < H && X0 + 4 * grp < W;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = tid + 512 * k, row = (item >> 3) & 3, grp = item & 7;
            const int y = Y0 - 1 + row;
            ik[k] = vi_ch[k] && y >= 0 && y < H && X0 + 4 * grp < W;
        }
        const int y = Y0 - 1 + ((tid >> 1) & 3);
        hx = (tid & 1) ? X0 + BTW : X0 - 1;
        hk = vh_ch && y >= 0 && y < H && hx >= 0 && hx < W;
    };
    auto issue_v = [&](int tile) __attribute__((always_inline)) {
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        bool gk[2], ik[4], hk; int hx;
        vec_ok(X0, Y0, gk, ik, hk, hx);
        const char* gbase = reinterpret_cast<const char*>(g + ((int64_t)n * Cout + cb * 64) * plane);     // uniform
        const char* ibase = static_cast<const char*>(in_v) + ((int64_t)n * Cin + ib * 64) * plane * IE;
        const uint32_t tg = (uint32_t)(Y0 * W + X0) * 4u, ti = (uint32_t)((Y0 - 1) * W + X0) * IE;          // ti may wrap: rows >= 1 undo it
#pragma unroll
        for (int k = 0; k < 2; ++k) gq[k] = *reinterpret_cast<const f32x4*>(gbase + (gk[k] ? vg_off[k] + tg : 0u));
        if constexpr (MASKED) {
            const uint8_t* mbase = g_mask + ((int64_t)n * Cout + cb * 64) * plane;                         // uniform
#pragma unroll
            for (int k = 0; k < 2; ++k)
                mq[k] = g_mask ? *reinterpret_cast<const uint32_t*>(mbase + (gk[k] ? (vg_off[k] + tg) >> 2 : 0u)) : 0x01010101u;
        }
        if constexpr (INB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) iqb[k] = *reinterpret_cast<const u32x2i*>(ibase + (ik[k] ? vi_off[k] + ti : 0u));
            hqb = *reinterpret_cast<const uint16_t*>(ibase + (hk ? vh_off + (uint32_t)((Y0 - 1) * W + hx) * IE : 0u));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) iq[k] = *reinterpret_cast<const f32x4*>(ibase + (ik[k] ? vi_off[k] + ti : 0u));
            hq = *reinterpret_cast<const float*>(ibase + (hk ? vh_off + (uint32_t)((Y0 - 1) * W + hx) * IE : 0u));
        }
    };
    auto commit_v = [&](int tile) __attribute__((always_inline)) {
        int n, X0, Y0;
        geometry(tile, n, X0, Y0);
        bool gk[2], ik[4], hk; int hx;
        vec_ok(X0, Y0, gk, ik, hk, hx);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x4 pk;
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = gk[k] ? gq[k][e] : 0.f;
                if constexpr (MASKED) v = ((mq[k] >> (8 * e)) & 0xffu) == 0u ? 0.f : v;
                sum += v; pk[e] = (__bf16)v;
            }
            bsum2[k] += sum;
            *reinterpret_cast<bf16x4*>(g_t + vg_lds[k]) = pk;
        }
        if constexpr (INB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32x2i v = iqb[k];
                if (!ik[k]) { v[0] = 0u; v[1] = 0u; }
                *reinterpret_cast<u32x2i*>(i_t + vi_lds[k]) = v;
            }
            *reinterpret_cast<uint16_t*>(i_t + vh_lds) = hk ? hqb : (uint16_t)0;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(ik[k] ? iq[k][e] : 0.f);
                *reinterpret_cast<bf16x4*>(i_t + vi_lds[k]) = pk;
            }
            *reinterpret_cast<__bf16*>(i_t + vh_lds) = (__bf16)(hk ? hq : 0.f);
        }
    };

    const int ap_lane = (wi * 32 + r) * WG_P + q4 * 16;
    const int bp_lane = (wj * 16 + r) * WI_P + 16 + q4 * 16;
    const int tpw = (ntiles + ksplit - 1) / ksplit;
    const int t_first = run_tiles ? ks * tpw : ks, t_step = run_tiles ? 1 : ksplit;
    const int t_end = run_tiles ? (t_first + tpw < ntiles ? t_first + tpw : ntiles) : ntiles;
    if (t_first < t_end) { if constexpr (VEC) issue_v(t_first); else issue(t_first); }
    int parity = 0;
    for (int tile = t_first; tile < t_end; tile += t_step) {
        g_t = g_t2 + parity * WG_BYTES;                                 // this tile's buffers (uniform)
        i_t = i_t2 + parity * WI_BYTES;
        const unsigned char* ap = g_t + ap_lane;
        const unsigned char* bp = i_t + bp_lane;
        parity ^= 1;
        if constexpr (VEC) commit_v(tile); else commit(tile);
        __syncthreads();
        if (tile + t_step < t_end) { if constexpr (VEC) issue_v(tile + t_step); else issue(tile + t_step); }   // in flight during this tile's MFMAs
        bf16x8 a[2][2];                                                 // [output row][co half of 16]
#pragma unroll
        for (int orow = 0; orow < 2; ++orow)
#pragma unroll
            for (int u = 0; u < 2; ++u) a[orow][u] = *reinterpret_cast<const bf16x8*>(ap + (orow * 64 + u * 16) * WG_P);
#pragma unroll
        for (int ro = 0; ro < 4; ++ro) {
            const unsigned char* p = bp + ro * 64 * WI_P;
            const u32x4 cur = *reinterpret_cast<const u32x4*>(p);
            const uint32_t prevd = *reinterpret_cast<const uint32_t*>(p - 4);
            const uint32_t nextd = *reinterpret_cast<const uint32_t*>(p + 16);
            u32x4 f0, f2;
            f0[0] = __builtin_amdgcn_alignbit(cur[0], prevd, 16);
            f0[1] = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
            f0[2] = __builtin_amdgcn_alignbit(cur[2], cur[1], 16);
            f0[3] = __builtin_amdgcn_alignbit(cur[3], cur[2], 16);
            f2[0] = f0[1]; f2[1] = f0[2]; f2[2] = f0[3];
            f2[3] = __builtin_amdgcn_alignbit(nextd, cur[3], 16);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, f0), b1 = __builtin_bit_cast(bf16x8, cur), b2 = __builtin_bit_cast(bf16x8, f2);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int orow = ro - ky;
                if (orow >= 0 && orow < 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[u][ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[orow][u], b0, acc[u][ky * 3 + 0], 0, 0, 0);
                        acc[u][ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[orow][u], b1, acc[u][ky * 3 + 1], 0, 0, 0);
                        acc[u][ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[orow][u], b2, acc[u][ky * 3 + 2], 0, 0, 0);
                    }
                }
            }
        }
        // no barrier here: the next tile's stores go to the other buffer pair, whose last readers (tile t-1) are all behind the
        // barrier every wave has just passed
    }
    // ---- partial sums -> slab (wgrad_slab_index)   (D of 16x16x32: column = lane & 15, row = 4 * (lane >> 4) + register)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = cb * 64 + wi * 32 + u * 16 + q4 * 4 + e;
                const int ci = ib * 64 + wj * 16 + r;
                slab[wgrad_slab_index(ks, t, co, ci, CoutP, CinP)] = acc[u][t][e];
            }
    if (do_bias && VEC) {  // the 16 lanes that share a channel (tid & 15 = row, group) added in a fixed butterfly order
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = bsum2[k];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if ((tid & 15) == 0) bias_slab[(int64_t)ks * CoutP + cb * 64 + ((tid + 512 * k) >> 4)] = v;
        }
    }
    if (do_bias && !VEC) { // one row of partial sums per K slice: lanes of a wave added in a fixed butterfly order
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            float v = bsum[k];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) bias_slab[(int64_t)ks * CoutP + cb * 64 + wave + 8 * k] = v;
        }
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

hipError_t launch_pack_weights_3x3_bf16_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s)
{
    const int CO_f = conv3x3_bf16_co_block(Cout), CO_t = conv3x3_bf16_co_block(Cin);
    const int nchunks_f = (Cin + BKC - 1) / BKC, nchunks_t = (Cout + BKC - 1) / BKC;
    const int64_t n_f = wp_f ? packed_bf16_elems(Cin, Cout) : 0, n_t = wp_t ? packed_bf16_elems(Cout, Cin) : 0;
    hipLaunchKernelGGL(pack_weights_3x3_bf16_both, dim3(grid_1d_bf(n_f + n_t, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wp_f),
                       reinterpret_cast<__bf16*>(wp_t), Cin, Cout, CO_f, nchunks_f, n_f, CO_t, nchunks_t, n_t);
    return hipGetLastError();
}

int64_t pack_group_entry_bf16(int Cin, int Cout, int64_t* out)
{
    const int CO_f = conv3x3_bf16_co_block(Cout), CO_t = conv3x3_bf16_co_block(Cin);
    const int nchunks_f = (Cin + BKC - 1) / BKC, nchunks_t = (Cout + BKC - 1) / BKC;
    out[3] = Cin; out[4] = Cout;
    out[5] = CO_f; out[6] = nchunks_f; out[7] = (Cout + CO_f - 1) / CO_f; out[8] = packed_bf16_elems(Cin, Cout);
    out[9] = CO_t; out[10] = nchunks_t; out[11] = (Cin + CO_t - 1) / CO_t; out[12] = packed_bf16_elems(Cout, Cin);
    return (out[8] + out[12] + 255) / 256;
}

hipError_t launch_pack_weights_3x3_bf16_group(const int64_t* table, int n_entries, int64_t total_blocks, hipStream_t s)
{
    if (n_entries <= 0 || total_blocks <= 0) return hipSuccess;
    if (total_blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_weights_3x3_bf16_group, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_entries);
    return hipGetLastError();
}

bool conv3x3_bf16_supported(int N, int Cin, int H, int W, int Cout)
{
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO;
    // 32-bit byte offsets: inside one channel plane for the 16-byte staging path (W % 4 == 0; the channel is part of the uniform
    // 64-bit base), inside one whole image for the dword path (buffer addressing, padding lanes at 2^31)
    const int64_t plane_bytes = (int64_t)H * W * 4;
    const bool fits = (W % 4 == 0) ? plane_bytes < (int64_t)OOB : (int64_t)Cin * plane_bytes < (int64_t)OOB;
    return Cin > 0 && fits && (int64_t)N * ncb * 8 <= 65535;
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

// in_bf16 / out_bf16: the activation tensors are bf16 NCHW (inference inside a block; 16-byte staging path, no split-K with a bf16
// output -- the callers check with conv3x3_bf16_io_supported)
bool conv3x3_bf16_io_supported(int N, int Cin, int H, int W, int Cout, int out_bf16)
{
    if (!conv3x3_bf16_supported(N, Cin, H, W, Cout) || W % 4 != 0) return false;
    return !(out_bf16 && conv3x3_bf16_ksplit(N, Cin, H, W, Cout) > 1);
}

hipError_t launch_conv3x3_bf16_mfma_io(const void* in, int in_bf16, const float* w, const float* bias, const float* scale,
                                       const float* shift, void* out, int out_bf16, float* workspace, int64_t workspace_floats,
                                       int N, int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                                       hipStream_t s, const uint8_t* in_mask, uint8_t* out_mask)
{
    if (!conv3x3_bf16_supported(N, Cin, H, W, Cout)) return hipErrorInvalidValue;
    const int CO = conv3x3_bf16_co_block(Cout);
    const int ncb = (Cout + CO - 1) / CO, nchunks = (Cin + BKC - 1) / BKC;
    const int64_t welems = packed_bf16_elems(Cin, Cout);
    __bf16* wp = reinterpret_cast<__bf16*>(workspace);
    const bool prepacked = (w_transposed_flipped & 2) != 0;      // SSTEM_CONV_WEIGHT_PREPACKED, as in launch_conv3x3_mfma
    w_transposed_flipped &= 1;
    hipError_t e = hipSuccess;
    if (!prepacked) {
        hipLaunchKernelGGL(pack_weights_3x3_bf16, dim3(grid_1d_bf(welems, 256)), dim3(256), 0, s, w, wp, Cin, Cout, CO, nchunks,
                           ncb, w_transposed_flipped);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    int ksplit = conv3x3_bf16_ksplit(N, Cin, H, W, Cout);
    const int64_t out_elems = (int64_t)N * Cout * H * W;
    if (ksplit > 1 && workspace_floats < welems / 2 + (int64_t)ksplit * out_elems) ksplit = 1;
    if (out_bf16 && ksplit > 1) return hipErrorInvalidValue;
    float* slab = workspace + welems / 2;
    const dim3 grid((W + BTW - 1) / BTW, (H + BTH - 1) / BTH, (unsigned)(N * ncb * ksplit));
    static const bool novec = [] { const char* e = getenv("SSTEM_BF16_NOVEC"); return e && atoi(e) != 0; }();     // developer knob (A/B runs)
    const bool vec = (!novec || in_bf16) && W % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    if (in_bf16 && !vec) return hipErrorInvalidValue;
    if (!vec && (int64_t)Cin * H * W * 4 >= (int64_t)OOB) return hipErrorInvalidValue;      // dword path: whole image below 2 GiB
    static const int remap_knob = [] { const char* e = getenv("SSTEM_XCD_REMAP"); return e ? atoi(e) : 1; }();    // developer knob (A/B runs)
    const int remap = (remap_knob && (int64_t)grid.x * grid.y * grid.z < ((int64_t)1 << 31)) ? 1 : 0;   // 32-bit linear tile ids in the kernel
    const bool masked = in_mask != nullptr || out_mask != nullptr;
    if (masked && !vec) return hipErrorInvalidValue;                       // masks: 16-byte staging
    if (in_mask && in_bf16) return hipErrorInvalidValue;                   // ... and in_mask an fp32 input
    uint8_t* kernel_out_mask = ksplit > 1 ? nullptr : out_mask;             // a launch split over K leaves the mask to its slice-sum launch
#define SSTEM_BF16_FWD(A, B, V, IB, OB)                                                                                          \
    hipLaunchKernelGGL((conv3x3_bf16_mfma<A, B, 2, V, IB, OB>), grid, dim3(256), 0, s, in, wp, bias, scale, shift, out, N, Cin, H, W, \
                       Cout, nchunks, ncb, act, slope, ksplit, slab, remap)
#define SSTEM_BF16_FWD_M(A, B, IB, OB)                                                                                           \
    hipLaunchKernelGGL((conv3x3_bf16_mfma<A, B, 2, true, IB, OB, true>), grid, dim3(256), 0, s, in, wp, bias, scale, shift, out, N,  \
                       Cin, H, W, Cout, nchunks, ncb, act, slope, ksplit, slab, remap, in_mask, kernel_out_mask)
#define SSTEM_BF16_SHAPE(A, B)                                                                       \
    do {                                                                                             \
        if (masked && in_bf16) { if (out_bf16) SSTEM_BF16_FWD_M(A, B, true, true); else SSTEM_BF16_FWD_M(A, B, true, false); } \
        else if (masked) { if (out_bf16) SSTEM_BF16_FWD_M(A, B, false, true); else SSTEM_BF16_FWD_M(A, B, false, false); } \
        else                                                                                         \
        if (!vec) { if (out_bf16) SSTEM_BF16_FWD(A, B, false, false, true); else SSTEM_BF16_FWD(A, B, false, false, false); } \
        else if (in_bf16 && out_bf16) SSTEM_BF16_FWD(A, B, true, true, true);                        \
        else if (in_bf16) SSTEM_BF16_FWD(A, B, true, true, false);                                   \
        else if (out_bf16) SSTEM_BF16_FWD(A, B, true, false, true);                                  \
        else SSTEM_BF16_FWD(A, B, true, false, false);                                               \
    } while (0)
    if (CO == 64) SSTEM_BF16_SHAPE(2, 2); else SSTEM_BF16_SHAPE(1, 4);
#undef SSTEM_BF16_SHAPE
#undef SSTEM_BF16_FWD_M
#undef SSTEM_BF16_FWD
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    hipLaunchKernelGGL(conv3x3_bf16_splitk_epilogue, dim3(grid_1d_bf(out_elems, 256)), dim3(256), 0, s, slab, bias, scale,
                       shift, static_cast<float*>(out), out_elems, (int64_t)H * W, Cout, ksplit, act, slope, out_mask);
    return hipGetLastError();
}

hipError_t launch_conv3x3_bf16_mfma(const float* in, const float* w, const float* bias, const float* scale,
                                    const float* shift, float* out, float* workspace, int64_t workspace_floats, int N,
                                    int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                                    hipStream_t s)
{
    return launch_conv3x3_bf16_mfma_io(in, 0, w, bias, scale, shift, out, 0, workspace, workspace_floats, N, Cin, H, W, Cout, act, slope,
                                       w_transposed_flipped, s, nullptr, nullptr);
}

// pixel-tile split of the weight gradient: the plan of conv3x3_wgrad_mfma with fewer, longer workgroups (a tile's MFMA phase is
// a third as long here, so the partial-slab traffic weighs more): one 8-wave workgroup per CU (all its registers allow), at least
// 8 tiles each (16 left half the CUs idle on the 128x128 and smaller layers of an 8 x 256x256 step: 0.046 -> 0.035 ms).
// Tried and dropped: TWO tiles of loads in flight (register sets used alternately).  Written in plain C++ the compiler's
// loop-carried vmcnt bookkeeping waits for the set issued last (one tile in flight again); with the loads and the wait as opaque
// asm statements the timing did not move (16x64->64 at 256^2: 0.195 -> 0.187 ms) and larger shapes read registers before their
// loads had landed (NaN in the gradient) -- the kernel is not bound by the depth of its prefetch.  Measured: 256 against 512 workgroups 0.217 -> 0.195 ms on 16x64->64 at 256^2, 0.233 -> 0.174 on 16x128->128 at 128^2
struct WgradBf16Plan { int CinP, CoutP, ksplit, tx, ty; };
static WgradBf16Plan wgrad_bf16_plan(int N, int Cin, int H, int W, int Cout)
{
    static const int target = [] { const char* e = getenv("SSTEM_WGRAD_BF16_TARGET"); return e ? atoi(e) : 256; }();
    static const int min_tiles = [] { const char* e = getenv("SSTEM_WGRAD_BF16_MIN_TILES"); return e ? atoi(e) : 8; }();
    WgradBf16Plan p;
    p.CinP = (Cin + 63) / 64 * 64;
    p.CoutP = (Cout + 63) / 64 * 64;
    p.tx = (W + BTW - 1) / BTW;
    p.ty = (H + 1) / 2;
    const int64_t ntiles = (int64_t)N * p.tx * p.ty;
    const int blocks = (p.CinP / 64) * (p.CoutP / 64);
    int64_t k = (target + blocks - 1) / blocks;
    if (k > ntiles / min_tiles) k = ntiles / min_tiles;
    if (k < 1) k = 1;
    p.ksplit = (int)k;
    return p;
}

int64_t conv3x3_wgrad_bf16_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const WgradBf16Plan p = wgrad_bf16_plan(N, Cin, H, W, Cout);
    return (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) + (int64_t)p.ksplit * p.CoutP;
}

hipError_t launch_conv3x3_wgrad_bf16_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                          int H, int W, int Cout, hipStream_t s, int accumulate)
{
    return launch_conv3x3_wgrad_bf16_mfma_in(in, 0, g, gw, gb, workspace, N, Cin, H, W, Cout, s, accumulate, nullptr);
}

hipError_t launch_conv3x3_wgrad_bf16_mfma_in(const void* in, int in_bf16, const float* g, float* gw, float* gb, float* workspace, int N,
                                             int Cin, int H, int W, int Cout, hipStream_t s, int accumulate, const uint8_t* g_mask)
{
    if ((int64_t)H * W * 4 * 64 >= ((int64_t)1 << 32)) return hipErrorInvalidValue;      // 32-bit offsets over the 64 channels of a block
    const WgradBf16Plan p = wgrad_bf16_plan(N, Cin, H, W, Cout);
    float* bias_slab = gb ? workspace + (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) : nullptr;
    const int blocks = (p.CinP / 64) * (p.CoutP / 64);
    static const bool novec = [] { const char* e = getenv("SSTEM_BF16_NOVEC"); return e && atoi(e) != 0; }();     // developer knob (A/B runs)
    static const int runs = [] { const char* e = getenv("SSTEM_WGRAD_RUNS"); return e ? atoi(e) : 2; }();          // developer knob: 0 strided, 1 runs along x, 2 runs down a column strip
    const bool vec = (!novec || in_bf16) && W % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (in_bf16 && !vec) return hipErrorInvalidValue;
    if (g_mask && !vec) return hipErrorInvalidValue;
    if (g_mask && in_bf16)
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_mfma<true, true, true>), dim3((unsigned)(blocks * p.ksplit)), dim3(512), 0, s, in, g, workspace, N,
                           Cin, H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs, g_mask);
    else if (g_mask)
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_mfma<true, false, true>), dim3((unsigned)(blocks * p.ksplit)), dim3(512), 0, s, in, g, workspace, N,
                           Cin, H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs, g_mask);
    else if (vec && in_bf16)
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_mfma<true, true>), dim3((unsigned)(blocks * p.ksplit)), dim3(512), 0, s, in, g, workspace, N, Cin,
                           H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs);
    else if (vec)
        hipLaunchKernelGGL(conv3x3_wgrad_bf16_mfma<true>, dim3((unsigned)(blocks * p.ksplit)), dim3(512), 0, s, in, g, workspace, N, Cin,
                           H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_bf16_mfma<false>, dim3((unsigned)(blocks * p.ksplit)), dim3(512), 0, s, in, g, workspace, N, Cin,
                           H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_conv3x3_wgrad_reduce(workspace, gw, Cin, Cout, p.CinP, p.CoutP, p.ksplit, bias_slab, gb, p.ksplit, s, accumulate);
}

}  // namespace sstem
