// 3x3 weight gradient with split operands (bf16 pieces: SSTEM_CONV_MFMA_BF16X6 / _BF16X3; two fp16 pieces under amax scales: the recorded
// launches of SSTEM_CONV_MFMA_F16X3) -- a translation unit of its own since round 5 (conv_split_kernels.hip holds the forward / data
// gradient instances and takes minutes to compile; the arithmetic and the piece formats are described at its top).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_kernels.h"
#include "conv_split_common.h"

#ifndef SSTEM_SPLIT_DEV
#define SSTEM_SPLIT_DEV 0
#endif

// developer builds (tools/build_wgrad_dev.sh): -DSSTEM_WGRAD_STAMPS=1 adds shader-clock stamps around the phases of a tile (wave 0 of
// every workgroup adds its phase times to g_wgrad_stamps; sstem_debug_wgrad_stamps reads and clears them)
#ifndef SSTEM_WGRAD_STAMPS
#define SSTEM_WGRAD_STAMPS 0
#endif

namespace sstem {
#if SSTEM_WGRAD_STAMPS
__device__ unsigned long long g_wgrad_stamps[8];
#define STAMP(k) do { if (stamp_on) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_t; st_t = now_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
namespace {

// ---- 3x3 weight gradient with split operands ------------------------------------------------------------------------------
// gW[co][ci][tap] = sum over pixels of g[co][p] * in[ci][p + tap]: M = co, N = ci, K = pixels -- the geometry, LDS layouts, staging,
// slabs and the fixed-order reduce of conv3x3_wgrad_bf16_mfma (conv_bf16_kernels.hip), with both tiles held as P bf16 piece images
// and the products g_pa x in_pb, pa + pb < P, summed into the same fp32 accumulators (v_mfma_f32_16x16x32_bf16, K = 32 = one row
// of the 2-row x 32-column pixel tile).  One buffer set (P x 38 KB), two barriers per pixel tile: a tile's MFMA phase is
// P (P + 1) / 2 times as long as the bf16 kernel's.  The bias gradient is summed from the fp32 values.
// LDS row pitches (32 pixels + pad / 7 + 34 + pad elements).  96 B = 6 slots of 16 B: a ds_read_b128's 16-lane groups are not
// contiguous ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), and of the pitches that hold a row only 6, 10, 14 ... slots put the
// 16 fragment reads of every such group on 16 different slots (80 / 112 B, rounds 2-4: one extra LDS cycle per group)
constexpr int SWG_P = 96, SWI_P = 96;
// rows of a piece image start SWG_RP / SWI_RP bytes further on than their pitch says: the 16 lanes a ds_write_b64 group holds store
// 64 B of two consecutive rows each -- 64 x pitch is a multiple of 128 B, so without the offsets the two rows sat on the same banks (round 5, SQ_LDS_BANK_CONFLICT 60 % of the LDS cycles of the fp16 form: profiles/r05/j_*)
constexpr int SWG_RP = 64, SWI_RP = 64;
constexpr int SWG_BYTES = 2 * (64 * SWG_P + SWG_RP), SWI_BYTES = 4 * (64 * SWI_P + SWI_RP);
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));

// MASKED: g_mask (nullable, [N,Cout,H,W] bytes): g counts as 0 where the byte is 0 (the ReLU of the layer's output, see conv3x3_split_mfma)
// F16: the P = 2 pieces are fp16 under the tensors' power-of-two scales (in_amax / g_amax: their amax words; split_pieces_f16); the
// accumulators then hold the sums times both scales, taken out by one ldexp per stored value.  The bias gradient stays an fp32 sum.
template <int P, bool VEC, bool MASKED = false, bool F16 = false, bool PINGPONG = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_split_mfma(
    const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ slab,
    int N, int Cin, int H, int W, int Cout, int CinP, int CoutP, int ksplit, int tiles_x, int tiles_y,
    float* __restrict__ bias_slab, int run_tiles, const uint8_t* __restrict__ g_mask = nullptr,
    const float* __restrict__ in_amax = nullptr, const float* __restrict__ g_amax = nullptr)
{
    static_assert(!F16 || P == 2, "the fp16 pieces come in twos");
    int e_in = 141, e_g = 141;
    if constexpr (F16) { e_in = amax_exponent(amax_word_max(in_amax)); e_g = amax_exponent(amax_word_max(g_amax)); }
    const float s_in = scale_of_exponent(e_in), s_g = scale_of_exponent(e_g);
    auto pieces_of = [&](float v, float sc, __bf16 (&pc)[P]) __attribute__((always_inline)) {
        if constexpr (F16) split_pieces_f16(v * sc, pc); else split_pieces<P>(v, pc);
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
    // Two pieces (round 5): TWO buffer sets (2 x 74.5 KB), ONE barrier per tile, and the two waves of a SIMD half a tile apart: a
    // workgroup's waves w and w + 4 share a SIMD, its matrix pipe and its VALU issue (MI355X_MICROARCH.md, "Two waves per SIMD");
    // waves 0-3 multiply tile t and THEN split and store tile t + 1 into the other set, waves 4-7 store first and multiply after -- one
    // wave's MFMAs run beside its partner's staging instead of both phases taking turns (the stamps of tools/wgrad_stamps.py before:
    // MFMA phase 32 % of a workgroup's time, everything else one after the other).  Two sets alone, same order in every wave, changed
    // nothing (0.27 ms on 16 x 64 -> 64 at 256^2 either way).  Three pieces: one set (112 KB), two barriers, as before.
    constexpr bool DB = PINGPONG;
    static_assert(!PINGPONG || P == 2, "two buffer sets fit for two pieces only");
    constexpr int SET_BYTES = P * (SWG_BYTES + SWI_BYTES);
    unsigned char* const g_t0 = wlds;                              // [P][2 rows][64 co]
    unsigned char* const i_t0 = wlds + P * SWG_BYTES;              // [P][4 rows][64 ci]
    int wr_set = 0;                                                // the set the next commit stores into
    unsigned char* g_t = g_t0;
    unsigned char* i_t = i_t0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r = lane & 15;
    const int wi = wave >> 2, wj = wave & 3;
    const int nib = CinP / 64;
    uint32_t wgid = blockIdx.x;
    if (run_tiles) {                                               // XCD k owns neighbouring runs of tiles (see conv3x3_wgrad_bf16_mfma)
        const uint32_t total = gridDim.x, k8 = wgid & 7u, q8 = total >> 3, r8 = total & 7u;
        wgid = k8 * q8 + (k8 < r8 ? k8 : r8) + (wgid >> 3);
    }
    const int blk = (int)(wgid / (uint32_t)ksplit), ks = (int)(wgid % (uint32_t)ksplit);
    const int cb = blk / nib, ib = blk % nib;
    const int64_t plane = (int64_t)H * W;
    const int ntiles = N * tiles_y * tiles_x;
    const bool do_bias = (bias_slab != nullptr) && (ib == 0);

    f32x4s acc[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][t][e] = 0.f;

    auto geometry = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        if (run_tiles == 2) {
            const int ty = tile % tiles_y;
            const int r0 = tile / tiles_y;
            n = r0 / tiles_x; X0 = (r0 % tiles_x) * STW; Y0 = ty * 2;
        } else {
            const int tx = tile % tiles_x;
            const int r0 = tile / tiles_x;
            n = r0 / tiles_y; X0 = tx * STW; Y0 = (r0 % tiles_y) * 2;
        }
    };

    // ---- dword staging (any W): e = j*64 + lane of a channel's flat 4 x 34 input tile; 8 channels of each tile per wave
    constexpr int I_E = 4 * SIN_PW, I_J = 3, CH_W = 8;
    int er[I_J], ec[I_J];
#pragma unroll
    for (int j = 0; j < I_J; ++j) { const int e = j * 64 + lane; er[j] = e / SIN_PW; ec[j] = e - er[j] * SIN_PW; }
    float gv[VEC ? 1 : CH_W], ivp[VEC ? 1 : CH_W * I_J], bsum[CH_W];
#pragma unroll
    for (int k = 0; k < CH_W; ++k) bsum[k] = 0.f;
    auto lane_offsets = [&](int X0, int Y0, uint32_t (&off)[I_J], bool (&ok)[I_J]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < I_J; ++j) {
            const int yi = Y0 - 1 + er[j], xi = X0 - 1 + ec[j];
            ok[j] = (j * 64 + lane < I_E) && yi >= 0 && yi < H && xi >= 0 && xi < W;
            off[j] = ok[j] ? (uint32_t)(yi * W + xi) * 4u : 0u;
        }
    };
    auto issue = [&](int tile) __attribute__((always_inline)) {
        if constexpr (!VEC) {
            int n, X0, Y0;
            geometry(tile, n, X0, Y0);
            const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
            const uint32_t poff = (yy < H && xx < W) ? (uint32_t)(yy * W + xx) * 4u : 0u;
#pragma unroll
            for (int k = 0; k < CH_W; ++k) {
                const int co = cb * 64 + wave + 8 * k;
                const float* base = g + ((int64_t)n * Cout + (co < Cout ? co : 0)) * plane;
                float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + poff);
                if constexpr (MASKED) { if (g_mask && g_mask[((int64_t)n * Cout + (co < Cout ? co : 0)) * plane + (poff >> 2)] == 0) v = 0.f; }
                gv[k] = v;
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
        }
    };
    auto commit = [&](int tile) __attribute__((always_inline)) {
        if constexpr (!VEC) {
            int n, X0, Y0;
            geometry(tile, n, X0, Y0);
            const int yy = Y0 + (lane >> 5), xx = X0 + (lane & 31);
            const bool pix_ok = yy < H && xx < W;
#pragma unroll
            for (int k = 0; k < CH_W; ++k) {
                const int c = wave + 8 * k;
                const float v = (pix_ok && cb * 64 + c < Cout) ? gv[k] : 0.f;
                bsum[k] += v;
                __bf16 pc[P];
                pieces_of(v, s_g, pc);
#pragma unroll
                for (int p = 0; p < P; ++p)
                    *reinterpret_cast<__bf16*>(g_t + p * SWG_BYTES + ((lane >> 5) * 64 + c) * SWG_P + (lane >> 5) * SWG_RP + (lane & 31) * 2) = pc[p];
            }
            uint32_t off[I_J]; bool ok[I_J];
            lane_offsets(X0, Y0, off, ok);
#pragma unroll
            for (int k = 0; k < CH_W; ++k) {
                const int c = wave + 8 * k;
                const bool ch_ok = ib * 64 + c < Cin;
#pragma unroll
                for (int j = 0; j < I_J; ++j)
                    if (j * 64 + lane < I_E) {
                        __bf16 pc[P];
                        pieces_of((ch_ok && ok[j]) ? ivp[k * I_J + j] : 0.f, s_in, pc);
#pragma unroll
                        for (int p = 0; p < P; ++p)
                            *reinterpret_cast<__bf16*>(i_t + p * SWI_BYTES + (er[j] * 64 + c) * SWI_P + er[j] * SWI_RP + (7 + ec[j]) * 2) = pc[p];
                    }
            }
        }
    };

    // ---- 16-byte staging (W % 4 == 0): items of 4 pixels; channel, row and group of an item are fixed per thread
    const uint32_t plane4 = (uint32_t)plane * 4u;
    uint32_t vg_off[2], vi_off[4], vh_off;
    int vg_lds[2], vi_lds[4], vh_lds;
    bool vg_ch[2], vi_ch[4], vh_ch;
    f32x4s gq[2], iq[4];
    uint32_t mq[MASKED ? 2 : 1];
    float hq = 0.f, bsum2[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = tid + 512 * k, ch = item >> 4, row = (item >> 3) & 1, grp = item & 7;
        vg_ch[k] = cb * 64 + ch < Cout;
        vg_off[k] = (uint32_t)ch * plane4 + (uint32_t)(row * W + 4 * grp) * 4u;
        vg_lds[k] = (row * 64 + ch) * SWG_P + row * SWG_RP + grp * 8;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int item = tid + 512 * k, ch = item >> 5, row = (item >> 3) & 3, grp = item & 7;
        vi_ch[k] = ib * 64 + ch < Cin;
        vi_off[k] = ((uint32_t)ch * (uint32_t)plane + (uint32_t)(row * W + 4 * grp)) * 4u;
        vi_lds[k] = (row * 64 + ch) * SWI_P + row * SWI_RP + 16 + grp * 8;
    }
    {
        const int ch = tid >> 3, row = (tid >> 1) & 3, side = tid & 1;
        vh_ch = ib * 64 + ch < Cin;
        vh_off = ((uint32_t)ch * (uint32_t)plane + (uint32_t)(row * W)) * 4u;
        vh_lds = (row * 64 + ch) * SWI_P + row * SWI_RP + (side ? 40 : 7) * 2;
    }
    // (bitwise, not &&: the short-circuit forms compiled to a chain of exec-mask branches, seven per tile)
    auto vec_ok = [&](int X0, int Y0, bool (&gk)[2], bool (&ik)[4], bool& hk, int& hx) __attribute__((always_inline)) {
        const bool xin = X0 + 4 * (tid & 7) < W;                        // the group of four columns is the same for all of a thread's items
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 512 * k, row = (item >> 3) & 1;
            gk[k] = vg_ch[k] & (Y0 + row < H) & xin;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = tid + 512 * k, row = (item >> 3) & 3;
            const uint32_t y = (uint32_t)(Y0 - 1 + row);
            ik[k] = vi_ch[k] & (y < (uint32_t)H) & xin;
        }
        const uint32_t y = (uint32_t)(Y0 - 1 + ((tid >> 1) & 3));
        hx = (tid & 1) ? X0 + STW : X0 - 1;
        hk = vh_ch & (y < (uint32_t)H) & ((uint32_t)hx < (uint32_t)W);
    };
    // the staged tiles follow one another (run_tiles: consecutive tile numbers): their coordinates by increments, not by two divisions
    int nx_n = 0, nx_tx = 0, nx_ty = 0;
    auto geometry_seq = [&](int tile, int& n, int& X0, int& Y0) __attribute__((always_inline)) {
        if (!run_tiles) { geometry(tile, n, X0, Y0); return; }
        n = nx_n; X0 = nx_tx * STW; Y0 = nx_ty * 2;
        if (run_tiles == 2) {
            if (++nx_ty == tiles_y) { nx_ty = 0; if (++nx_tx == tiles_x) { nx_tx = 0; ++nx_n; } }
        } else {
            if (++nx_tx == tiles_x) { nx_tx = 0; if (++nx_ty == tiles_y) { nx_ty = 0; ++nx_n; } }
        }
    };
    bool gk[2], ik[4], hk = false;          // the staged tile's validity flags: set when it is issued, read again when it is committed
    auto issue_v = [&](int tile) __attribute__((always_inline)) {
        int n, X0, Y0;
        geometry_seq(tile, n, X0, Y0);
        int hx;
        vec_ok(X0, Y0, gk, ik, hk, hx);
        const char* gbase = reinterpret_cast<const char*>(g + ((int64_t)n * Cout + cb * 64) * plane);     // uniform
        const char* ibase = reinterpret_cast<const char*>(in + ((int64_t)n * Cin + ib * 64) * plane);
        const uint32_t tg = (uint32_t)(Y0 * W + X0) * 4u, ti = (uint32_t)((Y0 - 1) * W + X0) * 4u;         // ti may wrap: rows >= 1 undo it
#pragma unroll
        for (int k = 0; k < 2; ++k) gq[k] = *reinterpret_cast<const f32x4s*>(gbase + (gk[k] ? vg_off[k] + tg : 0u));
        if constexpr (MASKED) {
            if (g_mask) {
                const uint8_t* mbase = g_mask + ((int64_t)n * Cout + cb * 64) * plane;                      // uniform
#pragma unroll
                for (int k = 0; k < 2; ++k) mq[k] = *reinterpret_cast<const uint32_t*>(mbase + (gk[k] ? (vg_off[k] + tg) >> 2 : 0u));
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) iq[k] = *reinterpret_cast<const f32x4s*>(ibase + (ik[k] ? vi_off[k] + ti : 0u));
        hq = *reinterpret_cast<const float*>(ibase + (hk ? vh_off + (uint32_t)((Y0 - 1) * W + hx) * 4u : 0u));
    };
    typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
    // fp16 pieces of a pair: head = fp16(v s), tail = fp16(fma(v, s, -head)), written into the halves of two registers by four
    // mixed-precision fmas (as conv3x3_split_mfma's staging; the same values as split_pieces_f16)
    auto pair_f16 = [&](float v0, float v1, float sc, uint32_t& hd, uint32_t& tl) __attribute__((always_inline)) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hd) : "v"(v0), "v"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hd) : "v"(v1), "v"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(tl) : "v"(v0), "v"(sc), "v"(hd));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(tl) : "v"(v1), "v"(sc), "v"(hd));
    };
    auto commit_v = [&](int) __attribute__((always_inline)) {
        if constexpr (F16) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = gk[k] ? gq[k][e] : 0.f;
                    if constexpr (MASKED) { if (g_mask && ((mq[k] >> (8 * e)) & 0xffu) == 0u) v[e] = 0.f; }
                }
                bsum2[k] += (v[0] + v[1]) + (v[2] + v[3]);
                uint32_t h0, h1, t0, t1;
                pair_f16(v[0], v[1], s_g, h0, t0);
                pair_f16(v[2], v[3], s_g, h1, t1);
                *reinterpret_cast<u32x2s*>(g_t + vg_lds[k]) = u32x2s{h0, h1};
                *reinterpret_cast<u32x2s*>(g_t + SWG_BYTES + vg_lds[k]) = u32x2s{t0, t1};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t h0, h1, t0, t1;
                pair_f16(ik[k] ? iq[k][0] : 0.f, ik[k] ? iq[k][1] : 0.f, s_in, h0, t0);
                pair_f16(ik[k] ? iq[k][2] : 0.f, ik[k] ? iq[k][3] : 0.f, s_in, h1, t1);
                *reinterpret_cast<u32x2s*>(i_t + vi_lds[k]) = u32x2s{h0, h1};
                *reinterpret_cast<u32x2s*>(i_t + SWI_BYTES + vi_lds[k]) = u32x2s{t0, t1};
            }
            __bf16 pc[P];
            pieces_of(hk ? hq : 0.f, s_in, pc);
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<__bf16*>(i_t + p * SWI_BYTES + vh_lds) = pc[p];
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x4s pk[P];
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = gk[k] ? gq[k][e] : 0.f;
                if constexpr (MASKED) { if (g_mask && ((mq[k] >> (8 * e)) & 0xffu) == 0u) v = 0.f; }
                sum += v;
                __bf16 pc[P];
                pieces_of(v, s_g, pc);
#pragma unroll
                for (int p = 0; p < P; ++p) pk[p][e] = pc[p];
            }
            bsum2[k] += sum;
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<bf16x4s*>(g_t + p * SWG_BYTES + vg_lds[k]) = pk[p];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x4s pk[P];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 pc[P];
                pieces_of(ik[k] ? iq[k][e] : 0.f, s_in, pc);
#pragma unroll
                for (int p = 0; p < P; ++p) pk[p][e] = pc[p];
            }
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<bf16x4s*>(i_t + p * SWI_BYTES + vi_lds[k]) = pk[p];
        }
        {
            __bf16 pc[P];
            pieces_of(hk ? hq : 0.f, s_in, pc);
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<__bf16*>(i_t + p * SWI_BYTES + vh_lds) = pc[p];
        }
    };

    const unsigned char* ap0 = g_t0 + (wi * 32 + r) * SWG_P + q4 * 16;
    const unsigned char* bp0 = i_t0 + (wj * 16 + r) * SWI_P + 16 + q4 * 16;
    const int tpw = (ntiles + ksplit - 1) / ksplit;
    const int t_first = run_tiles ? ks * tpw : ks, t_step = run_tiles ? 1 : ksplit;
    const int t_end = run_tiles ? (t_first + tpw < ntiles ? t_first + tpw : ntiles) : ntiles;
#if SSTEM_WGRAD_STAMPS
    const bool stamp_on = tid == 0;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif
    auto stage = [&](int tile) __attribute__((always_inline)) { if constexpr (VEC) issue_v(tile); else issue(tile); };
    auto store = [&](int tile) __attribute__((always_inline)) {
        g_t = g_t0 + wr_set * SET_BYTES; i_t = i_t0 + wr_set * SET_BYTES;
        if constexpr (VEC) commit_v(tile); else commit(tile);
    };
    auto mfma_phase = [&](const unsigned char* ap, const unsigned char* bp) __attribute__((always_inline)) {
        bf16x8 a[P][2][2];                                              // [piece][output row][co half of 16]
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int orow = 0; orow < 2; ++orow)
#pragma unroll
                for (int u = 0; u < 2; ++u) a[p][orow][u] = *reinterpret_cast<const bf16x8*>(ap + p * SWG_BYTES + (orow * 64 + u * 16) * SWG_P + orow * SWG_RP);
        // the B operand of step (input row ro, piece pb) is read one step ahead and pinned there (left alone the scheduler sinks the reads to
        // their use: an LDS round trip in front of every step's MFMAs)
        typedef uint32_t u32x2r __attribute__((ext_vector_type(2)));
        u32x4s curv[2];
        uint32_t pvv[2], nxv[2];
        auto read_b = [&](int step, int slot) __attribute__((always_inline)) {
            const unsigned char* p = bp + (step % P) * SWI_BYTES + (step / P) * (64 * SWI_P + SWI_RP);
            curv[slot] = *reinterpret_cast<const u32x4s*>(p);
            // the neighbours' edge elements as 8-byte reads (a dword read's lanes all sit on 8 of the 32 banks: 4-way; these: 2-way)
            pvv[slot] = (*reinterpret_cast<const u32x2r*>(p - 8))[1];
            nxv[slot] = (*reinterpret_cast<const u32x2r*>(p + 16))[0];
        };
        read_b(0, 0);
#pragma unroll
        for (int step = 0; step < 4 * P; ++step) {
            const int ro = step / P, pb = step % P, slot = step & 1;
            if (step + 1 < 4 * P) read_b(step + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4s cur = curv[slot];
            const uint32_t prevd = pvv[slot], nextd = nxv[slot];
            u32x4s f0, f2;
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
                    for (int pa = 0; pa + pb < P; ++pa) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if constexpr (F16) {
                                const f16x8 ah = __builtin_bit_cast(f16x8, a[pa][orow][u]);
                                acc[u][ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, b0), acc[u][ky * 3 + 0], 0, 0, 0);
                                acc[u][ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, b1), acc[u][ky * 3 + 1], 0, 0, 0);
                                acc[u][ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, b2), acc[u][ky * 3 + 2], 0, 0, 0);
                            } else {
                                acc[u][ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][orow][u], b0, acc[u][ky * 3 + 0], 0, 0, 0);
                                acc[u][ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][orow][u], b1, acc[u][ky * 3 + 1], 0, 0, 0);
                                acc[u][ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][orow][u], b2, acc[u][ky * 3 + 2], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    };
    if (run_tiles && t_first < t_end) {
        int X0, Y0;
        geometry(t_first, nx_n, X0, Y0);
        nx_tx = X0 / STW; nx_ty = Y0 >> 1;
    }
    if (t_first < t_end) stage(t_first);
    STAMP(0);
    if constexpr (DB) {
        if (t_first < t_end) {
            store(t_first);
            if (t_first + t_step < t_end) stage(t_first + t_step);
        }
        __syncthreads();
        const bool multiply_first = wave < 4;                           // (uniform per wave)
        for (int tile = t_first; tile < t_end; tile += t_step) {
            const int rd = wr_set * SET_BYTES;                          // the set stored before the last barrier
            wr_set ^= 1;
            STAMP(2);
            if (multiply_first) mfma_phase(ap0 + rd, bp0 + rd);
            STAMP(4);
            if (tile + t_step < t_end) {
                store(tile + t_step);                                   // (its loads were issued a tile ago)
                STAMP(1);
                if (tile + 2 * t_step < t_end) stage(tile + 2 * t_step);
                STAMP(3);
            }
            if (!multiply_first) mfma_phase(ap0 + rd, bp0 + rd);
            STAMP(4);
            __syncthreads();                                            // tile + 1 is stored, and every wave is done reading tile - 1's set
            STAMP(5);
        }
    } else {
        for (int tile = t_first; tile < t_end; tile += t_step) {
            store(tile);
            STAMP(1);
            __syncthreads();
            STAMP(2);
            if (tile + t_step < t_end) stage(tile + t_step);            // in flight during this tile's MFMAs
            STAMP(3);
            mfma_phase(ap0, bp0);
            STAMP(4);
            __syncthreads();                                            // every wave has read this tile before the next one is stored
            STAMP(5);
        }
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
                slab[wgrad_slab_index(ks, t, co, ci, CoutP, CinP)] = F16 ? ldexpf(acc[u][t][e], e_in + e_g - 282) : acc[u][t][e];
            }
    if (do_bias && VEC) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = bsum2[k];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if ((tid & 15) == 0) bias_slab[(int64_t)ks * CoutP + cb * 64 + ((tid + 512 * k) >> 4)] = v;
        }
    }
    if (do_bias && !VEC) {
#pragma unroll
        for (int k = 0; k < CH_W; ++k) {
            float v = bsum[k];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) bias_slab[(int64_t)ks * CoutP + cb * 64 + wave + 8 * k] = v;
        }
    }
#if SSTEM_WGRAD_STAMPS
    STAMP(6);
    if (stamp_on) {
        for (int k = 0; k < 7; ++k) atomicAdd(&g_wgrad_stamps[k], st_acc[k]);
        atomicAdd(&g_wgrad_stamps[7], 1ull);
    }
#endif
}

}  // namespace

// pixel-tile split of the weight gradient: the plan of conv3x3_wgrad_bf16_mfma (one 8-wave workgroup per CU) with at least 2 tiles per
// workgroup instead of 8 -- a tile's MFMA phase is six times as long here (measured at 2 / 4 / 8: 2x64->64 at 128^2 0.033 / 0.037 / 0.056 ms,
// 8x64->128 at 64^2 0.044 / 0.044 / 0.059 ms)
struct WgradSplitPlan { int CinP, CoutP, ksplit, tx, ty; };
static WgradSplitPlan wgrad_split_plan(int N, int Cin, int H, int W, int Cout)
{
    static const int target = [] { const char* e = getenv("SSTEM_WGRAD_SPLIT_TARGET"); return e ? atoi(e) : 256; }();
    static const int min_tiles = [] { const char* e = getenv("SSTEM_WGRAD_SPLIT_MIN_TILES"); return e ? atoi(e) : 2; }();
    WgradSplitPlan p;
    p.CinP = (Cin + 63) / 64 * 64;
    p.CoutP = (Cout + 63) / 64 * 64;
    p.tx = (W + STW - 1) / STW;
    p.ty = (H + 1) / 2;
    const int64_t ntiles = (int64_t)N * p.tx * p.ty;
    const int blocks = (p.CinP / 64) * (p.CoutP / 64);
    int64_t k = (target + blocks - 1) / blocks;
    if (k > ntiles / min_tiles) k = ntiles / min_tiles;
    if (k < 1) k = 1;
    p.ksplit = (int)k;
    return p;
}

int64_t conv3x3_wgrad_split_workspace_floats(int N, int Cin, int H, int W, int Cout)
{
    const WgradSplitPlan p = wgrad_split_plan(N, Cin, H, W, Cout);
    return (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) + (int64_t)p.ksplit * p.CoutP;
}

bool conv3x3_wgrad_split_supported(int N, int Cin, int H, int W, int Cout) { return (int64_t)H * W * 4 * 64 < ((int64_t)1 << 32); }

hipError_t launch_conv3x3_wgrad_split_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                           int H, int W, int Cout, int pieces, hipStream_t s, int accumulate, const uint8_t* g_mask,
                                           const float* in_amax, const float* g_amax)
{
    if (pieces != 2 && pieces != 3) return hipErrorInvalidValue;
    const bool f16 = in_amax != nullptr || g_amax != nullptr;            // both words: the two-piece fp16 form
    if (f16 && (pieces != 2 || !in_amax || !g_amax)) return hipErrorInvalidValue;
    if (!conv3x3_wgrad_split_supported(N, Cin, H, W, Cout)) return hipErrorInvalidValue;
    const WgradSplitPlan p = wgrad_split_plan(N, Cin, H, W, Cout);
    float* bias_slab = gb ? workspace + (int64_t)p.ksplit * wgrad_slab_floats(p.CoutP, p.CinP) : nullptr;
    const int blocks = (p.CinP / 64) * (p.CoutP / 64);
    static const bool novec = [] { const char* e = getenv("SSTEM_BF16_NOVEC"); return e && atoi(e) != 0; }();
    static const int runs = [] { const char* e = getenv("SSTEM_WGRAD_RUNS"); return e ? atoi(e) : 2; }();
    const bool vec = !novec && W % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    const int lds = pieces * (SWG_BYTES + SWI_BYTES);
    // the fp16 form with two buffer sets and the waves of a SIMD half a tile apart (SSTEM_WGRAD_PINGPONG=0: one set, two barriers)
    const char* env_pp = getenv("SSTEM_WGRAD_PINGPONG");
    const bool pingpong = env_pp ? atoi(env_pp) != 0 : true;
    hipError_t e;
#define SSTEM_WGRAD_SPLIT(PP, V, M)                                                                                                \
    do {                                                                                                                           \
        static bool done[64] = {};                                                                                                 \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_wgrad_split_mfma<PP, V, M>), lds, done);                         \
        if (e != hipSuccess) return e;                                                                                             \
        hipLaunchKernelGGL((conv3x3_wgrad_split_mfma<PP, V, M>), dim3((unsigned)(blocks * p.ksplit)), dim3(512), lds, s, in, g, workspace, \
                           N, Cin, H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs, g_mask);                    \
    } while (0)
#define SSTEM_WGRAD_SPLIT_F16_PP(V, M, PP)                                                                                         \
    do {                                                                                                                           \
        static bool done[64] = {};                                                                                                 \
        const int ldsp = (PP ? 2 : 1) * lds;                                                                                       \
        e = wgrad_split_lds(reinterpret_cast<const void*>(conv3x3_wgrad_split_mfma<2, V, M, true, PP>), ldsp, done);               \
        if (e != hipSuccess) return e;                                                                                             \
        hipLaunchKernelGGL((conv3x3_wgrad_split_mfma<2, V, M, true, PP>), dim3((unsigned)(blocks * p.ksplit)), dim3(512), ldsp, s, in, g, \
                           workspace, N, Cin, H, W, Cout, p.CinP, p.CoutP, p.ksplit, p.tx, p.ty, bias_slab, runs, g_mask, in_amax, g_amax); \
    } while (0)
#define SSTEM_WGRAD_SPLIT_F16(V, M) do { if (pingpong) SSTEM_WGRAD_SPLIT_F16_PP(V, M, true); else SSTEM_WGRAD_SPLIT_F16_PP(V, M, false); } while (0)
#define SSTEM_WGRAD_SPLIT_PV(PP, V) do { if (g_mask) SSTEM_WGRAD_SPLIT(PP, V, true); else SSTEM_WGRAD_SPLIT(PP, V, false); } while (0)
#if SSTEM_SPLIT_DEV
    return hipErrorInvalidValue;
#else
    if (f16) {
        if (vec) { if (g_mask) SSTEM_WGRAD_SPLIT_F16(true, true); else SSTEM_WGRAD_SPLIT_F16(true, false); }
        else { if (g_mask) SSTEM_WGRAD_SPLIT_F16(false, true); else SSTEM_WGRAD_SPLIT_F16(false, false); }
    }
    else if (pieces == 3) { if (vec) SSTEM_WGRAD_SPLIT_PV(3, true); else SSTEM_WGRAD_SPLIT_PV(3, false); }
    else { if (vec) SSTEM_WGRAD_SPLIT_PV(2, true); else SSTEM_WGRAD_SPLIT_PV(2, false); }
#endif
#undef SSTEM_WGRAD_SPLIT_PV
#undef SSTEM_WGRAD_SPLIT_F16
#undef SSTEM_WGRAD_SPLIT_F16_PP
#undef SSTEM_WGRAD_SPLIT
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_conv3x3_wgrad_reduce(workspace, gw, Cin, Cout, p.CinP, p.CoutP, p.ksplit, bias_slab, gb, p.ksplit, s, accumulate);
}

}  // namespace sstem

#if SSTEM_WGRAD_STAMPS
extern "C" int sstem_debug_wgrad_stamps(unsigned long long* out8)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sstem::g_wgrad_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(sstem::g_wgrad_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : 3;
}
#endif
