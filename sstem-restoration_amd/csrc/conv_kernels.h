// Internal launcher interface between the C-ABI (sstem_capi.hip) and the gfx950 conv kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {

// optional extras of a fused convolution launch (null / zero = absent)
struct ConvExtra {
    const float* residual;   // [N,Cout,Ho,Wo]: out = (act(affine(conv + bias)) + residual) * res_scale
    float res_scale;
    float* bn_part;          // train-mode BatchNorm statistics partials, [Cout][partials per channel][3] = (count, mean, M2)
    int bn_tiles;            // partials per channel (filled in by the launcher)
    const uint8_t* in_mask;  // split ids: [N,Cin,H,W] bytes, an input element counts as 0 where its byte is 0
    uint8_t* out_mask;       // split ids: [N,Cout,H,W] bytes, receives (activation output > 0)
    const float* in_amax;    // SSTEM_CONV_MFMA_F16X3: amax word of the input (1024 floats = sstem_amax_word_floats(), 4 KB, whose maximum bounds |input|)
    float* out_amax;         // split ids: amax word that receives the largest stored magnitude (nullable)
    int f16;                 // split launcher: the two pieces are fp16 (SSTEM_CONV_MFMA_F16X3)
    int out_blocked;         // split launcher: 1 = the output is stored in the row-segment layout [N][H][ceil(W/64)][Cout][64] (sstem_sepconv.h),
                             // 2 = the sub-pixel ConvTranspose store (SSTEM_LAYOUT_CONVT_PARITY)
    int64_t out_img_stride;  // split launcher: floats between the images of the output tensor (0: back to back) -- a channel block of a larger tensor
    float* pool_out;         // split launcher (fp16 id): [N,Cout,H/2,W/2] receives the 2 x 2 pooling of the output as well (nullable)
    int pool_kind;           // 1 = max, 2 = average
};
inline ConvExtra no_extra() { return ConvExtra{nullptr, 1.f, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0}; }

int conv3x3_co_block(int Cout);
int64_t conv3x3_workspace_floats(int Cin, int Cout);
int conv3x3_ksplit(int N, int Cin, int H, int W, int Cout);
int64_t conv3x3_forward_workspace_floats(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv3x3_mfma(const float* in, const float* w, const float* bias, const float* scale,
                               const float* shift, float* out, float* workspace, int64_t workspace_floats, int N,
                               int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                               hipStream_t s, const ConvExtra& ex = no_extra());
int64_t conv3x3_bn_partials(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv2d_direct(const float* in, const float* w, const float* bias, const float* scale,
                                const float* shift, float* out, int N, int Cin, int H, int W, int Cout,
                                int KH, int KW, int PH, int PW, int act, float slope, hipStream_t s);
// 3 x 3 / s1 / p1 with 1, 2, 3, 4, 6 or 8 output channels and W % 4 == 0: the streaming fp32 kernel (out_amax: amax word, nullable)
bool conv3x3_stream_small_supported(int N, int Cin, int H, int W, int Cout);
bool conv3x3_first_u8_supported(int N, int H, int W, int Cout);
hipError_t launch_conv3x3_first_u8(const uint8_t* frames, const float* w, const float* bias, float* out, float* planes, int N, int H, int W,
                                   int Cout, int act, float slope, float* out_amax, hipStream_t s);
hipError_t launch_conv3x3_stream_small(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                       float* out, int N, int Cin, int H, int W, int Cout, int act, float slope, float* out_amax, hipStream_t s);
hipError_t launch_convT3x3s2_direct(const float* in, const float* w, const float* bias, const float* scale,
                                    const float* shift, float* out, int N, int Cin, int H, int W, int Cout,
                                    int act, float slope, hipStream_t s);

hipError_t launch_conv2d_wgrad_direct(const float* in, const float* g, float* gw, int N, int Cin, int H, int W,
                                      int Cout, int KH, int KW, int PH, int PW, hipStream_t s, int accumulate = 0);
hipError_t launch_convT3x3s2_wgrad_direct(const float* in, const float* g, float* gw, int N, int Cin, int H,
                                          int W, int Cout, hipStream_t s);
hipError_t launch_convT3x3s2_dgrad_direct(const float* g, const float* w, float* gin, int N, int Cin, int H,
                                          int W, int Cout, hipStream_t s);

int64_t conv3x3_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv3x3_wgrad_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                     int H, int W, int Cout, hipStream_t s, int accumulate = 0);      // gb: bias gradient [Cout], nullable;
                                                                                                      // accumulate: gw += / gb +=

// conv_bf16_kernels.hip: bf16-operand / fp32-accumulate 3x3 kernels (algorithm id SSTEM_CONV_MFMA_BF16)
bool conv3x3_bf16_supported(int N, int Cin, int H, int W, int Cout);
int conv3x3_bf16_ksplit(int N, int Cin, int H, int W, int Cout);
int64_t conv3x3_bf16_packed_floats(int Cin, int Cout);
int64_t conv3x3_bf16_forward_workspace_floats(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv3x3_bf16_mfma(const float* in, const float* w, const float* bias, const float* scale,
                                    const float* shift, float* out, float* workspace, int64_t workspace_floats, int N,
                                    int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                                    hipStream_t s);

int conv3x3_bf16_co_block(int Cout);

// conv_split_kernels.hip: fp32 convolutions on the bf16 matrix cores, operands split into `pieces` bf16 pieces
// (2: SSTEM_CONV_MFMA_BF16X3, 3: SSTEM_CONV_MFMA_BF16X6)
bool conv3x3_split_supported(int N, int Cin, int H, int W, int Cout);
bool conv3x3_split_f16_supported(int N, int Cin, int H, int W, int Cout);
int conv3x3_split_ksplit(int N, int Cin, int H, int W, int Cout);
int64_t conv3x3_split_packed_floats(int Cin, int Cout, int pieces, int f16 = 0);
int64_t conv3x3_split_forward_workspace_floats(int N, int Cin, int H, int W, int Cout, int pieces, int f16 = 0);
// max |x| of n floats into an amax word (1024 floats = sstem_amax_word_floats(), zeroed by the caller; see conv_split_kernels.hip)
hipError_t launch_amax(const float* x, int64_t n, float* word, hipStream_t s);
hipError_t launch_conv3x3_split_mfma(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                     float* out, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout,
                                     int act, float slope, int w_transposed_flipped, int pieces, hipStream_t s,
                                     const ConvExtra& ex = no_extra());
hipError_t launch_pack_weights_3x3_split_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, int pieces, hipStream_t s);
int64_t conv3x3_wgrad_split_workspace_floats(int N, int Cin, int H, int W, int Cout);
bool conv3x3_wgrad_split_supported(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv3x3_wgrad_split_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                           int H, int W, int Cout, int pieces, hipStream_t s, int accumulate = 0,
                                           const uint8_t* g_mask = nullptr, const float* in_amax = nullptr, const float* g_amax = nullptr);
int64_t pack_group_entry_split(int Cin, int Cout, int pieces, int64_t* out);
hipError_t launch_pack_weights_3x3_split_f16_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s);
int64_t pack_group_entry_split_f16(int Cin, int Cout, int64_t* out);
hipError_t launch_pack_weights_3x3_split_f16_group(const int64_t* table, int n_entries, int64_t total_blocks, int64_t amax_blocks,
                                                   float* bounds, hipStream_t s);
hipError_t launch_pack_weights_3x3_split_group(const int64_t* table, int n_entries, int64_t total_blocks, int pieces, hipStream_t s);

int64_t conv3x3_wgrad_bf16_workspace_floats(int N, int Cin, int H, int W, int Cout);
hipError_t launch_conv3x3_wgrad_bf16_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                          int H, int W, int Cout, hipStream_t s, int accumulate = 0);
hipError_t launch_conv3x3_wgrad_reduce(const float* slabs, float* gw, int Cin, int Cout, int CinP, int CoutP, int ksplit,
                                       const float* bias_slab, float* gb, int bias_rows, hipStream_t s, int accumulate = 0);

// both packings (forward, and transposed + flipped for the data gradient) of one layer's [Cout,Cin,3,3] weights in one launch;
// either destination may be null
hipError_t launch_pack_weights_3x3_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s);
hipError_t launch_pack_weights_3x3_bf16_both(const float* w, float* wp_f, float* wp_t, int Cin, int Cout, hipStream_t s);

// convt_kernels.hip: ConvTranspose2d(k3,s2,p1,op1) by output-parity decomposition on the fp32 matrix cores
int64_t convT3x3s2_forward_workspace_floats(int N, int Cin, int H, int W, int Cout);
int64_t convT3x3s2_dgrad_workspace_floats(int N, int Cin, int H, int W, int Cout);
int64_t convT3x3s2_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout);
int64_t convT3x3s2_bn_partials(int N, int Cin, int H, int W, int Cout);
hipError_t launch_convT3x3s2_mfma(const float* in, const float* w, const float* bias, const float* scale, const float* shift,
                                  float* out, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout,
                                  int act, float slope, int prepacked, hipStream_t s, const ConvExtra& ex = no_extra());
hipError_t launch_convT3x3s2_dgrad_mfma(const float* g, const float* w, float* gin, float* workspace, int64_t workspace_floats,
                                        int N, int Cin, int H, int W, int Cout, hipStream_t s);
hipError_t launch_convT3x3s2_wgrad_mfma(const float* in, const float* g, float* gw, float* gb, float* workspace, int N, int Cin,
                                        int H, int W, int Cout, hipStream_t s, int accumulate);

// one launch packs the weights of many layers (table on the device: 16 int64 per entry, see pack_weights_3x3_group)
int64_t pack_group_entry(int Cin, int Cout, int64_t* out);
int64_t pack_group_entry_bf16(int Cin, int Cout, int64_t* out);
hipError_t launch_pack_weights_3x3_group(const int64_t* table, int n_entries, int64_t total_blocks, hipStream_t s);
hipError_t launch_pack_weights_3x3_bf16_group(const int64_t* table, int n_entries, int64_t total_blocks, hipStream_t s);

bool conv3x3_bf16_io_supported(int N, int Cin, int H, int W, int Cout, int out_bf16);
hipError_t launch_conv3x3_bf16_mfma_io(const void* in, int in_bf16, const float* w, const float* bias, const float* scale,
                                       const float* shift, void* out, int out_bf16, float* workspace, int64_t workspace_floats,
                                       int N, int Cin, int H, int W, int Cout, int act, float slope, int w_transposed_flipped,
                                       hipStream_t s, const uint8_t* in_mask = nullptr, uint8_t* out_mask = nullptr);

hipError_t launch_conv3x3_wgrad_bf16_mfma_in(const void* in, int in_bf16, const float* g, float* gw, float* gb, float* workspace, int N,
                                             int Cin, int H, int W, int Cout, hipStream_t s, int accumulate = 0,
                                             const uint8_t* g_mask = nullptr);

// Weight-gradient slabs (the 3x3 weight-gradient kernels write them, conv3x3_wgrad_reduce adds them): [K slice][co][64-channel block of
// ci][tap][64 ci] -- everything one reduce workgroup reads of a slice, (co, 64 ci, nine taps), is ONE run of 2,304 bytes (as
// [slice][tap][co][ci] it was nine 256-byte rows 4 * CoutP * CinP bytes apart; random 256-byte rows read at well under the rate of 2 KB rows).
__host__ __device__ inline int64_t wgrad_slab_floats(int CoutP, int CinP) { return (int64_t)9 * CoutP * ((CinP + 63) / 64 * 64); }
__host__ __device__ inline int64_t wgrad_slab_index(int ks, int t, int co, int ci, int CoutP, int CinP)
{
    const int cblocks = (CinP + 63) >> 6;
    return ((((int64_t)ks * CoutP + co) * cblocks + (ci >> 6)) * 9 + t) * 64 + (ci & 63);
}

// ---- grouped weight-gradient reduce (round 5) ----------------------------------------------------------------------------------------
// A training step's backward pass runs one slab-reduce launch per 3x3 / ConvTranspose layer (17 of the 288 launches of the 2-sample
// fusion step, 13 us each on slabs of a few hundred KB).  With bit 1 of `accumulate` set (accumulate == 3: add into the gradient sink AND
// defer), launch_conv3x3_wgrad_reduce / the ConvTranspose launcher record the job instead of launching it; wgrad_deferred_flush() runs
// ALL recorded jobs as ONE launch (the per-layer kernels' bodies on the per-layer kernels' workgroup shapes: the same sums in the same
// order, bit for bit).  The slabs must stay alive until the flush; the jobs ride in the kernel's arguments (no table in device memory:
// a captured graph holds them by value).
struct WgradReduceJob {
    const float* slab; float* gw; const float* bias_slab; float* gb;
    int Cin, Cout, CinP, CoutP, ksplit, bias_rows, wblocks, bblocks, ngroups, kind, accumulate, block0;     // kind 0: 3x3, nine taps per workgroup; 1: three; 2: ConvTranspose
};
constexpr int WGRAD_GROUP_MAX_JOBS = 40;               // 40 x 80 bytes of kernel arguments
struct WgradReduceGroup { int n; int pad; WgradReduceJob job[WGRAD_GROUP_MAX_JOBS]; };
void wgrad_defer(const WgradReduceJob& j);
int wgrad_deferred_count();
void wgrad_deferred_drop();
hipError_t wgrad_deferred_flush(hipStream_t s);

// the ConvTranspose reduce of one (tap, co, 64 ci) row block: shared by convT_wgrad_reduce (convt_kernels.hip) and the grouped launch
__device__ inline void convT_wgrad_reduce_body(const float* __restrict__ slab, float* __restrict__ gw, int Cin, int Cout, int CinP, int CoutP,
                                               int nslices, const float* __restrict__ bias_slab, float* __restrict__ gb, int bias_rows,
                                               int wblocks, int accumulate, int blk0, int ngroups, float (*part)[64])
{
    const int e = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const bool on = kg < ngroups;
    auto combine = [&]() -> float {
        float v = part[0][e];
        for (int k = 1; k < ngroups; ++k) v += part[k][e];
        return v;
    };
    if (blk0 >= wblocks) {
        const int co = (blk0 - wblocks) * 64 + e;
        float s = 0.f;
        if (on && co < CoutP) {
#pragma unroll 4
            for (int r = kg; r < bias_rows; r += ngroups) s += bias_slab[(int64_t)r * CoutP + co];
        }
        if (on) part[kg][e] = s;
        __syncthreads();
        if (kg == 0 && co < Cout) {
            const float v = combine();
            gb[co] = accumulate ? gb[co] + v : v;
        }
        return;
    }
    const int64_t rows = (int64_t)9 * CoutP;
    const int cblocks = (CinP + 63) / 64;
    const int64_t slice = rows * CinP;
    for (int64_t blk = blk0; blk < rows * cblocks; blk += wblocks) {
        const int64_t row = blk / cblocks;
        const int ci = (int)(blk % cblocks) * 64 + e;
        const int t = (int)(row / CoutP), co = (int)(row % CoutP);
        float s = 0.f;
        if (on && ci < CinP) {
            const float* p = slab + row * CinP + ci;
#pragma unroll 4
            for (int k = kg; k < nslices; k += ngroups) s += p[(int64_t)k * slice];
        }
        if (on) part[kg][e] = s;
        __syncthreads();
        if (kg == 0 && ci < Cin && co < Cout) {
            const float v = combine();
            float* dst = gw + ((int64_t)ci * Cout + co) * 9 + t;
            *dst = accumulate ? *dst + v : v;
        }
        __syncthreads();
    }
}
}  // namespace sstem
