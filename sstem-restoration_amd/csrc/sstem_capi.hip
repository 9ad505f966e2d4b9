// C-ABI of libsstem_hip.so -- see include/sstem_sepconv.h for the contract and the reference
// interfaces (libs/sepconv/src/SeparableConvolution_cuda.{h,c}) each entry point replaces.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sstem_sepconv.h"
#include "../../include/sstem_conv.h"
#include "../../include/sstem_warp.h"
#include "../../include/sstem_io.h"
#include "../../include/sstem_resize.h"
#include "../../include/sstem_norm.h"
#include "sepconv_kernels.h"
#include "conv_kernels.h"
#include "warp_kernels.h"
#include "misc_kernels.h"
#include "norm_kernels.h"
#include <math.h>

namespace {

thread_local char g_last_error[512] = "";

int fail(int status, const char* fmt, const char* a = "", const char* b = "")
{
    snprintf(g_last_error, sizeof(g_last_error), fmt, a, b);
    return status;
}

int hip_fail(const char* where, hipError_t e)
{
    return fail(SSTEM_ERR_HIP, "%s: %s", where, hipGetErrorString(e));
}

// all element counts of a call must be addressable in int64 (the reference used 32-bit int,
// kernel.cu:26,32 -- 64-bit here on purpose: B=64 x 51 x 1024^2 already exceeds 2^31).
bool sizes_ok(int64_t B, int64_t C, int64_t H, int64_t W)
{
    if (B < 0 || C < 0 || H < 0 || W < 0) return false;
    const int64_t lim = (int64_t)1 << 40;  // 1 Ti elements: far above 288 GB of fp32
    if (B > lim || C > lim || H > lim || W > lim) return false;
    const __int128 hw = (__int128)(H + 50) * (W + 50);
    const __int128 big = (__int128)B * (C > 51 ? C : 51) * hw;
    return big < ((__int128)1 << 46);
}

}  // namespace

// weight-gradient launches: bit 0 = add into the gradient buffers, bit 1 (only with bit 0) = defer the slab reduce to
// sstem_wgrad_deferred_flush (include/sstem_conv.h)
static inline int wgrad_flags(int accumulate) { return accumulate == 3 ? 3 : (accumulate ? 1 : 0); }

extern "C" {

int sstem_wgrad_deferred_count(void) { return sstem::wgrad_deferred_count(); }
void sstem_wgrad_deferred_drop(void) { sstem::wgrad_deferred_drop(); }
int sstem_wgrad_deferred_flush(void* stream)
{
    hipError_t e = sstem::wgrad_deferred_flush(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("grouped weight-gradient reduce launch", e);
    return SSTEM_OK;
}

int sstem_version(void) { return 100; }  // 0.1.0

const char* sstem_status_string(int status)
{
    switch (status) {
        case SSTEM_OK: return "ok";
        case SSTEM_ERR_NULL_POINTER: return "null pointer argument";
        case SSTEM_ERR_BAD_SHAPE: return "bad shape";
        case SSTEM_ERR_UNSUPPORTED: return "unsupported configuration";
        case SSTEM_ERR_HIP: return "HIP runtime error";
        case SSTEM_ERR_NO_DEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

const char* sstem_last_error(void) { return g_last_error; }

int64_t sstem_sepconv_forward_bytes(int64_t B, int64_t C, int64_t H, int64_t W)
{
    return 4 * (B * C * (H + 50) * (W + 50) + 2 * B * 51 * H * W + B * C * H * W);
}

int64_t sstem_sepconv_backward_bytes(int64_t B, int64_t C, int64_t H, int64_t W)
{
    return 4 * (B * C * H * W + B * C * (H + 50) * (W + 50) + 4 * B * 51 * H * W);
}

int sstem_sepconv_forward_f32_algo(const float* input, const float* vertical,
                                   const float* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W,
                                   void* stream, int algo)
{
    if (!sizes_ok(B, C, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "forward: negative or oversized shape");
    if (B == 0 || C == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !vertical || !horizontal || !output)
        return fail(SSTEM_ERR_NULL_POINTER, "forward: null tensor pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (algo == SSTEM_SEPCONV_AUTO)
        algo = sstem::mfma_grid_ok(B, H, W) ? SSTEM_SEPCONV_MFMA : SSTEM_SEPCONV_DIRECT;
    if (algo == SSTEM_SEPCONV_MFMA) {
        if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "forward: grid too large for the MFMA kernel");
        e = sstem::launch_fwd_mfma(input, vertical, horizontal, output, B, C, H, W, s);
    } else if (algo == SSTEM_SEPCONV_DIRECT) {
        e = sstem::launch_fwd_direct(input, vertical, horizontal, output, B, C, H, W,
                                     SSTEM_SEPCONV_FILTER, s);
    } else {
        return fail(SSTEM_ERR_UNSUPPORTED, "forward: unknown algorithm id");
    }
    if (e != hipSuccess) return hip_fail("sepconv forward launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_forward_f32(const float* input, const float* vertical,
                              const float* horizontal, float* output,
                              int64_t B, int64_t C, int64_t H, int64_t W, void* stream)
{
    return sstem_sepconv_forward_f32_algo(input, vertical, horizontal, output, B, C, H, W,
                                          stream, SSTEM_SEPCONV_AUTO);
}

int sstem_sepconv_interp_apply_f32(const float* i1, const float* i2,
                                   const float* k1v, const float* k1h,
                                   const float* k2v, const float* k2h, float* output,
                                   int64_t B, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "interp apply: negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!i1 || !i2 || !k1v || !k1h || !k2v || !k2h || !output)
        return fail(SSTEM_ERR_NULL_POINTER, "interp apply: null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "interp apply: grid too large");
    hipError_t e = sstem::launch_interp_fused(i1, i2, k1v, k1h, k2v, k2h, output, B, H, W,
                                              static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("interp apply launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_interp_apply_gray_f32(const float* g1, const float* g2,
                                        const float* k1v, const float* k1h,
                                        const float* k2v, const float* k2h, float* output,
                                        int64_t B, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "gray interp apply: negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!g1 || !g2 || !k1v || !k1h || !k2v || !k2h || !output)
        return fail(SSTEM_ERR_NULL_POINTER, "gray interp apply: null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "gray interp apply: grid too large");
    if (!sstem::interp_fused_gray_ok(H, W))
        return fail(SSTEM_ERR_UNSUPPORTED, "gray interp apply: 51*H*W*4 bytes per image must stay below 4 GiB "
                                           "(use sstem_sepconv_interp_apply_f32 on the replicated frames)");
    hipError_t e = sstem::launch_interp_fused_gray(g1, g2, k1v, k1h, k2v, k2h, output, B, H, W,
                                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("gray interp apply launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_interp_apply_gray_supported(int64_t B, int64_t H, int64_t W)
{
    return (sizes_ok(B, 3, H, W) && B > 0 && H > 0 && W > 0 && sstem::mfma_grid_ok(B, H, W) && sstem::interp_fused_gray_ok(H, W)) ? 1 : 0;
}

int64_t sstem_sepconv_coef_blocked_floats(int64_t B, int64_t H, int64_t W)
{
    if (!sizes_ok(B, 3, H, W)) return 0;
    return sstem::coef_blocked_floats(B, H, W);
}

int sstem_sepconv_coef_to_blocked_f32(const float* coef, float* blocked, int64_t B, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "coef to blocked: negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!coef || !blocked) return fail(SSTEM_ERR_NULL_POINTER, "coef to blocked: null tensor pointer");
    hipError_t e = sstem::launch_coef_to_blocked(coef, blocked, B, H, W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("coef to blocked launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_interp_apply_gray_blocked_supported(int64_t B, int64_t H, int64_t W)
{
    return (sizes_ok(B, 3, H, W) && B > 0 && H > 0 && W > 0 && sstem::mfma_grid_ok(B, H, W) &&
            sstem::interp_fused_gray_blocked_ok(H, W)) ? 1 : 0;
}

int sstem_sepconv_interp_apply_gray_blocked_f32(const float* g1, const float* g2,
                                                const float* k1v, const float* k1h,
                                                const float* k2v, const float* k2h, float* output,
                                                int64_t B, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "blocked gray interp apply: negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!g1 || !g2 || !k1v || !k1h || !k2v || !k2h || !output)
        return fail(SSTEM_ERR_NULL_POINTER, "blocked gray interp apply: null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "blocked gray interp apply: grid too large");
    if (!sstem::interp_fused_gray_blocked_ok(H, W))
        return fail(SSTEM_ERR_UNSUPPORTED, "blocked gray interp apply: one image's blocked coefficients must stay below 4 GiB");
    hipError_t e = sstem::launch_interp_fused_gray_blocked(g1, g2, k1v, k1h, k2v, k2h, output, B, H, W,
                                                           static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("blocked gray interp apply launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_interp_apply_gray_u8_f32(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                           const float* k2v, const float* k2h, float* output, uint8_t* output_u8,
                                           int64_t B, int64_t H, int64_t W, int blocked_coefficients, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "gray interp apply (uint8 store): negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!g1 || !g2 || !k1v || !k1h || !k2v || !k2h || !output || !output_u8)
        return fail(SSTEM_ERR_NULL_POINTER, "gray interp apply (uint8 store): null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "gray interp apply (uint8 store): grid too large");
    if (blocked_coefficients ? !sstem::interp_fused_gray_blocked_ok(H, W) : !sstem::interp_fused_gray_ok(H, W))
        return fail(SSTEM_ERR_UNSUPPORTED, "gray interp apply (uint8 store): one image's coefficients must stay below 4 GiB");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = blocked_coefficients ? sstem::launch_interp_fused_gray_blocked(g1, g2, k1v, k1h, k2v, k2h, output, B, H, W, s, output_u8)
                                        : sstem::launch_interp_fused_gray(g1, g2, k1v, k1h, k2v, k2h, output, B, H, W, s, output_u8);
    if (e != hipSuccess) return hip_fail("gray interp apply (uint8 store) launch", e);
    return SSTEM_OK;
}

int64_t sstem_sepconv_interp_apply_bytes(int64_t B, int64_t H, int64_t W, int frame_planes)
{
    return 4 * (2 * B * frame_planes * H * W + 4 * B * 51 * H * W + B * H * W);
}

int sstem_sepconv_backward_f32_algo(const float* grad_output, const float* input,
                                    const float* vertical, const float* horizontal,
                                    float* grad_input, float* grad_vertical,
                                    float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W,
                                    void* stream, int algo)
{
    (void)grad_input;  // never written: kernel.cu:152-206 of the reference does not touch it either
    if (!sizes_ok(B, C, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "backward: negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (C > 3)
        return fail(SSTEM_ERR_UNSUPPORTED,
                    "backward: C > 3 (the reference kernels hard-code three channels, kernel.cu:100-108)");
    if (!vertical || !horizontal || !grad_vertical || !grad_horizontal || (C > 0 && (!grad_output || !input)))
        return fail(SSTEM_ERR_NULL_POINTER, "backward: null tensor pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C == 0) {  // sum over zero channels
        const size_t bytes = (size_t)B * 51 * H * W * sizeof(float);
        hipError_t e = hipMemsetAsync(grad_vertical, 0, bytes, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_horizontal, 0, bytes, s);
        if (e != hipSuccess) return hip_fail("sepconv backward memset", e);
        return SSTEM_OK;
    }
    hipError_t e;
    if (algo == SSTEM_SEPCONV_AUTO)
        algo = sstem::mfma_grid_ok(B, H, W) ? SSTEM_SEPCONV_MFMA : SSTEM_SEPCONV_DIRECT;
    if (algo == SSTEM_SEPCONV_MFMA) {
        if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "backward: grid too large for the MFMA kernel");
        e = sstem::launch_bwd_mfma(grad_output, input, vertical, horizontal, grad_vertical,
                                   grad_horizontal, B, C, H, W, s);
    } else if (algo == SSTEM_SEPCONV_DIRECT) {
        e = sstem::launch_bwd_direct(grad_output, input, vertical, horizontal, grad_vertical,
                                     grad_horizontal, B, C, H, W, SSTEM_SEPCONV_FILTER, s);
    } else {
        return fail(SSTEM_ERR_UNSUPPORTED, "backward: unknown algorithm id");
    }
    if (e != hipSuccess) return hip_fail("sepconv backward launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_backward_f32(const float* grad_output, const float* input,
                               const float* vertical, const float* horizontal,
                               float* grad_input, float* grad_vertical, float* grad_horizontal,
                               int64_t B, int64_t C, int64_t H, int64_t W, void* stream)
{
    return sstem_sepconv_backward_f32_algo(grad_output, input, vertical, horizontal, grad_input,
                                           grad_vertical, grad_horizontal, B, C, H, W, stream,
                                           SSTEM_SEPCONV_AUTO);
}

// ---- dense convolution blocks (include/sstem_conv.h) -------------------------------------------
int64_t sstem_conv3x3_workspace_floats(int64_t Cin, int64_t Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    return sstem::conv3x3_workspace_floats((int)Cin, (int)Cout);
}

static bool conv_sizes_ok(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout)
{
    if (N < 0 || Cin < 0 || H < 0 || W < 0 || Cout < 0) return false;
    const int64_t lim = (int64_t)1 << 30;
    if (N > lim || Cin > lim || Cout > lim || H > lim || W > lim) return false;
    if (H * W >= ((int64_t)1 << 31)) return false;               // in-plane offsets are 32-bit
    if ((__int128)N * (Cin > Cout ? Cin : Cout) * H * W >= ((__int128)1 << 46)) return false;
    return true;
}

int64_t sstem_conv3x3_forward_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || Cin <= 0 || Cout <= 0) return 0;
    return sstem::conv3x3_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
}

// pieces per operand of the split ids (0: not a split id)
static inline int split_pieces_of(int algo) { return algo == SSTEM_CONV_MFMA_BF16X3 ? 2 : algo == SSTEM_CONV_MFMA_BF16X6 ? 3 : 0; }
// ... and of the scaled-forward entry's ids (the fp16 id has two)
static inline int scaled_pieces_of(int algo) { return algo == SSTEM_CONV_MFMA_F16X3 ? 2 : split_pieces_of(algo); }

int64_t sstem_conv3x3_packed_floats(int64_t Cin, int64_t Cout, int algo)
{
    if (Cin <= 0 || Cout <= 0 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    if (algo == SSTEM_CONV_MFMA_BF16) return sstem::conv3x3_bf16_packed_floats((int)Cin, (int)Cout);
    if (split_pieces_of(algo)) return sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, split_pieces_of(algo));
    if (algo == SSTEM_CONV_MFMA_F16X3) return sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, 2, 1);
    if (algo == SSTEM_CONV_MFMA) return sstem::conv3x3_workspace_floats((int)Cin, (int)Cout);
    return 0;
}

int sstem_conv3x3_pack_weights_f32(const float* weight, int64_t Cin, int64_t Cout, int algo, float* packed_forward,
                                   float* packed_transposed, void* stream)
{
    if (Cin <= 0 || Cout <= 0 || Cin > (1 << 20) || Cout > (1 << 20)) return fail(SSTEM_ERR_BAD_SHAPE, "pack_weights: bad shape");
    if (!weight) return fail(SSTEM_ERR_NULL_POINTER, "pack_weights: null weight");
    if (!packed_forward && !packed_transposed) return SSTEM_OK;
    hipError_t e;
    if (algo == SSTEM_CONV_MFMA_BF16)
        e = sstem::launch_pack_weights_3x3_bf16_both(weight, packed_forward, packed_transposed, (int)Cin, (int)Cout, static_cast<hipStream_t>(stream));
    else if (split_pieces_of(algo))
        e = sstem::launch_pack_weights_3x3_split_both(weight, packed_forward, packed_transposed, (int)Cin, (int)Cout, split_pieces_of(algo),
                                                      static_cast<hipStream_t>(stream));
    else if (algo == SSTEM_CONV_MFMA_F16X3)
        e = sstem::launch_pack_weights_3x3_split_f16_both(weight, packed_forward, packed_transposed, (int)Cin, (int)Cout,
                                                          static_cast<hipStream_t>(stream));
    else if (algo == SSTEM_CONV_MFMA)
        e = sstem::launch_pack_weights_3x3_both(weight, packed_forward, packed_transposed, (int)Cin, (int)Cout, static_cast<hipStream_t>(stream));
    else
        return fail(SSTEM_ERR_UNSUPPORTED, "pack_weights: an explicit MFMA algorithm id is needed (SSTEM_CONV_MFMA or _MFMA_BF16)");
    if (e != hipSuccess) return hip_fail("pack_weights launch", e);
    return SSTEM_OK;
}

int64_t sstem_conv3x3_pack_group_entry(int64_t Cin, int64_t Cout, int algo, int64_t* entry16)
{
    if (!entry16 || Cin <= 0 || Cout <= 0 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    if (algo == SSTEM_CONV_MFMA_BF16) return sstem::pack_group_entry_bf16((int)Cin, (int)Cout, entry16);
    if (split_pieces_of(algo)) return sstem::pack_group_entry_split((int)Cin, (int)Cout, split_pieces_of(algo), entry16);
    if (algo == SSTEM_CONV_MFMA_F16X3) return sstem::pack_group_entry_split_f16((int)Cin, (int)Cout, entry16);
    if (algo == SSTEM_CONV_MFMA) return sstem::pack_group_entry((int)Cin, (int)Cout, entry16);
    return 0;
}

int sstem_conv3x3_pack_weights_group_f32(const int64_t* table, int64_t n_entries, int64_t total_blocks, int algo, void* stream)
{
    if (n_entries < 0 || total_blocks < 0 || n_entries > (1 << 20)) return fail(SSTEM_ERR_BAD_SHAPE, "pack group: bad counts");
    if (n_entries == 0) return SSTEM_OK;
    if (!table) return fail(SSTEM_ERR_NULL_POINTER, "pack group: null table");
    hipError_t e;
    if (algo == SSTEM_CONV_MFMA_BF16)
        e = sstem::launch_pack_weights_3x3_bf16_group(table, (int)n_entries, total_blocks, static_cast<hipStream_t>(stream));
    else if (split_pieces_of(algo))
        e = sstem::launch_pack_weights_3x3_split_group(table, (int)n_entries, total_blocks, split_pieces_of(algo), static_cast<hipStream_t>(stream));
    else if (algo == SSTEM_CONV_MFMA)
        e = sstem::launch_pack_weights_3x3_group(table, (int)n_entries, total_blocks, static_cast<hipStream_t>(stream));
    else
        return fail(SSTEM_ERR_UNSUPPORTED, "pack group: an explicit MFMA algorithm id is needed");
    if (e != hipSuccess) return hip_fail("pack group launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_pack_weights_group_f16(const int64_t* table, int64_t n_entries, int64_t total_blocks, int64_t bound_blocks, float* bounds,
                                         void* stream)
{
    if (n_entries < 0 || total_blocks < 0 || bound_blocks < 0 || n_entries > (1 << 20)) return fail(SSTEM_ERR_BAD_SHAPE, "pack group f16: bad counts");
    if (n_entries == 0) return SSTEM_OK;
    if (!table || !bounds) return fail(SSTEM_ERR_NULL_POINTER, "pack group f16: null table / bounds");
    const hipError_t e = sstem::launch_pack_weights_3x3_split_f16_group(table, (int)n_entries, total_blocks, bound_blocks, bounds,
                                                                        static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("pack group f16 launch", e);
    return SSTEM_OK;
}

int64_t sstem_conv3x3_forward_workspace_floats_algo(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || Cin <= 0 || Cout <= 0) return 0;
    if (algo == SSTEM_CONV_MFMA_BF16)
        return sstem::conv3x3_bf16_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if (split_pieces_of(algo))
        return sstem::conv3x3_split_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout, split_pieces_of(algo));
    if (algo == SSTEM_CONV_MFMA_F16X3)
        return sstem::conv3x3_split_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout, 2, 1);
    if (algo == SSTEM_CONV_DIRECT) return 0;
    return sstem::conv3x3_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
}

int sstem_conv2d_forward_f32(const float* input, const float* weight, const float* bias,
                             const float* scale, const float* shift, float* output,
                             float* workspace, int64_t workspace_floats,
                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                             int KH, int KW, int pad_h, int pad_w, int weight_transposed,
                             int act, float slope, void* stream, int algo)
{
    return sstem_conv2d_forward_ex_f32(input, weight, bias, scale, shift, nullptr, 1.f, output, nullptr, workspace, workspace_floats,
                                       N, Cin, H, W, Cout, KH, KW, pad_h, pad_w, weight_transposed, act, slope, stream, algo);
}

int64_t sstem_conv_bn_partials(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int KH, int KW, int transposed, int algo)
{
    if (!conv_sizes_ok(N, Cin, transposed ? 2 * H : H, transposed ? 2 * W : W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    if (KH != 3 || KW != 3) return 0;
    if (algo == SSTEM_CONV_AUTO) algo = (N * ((Cout + 31) / 32) < 65536) ? SSTEM_CONV_MFMA : SSTEM_CONV_DIRECT;
    if (algo != SSTEM_CONV_MFMA) return 0;
    if (transposed) return sstem::convT3x3s2_bn_partials((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    return sstem::conv3x3_bn_partials((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
}

int sstem_conv2d_forward_ex_f32(const float* input, const float* weight, const float* bias,
                                const float* scale, const float* shift, const float* residual, float residual_scale,
                                float* output, float* bn_partials, float* workspace, int64_t workspace_floats,
                                int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                int KH, int KW, int pad_h, int pad_w, int weight_transposed,
                                int act, float slope, void* stream, int algo)
{
    const sstem::ConvExtra ex{residual, residual_scale, bn_partials, 0};
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || KH <= 0 || KW <= 0 || pad_h < 0 || pad_w < 0)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv2d: bad shape");
    if (2 * pad_h != KH - 1 || 2 * pad_w != KW - 1)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: only stride-1 'same' padding is supported");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: unknown activation id");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv2d: null tensor pointer");
    const bool is3x3 = (KH == 3 && KW == 3);
    if (weight_transposed && !is3x3) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: transposed weights need 3x3");
    if (weight_transposed < 0 || weight_transposed > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: unknown weight_transposed flags");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (algo == SSTEM_CONV_AUTO) algo = (is3x3 && Cin > 0 && N * ((Cout + 31) / 32) < 65536) ? SSTEM_CONV_MFMA : SSTEM_CONV_DIRECT;
    hipError_t e;
    if (algo == SSTEM_CONV_MFMA) {
        if (!is3x3 || Cin == 0) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: the MFMA kernel is 3x3/s1/p1 only");
        const int64_t need = sstem::conv3x3_workspace_floats((int)Cin, (int)Cout);
        if (!workspace || workspace_floats < need)
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d: workspace too small (see sstem_conv3x3_workspace_floats)");
        if (bn_partials && (scale || shift || act != 0 || residual))
            return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: bn_partials are the statistics of the raw conv + bias output (no affine, activation or residual)");
        if (bn_partials && workspace_floats < sstem::conv3x3_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d: bn_partials need the full workspace (sstem_conv3x3_forward_workspace_floats)");
        e = sstem::launch_conv3x3_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N,
                                       (int)Cin, (int)H, (int)W, (int)Cout, act, slope, weight_transposed & 3, s, ex);
    } else if (split_pieces_of(algo)) {
        const int pieces = split_pieces_of(algo);
        if (!is3x3 || Cin == 0) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: the split-bf16 MFMA kernel is 3x3/s1/p1 only");
        if (bn_partials) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: bn_partials need the fp32 3x3 MFMA kernel (SSTEM_CONV_MFMA)");
        if (!sstem::conv3x3_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
            return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: the split-bf16 MFMA kernel needs a channel plane (W % 4 == 0) or a whole input image below 2 GiB");
        if (!workspace || workspace_floats < sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, pieces))
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
        e = sstem::launch_conv3x3_split_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N, (int)Cin,
                                             (int)H, (int)W, (int)Cout, act, slope, weight_transposed & 3, pieces, s, ex);
    } else if (residual || bn_partials) {
        return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: residual / bn_partials need the fp32 3x3 MFMA kernel (SSTEM_CONV_MFMA)");
    } else if (algo == SSTEM_CONV_MFMA_BF16) {
        if (!is3x3 || Cin == 0) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: the bf16 MFMA kernel is 3x3/s1/p1 only");
        if (!sstem::conv3x3_bf16_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
            return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: the bf16 MFMA kernel needs a channel plane (W % 4 == 0) or a whole input image below 2 GiB");
        if (!workspace || workspace_floats < sstem::conv3x3_bf16_packed_floats((int)Cin, (int)Cout))
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
        e = sstem::launch_conv3x3_bf16_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N,
                                            (int)Cin, (int)H, (int)W, (int)Cout, act, slope, weight_transposed & 3, s);
    } else if (algo == SSTEM_CONV_DIRECT) {
        if (weight_transposed) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: direct kernel takes plain [Cout,Cin,KH,KW] weights only");
        e = sstem::launch_conv2d_direct(input, weight, bias, scale, shift, output, (int)N, (int)Cin, (int)H,
                                        (int)W, (int)Cout, KH, KW, pad_h, pad_w, act, slope, s);
    } else {
        return fail(SSTEM_ERR_UNSUPPORTED, "conv2d: unknown algorithm id");
    }
    if (e != hipSuccess) return hip_fail("conv2d launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_forward_masked_f32(const float* input, const uint8_t* input_mask, const float* weight, const float* bias,
                                     const float* scale, const float* shift, float* output, uint8_t* output_mask,
                                     float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                     int64_t Cout, int weight_flags, int act, float slope, void* stream, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || Cin <= 0) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 masked: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 masked: unknown activation id");
    if (weight_flags < 0 || weight_flags > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 masked: unknown weight flags");
    const int pieces = split_pieces_of(algo);
    if (!pieces) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 masked: a split-bf16 id is needed (SSTEM_CONV_MFMA_BF16X6 / _BF16X3)");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 masked: null tensor pointer");
    if (!sstem::conv3x3_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 masked: outside the split kernel's range");
    if (!workspace || workspace_floats < sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, pieces))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 masked: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
    const sstem::ConvExtra ex{nullptr, 1.f, nullptr, 0, input_mask, output_mask};
    const hipError_t e = sstem::launch_conv3x3_split_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N,
                                                          (int)Cin, (int)H, (int)W, (int)Cout, act, slope, weight_flags,
                                                          pieces, static_cast<hipStream_t>(stream), ex);
    if (e != hipSuccess) return hip_fail("conv3x3 masked launch", e);
    return SSTEM_OK;
}

int64_t sstem_amax_word_floats(void) { return 1024; }

int sstem_amax_f32(const float* x, int64_t n, float* word, void* stream)
{
    if (n < 0) return fail(SSTEM_ERR_BAD_SHAPE, "amax: negative count");
    if (n == 0) return SSTEM_OK;
    if (!x || !word) return fail(SSTEM_ERR_NULL_POINTER, "amax: null pointer");
    const hipError_t e = sstem::launch_amax(x, n, word, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("amax launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_forward_scaled_f32(const float* input, const float* input_amax, const float* weight, const float* bias,
                                     const float* scale, const float* shift, const float* residual, float residual_scale,
                                     float* output, float* output_amax, float* workspace, int64_t workspace_floats,
                                     int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                     void* stream, int algo, int output_layout)
{
    return sstem_conv3x3_forward_scaled_strided_f32(input, input_amax, weight, bias, scale, shift, residual, residual_scale, output, output_amax,
                                                    workspace, workspace_floats, N, Cin, H, W, Cout, weight_flags, act, slope, stream, algo,
                                                    output_layout, 0, nullptr, SSTEM_POOL_NONE);
}

int sstem_conv3x3_forward_scaled_strided_f32(const float* input, const float* input_amax, const float* weight, const float* bias,
                                             const float* scale, const float* shift, const float* residual, float residual_scale,
                                             float* output, float* output_amax, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                             void* stream, int algo, int output_layout, int64_t output_image_stride, float* pooled_output,
                                             int pool_kind)
{
    if (pooled_output && (pool_kind != SSTEM_POOL_MAX && pool_kind != SSTEM_POOL_AVG))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: unknown pooling kind");
    if (pooled_output && (algo != SSTEM_CONV_MFMA_F16X3 || output_layout != SSTEM_LAYOUT_NCHW || residual || H % 8 != 0 || W % 32 != 0))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: a pooled copy needs SSTEM_CONV_MFMA_F16X3, the NCHW store without residual, H % 8 == 0 and W % 32 == 0");
    if (output_image_stride != 0 && (output_layout == SSTEM_LAYOUT_ROW_SEGMENTS || output_image_stride < Cout * H * W))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: an output image stride is for NCHW / sub-pixel stores and cannot be smaller than one image");
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || Cin <= 0) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 scaled: bad shape");
    if (output_layout != SSTEM_LAYOUT_NCHW && output_layout != SSTEM_LAYOUT_ROW_SEGMENTS && output_layout != SSTEM_LAYOUT_CONVT_PARITY)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: unknown output layout");
    if (output_layout == SSTEM_LAYOUT_CONVT_PARITY &&
        (algo != SSTEM_CONV_MFMA_F16X3 || Cout % 128 != 0 || Cin % 16 != 0 || W % 4 != 0 || Cout * H * W * 4 >= ((int64_t)1 << 32)))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: the sub-pixel ConvTranspose store needs SSTEM_CONV_MFMA_F16X3, Cout = 4 C with C a multiple "
                                           "of 32, Cin a multiple of 16, W a multiple of 4 and one output image below 4 GiB");
    if (output_layout == SSTEM_LAYOUT_ROW_SEGMENTS && (residual || H * ((W + 63) / 64) * Cout * 256 >= ((int64_t)1 << 32)))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: the row-segment output takes no residual and one image of it must stay below 4 GiB");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: unknown activation id");
    if (weight_flags < 0 || weight_flags > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: unknown weight flags");
    if (algo == SSTEM_CONV_DIRECT) {
        // the streaming fp32 kernel for a handful of output channels: exact fp32 products, so no input bound is read; it leaves the
        // output's bound like the split ids do (a link of an fp16 chain)
        if (residual || output_layout != SSTEM_LAYOUT_NCHW || pooled_output || weight_flags != 0 ||
            (output_image_stride != 0 && output_image_stride != Cout * H * W))
            return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: SSTEM_CONV_DIRECT takes plain weights and stores plain NCHW (no residual, pooled copy or image stride)");
        if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
        if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 scaled: null tensor pointer");
        if (!sstem::conv3x3_stream_small_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
            return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: SSTEM_CONV_DIRECT here is the streaming kernel (sstem_conv3x3_stream_small_supported)");
        const hipError_t e = sstem::launch_conv3x3_stream_small(input, weight, bias, scale, shift, output, (int)N, (int)Cin, (int)H, (int)W,
                                                                (int)Cout, act, slope, output_amax, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail("conv3x3 scaled launch (streaming kernel)", e);
        return SSTEM_OK;
    }
    const int pieces = scaled_pieces_of(algo);
    const int f16 = algo == SSTEM_CONV_MFMA_F16X3 ? 1 : 0;
    if (!pieces) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: a split id (SSTEM_CONV_MFMA_F16X3 / _BF16X6 / _BF16X3) or SSTEM_CONV_DIRECT is needed");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || (!output && !pooled_output)) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 scaled: null tensor pointer");
    if (f16 && !input_amax) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 scaled: SSTEM_CONV_MFMA_F16X3 needs the input's amax word (sstem_amax_f32)");
    if (!sstem::conv3x3_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout) ||
        (f16 && !sstem::conv3x3_split_f16_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout)))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled: outside the split kernel's range (sstem_conv3x3_algo_supported)");
    if (!workspace || workspace_floats < sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, pieces, f16))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 scaled: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
    const sstem::ConvExtra ex{residual, residual_scale, nullptr, 0, nullptr, nullptr, input_amax, output_amax, f16,
                              output_layout == SSTEM_LAYOUT_ROW_SEGMENTS ? 1 : (output_layout == SSTEM_LAYOUT_CONVT_PARITY ? 2 : 0),
                              output_image_stride == Cout * H * W ? 0 : output_image_stride, pooled_output, pooled_output ? pool_kind : 0};
    const hipError_t e = sstem::launch_conv3x3_split_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N,
                                                          (int)Cin, (int)H, (int)W, (int)Cout, act, slope, weight_flags,
                                                          pieces, static_cast<hipStream_t>(stream), ex);
    if (e != hipSuccess) return hip_fail("conv3x3 scaled launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_forward_scaled_masked_f32(const float* input, const float* input_amax, const uint8_t* input_mask, const float* weight,
                                            const float* bias, const float* scale, const float* shift, float* output, float* output_amax,
                                            uint8_t* output_mask, float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin,
                                            int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope, void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || Cin <= 0) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 scaled masked: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled masked: unknown activation id");
    if (weight_flags < 0 || weight_flags > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled masked: unknown weight flags");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !input_amax || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 scaled masked: null tensor pointer");
    if ((input_mask || output_mask) && (W % 4 != 0 || (reinterpret_cast<uintptr_t>(input) & 15) != 0))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled masked: the masked fp16 instances take W % 4 == 0 and a 16-byte aligned input");
    if (!sstem::conv3x3_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout) ||
        !sstem::conv3x3_split_f16_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 scaled masked: outside the split kernel's range (sstem_conv3x3_algo_supported)");
    if (!workspace || workspace_floats < sstem::conv3x3_split_packed_floats((int)Cin, (int)Cout, 2, 1))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 scaled masked: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
    const sstem::ConvExtra ex{nullptr, 1.f, nullptr, 0, input_mask, output_mask, input_amax, output_amax, 1, 0, 0, nullptr, 0};
    const hipError_t e = sstem::launch_conv3x3_split_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N,
                                                          (int)Cin, (int)H, (int)W, (int)Cout, act, slope, weight_flags, 2,
                                                          static_cast<hipStream_t>(stream), ex);
    if (e != hipSuccess) return hip_fail("conv3x3 scaled masked launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_backward_weight_masked_f32(const float* input, const float* grad_output, const uint8_t* grad_mask, float* grad_weight,
                                             float* grad_bias, float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin,
                                             int64_t H, int64_t W, int64_t Cout, int accumulate, void* stream, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad masked: bad shape");
    const int pieces = split_pieces_of(algo);
    if (!pieces) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 wgrad masked: a split-bf16 id is needed (SSTEM_CONV_MFMA_BF16X6 / _BF16X3)");
    if (!input || !grad_output || !grad_weight) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 wgrad masked: null tensor pointer");
    if (Cin * Cout >= ((int64_t)1 << 31)) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad masked: Cin*Cout too large");
    if (!sstem::conv3x3_wgrad_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 wgrad masked: outside the split kernel's range");
    if (!workspace || workspace_floats < sstem::conv3x3_wgrad_split_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad masked: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
    const hipError_t e = sstem::launch_conv3x3_wgrad_split_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin,
                                                                (int)H, (int)W, (int)Cout, pieces, static_cast<hipStream_t>(stream),
                                                                wgrad_flags(accumulate), grad_mask);
    if (e != hipSuccess) return hip_fail("conv3x3 wgrad masked launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_backward_weight_scaled_masked_f32(const float* input, const float* input_amax, const float* grad_output,
                                                    const float* grad_amax, const uint8_t* grad_mask, float* grad_weight, float* grad_bias,
                                                    float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H,
                                                    int64_t W, int64_t Cout, int accumulate, void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad scaled: bad shape");
    if (!input || !input_amax || !grad_output || !grad_amax || !grad_weight)
        return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 wgrad scaled: null tensor pointer");
    if (Cin * Cout >= ((int64_t)1 << 31)) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad scaled: Cin*Cout too large");
    if (!sstem::conv3x3_wgrad_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 wgrad scaled: outside the split kernel's range");
    if (!workspace || workspace_floats < sstem::conv3x3_wgrad_split_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad scaled: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
    const hipError_t e = sstem::launch_conv3x3_wgrad_split_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin,
                                                                (int)H, (int)W, (int)Cout, 2, static_cast<hipStream_t>(stream),
                                                                wgrad_flags(accumulate), grad_mask, input_amax, grad_amax);
    if (e != hipSuccess) return hip_fail("conv3x3 wgrad scaled launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_algo_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    if (algo == SSTEM_CONV_DIRECT) return 1;
    if (algo == SSTEM_CONV_MFMA_F16X3) return sstem::conv3x3_split_f16_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout) ? 1 : 0;
    if (algo == SSTEM_CONV_MFMA_BF16 || scaled_pieces_of(algo)) return sstem::conv3x3_bf16_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout) ? 1 : 0;
    if (algo == SSTEM_CONV_MFMA || algo == SSTEM_CONV_AUTO) return N * ((Cout + 31) / 32) < 65536 ? 1 : 0;
    return 0;
}

int sstem_conv3x3_first_layer_u8_supported(int64_t N, int64_t H, int64_t W, int64_t Cout)
{
    if (N < 0 || H < 0 || W < 0 || N > 65535 || H > (1 << 20) || W > (1 << 20)) return 0;
    return sstem::conv3x3_first_u8_supported((int)N, (int)H, (int)W, (int)Cout) ? 1 : 0;
}

int sstem_conv3x3_first_layer_u8(const uint8_t* frames, const float* weight, const float* bias, float* output, float* planes,
                                 float* output_amax, int64_t N, int64_t H, int64_t W, int64_t Cout, int act, float slope, void* stream)
{
    if (!sstem_conv3x3_first_layer_u8_supported(N, H, W, Cout))
        return fail(SSTEM_ERR_UNSUPPORTED, "first layer from uint8 frames: Conv2d(6 -> 6), W % 4 == 0, 2*H*W < 2^31");
    if (!frames || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "first layer from uint8 frames: null pointer");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "first layer from uint8 frames: unknown activation");
    hipError_t e = sstem::launch_conv3x3_first_u8(frames, weight, bias, output, planes, (int)N, (int)H, (int)W, (int)Cout, act, slope,
                                                  output_amax, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("first layer from uint8 frames launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_stream_small_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return sstem::conv3x3_stream_small_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout) ? 1 : 0;
}

int sstem_conv3x3_bf16io_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int output_bf16)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return sstem::conv3x3_bf16_io_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout, output_bf16) ? 1 : 0;
}

int sstem_conv3x3_forward_bf16io(const void* input, int input_bf16, const float* weight, const float* bias, const float* scale,
                                 const float* shift, void* output, int output_bf16, float* workspace, int64_t workspace_floats,
                                 int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                 void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 bf16io: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io: unknown activation id");
    if (weight_flags < 0 || weight_flags > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io: unknown weight flags");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 bf16io: null tensor pointer");
    if (Cin == 0 || !sstem::conv3x3_bf16_io_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout, output_bf16))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io: needs W % 4 == 0, a channel plane below 2 GiB and, for a bf16 output, an unsplit launch "
                                           "(sstem_conv3x3_bf16io_supported)");
    if (!workspace || workspace_floats < sstem::conv3x3_bf16_packed_floats((int)Cin, (int)Cout))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 bf16io: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
    hipError_t e = sstem::launch_conv3x3_bf16_mfma_io(input, input_bf16 ? 1 : 0, weight, bias, scale, shift, output, output_bf16 ? 1 : 0,
                                                      workspace, workspace_floats, (int)N, (int)Cin, (int)H, (int)W, (int)Cout, act, slope,
                                                      weight_flags, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("conv3x3 bf16io launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_forward_bf16io_masked(const void* input, int input_bf16, const uint8_t* input_mask, const float* weight, const float* bias,
                                        const float* scale, const float* shift, void* output, int output_bf16, uint8_t* output_mask,
                                        float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                        int64_t Cout, int weight_flags, int act, float slope, void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 bf16io masked: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io masked: unknown activation id");
    if (weight_flags < 0 || weight_flags > 3) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io masked: unknown weight flags");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 bf16io masked: null tensor pointer");
    if (input_bf16 && input_mask)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io masked: input_mask needs an fp32 input tensor");
    if ((input_mask || output_mask) && (reinterpret_cast<uintptr_t>(input) & 15) != 0)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io masked: the masks need a 16-byte aligned input");
    if (Cin == 0 || !sstem::conv3x3_bf16_io_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout, output_bf16))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 bf16io masked: needs W % 4 == 0, a channel plane below 2 GiB and, for a bf16 output, an "
                                           "unsplit launch (sstem_conv3x3_bf16io_supported)");
    if (!workspace || workspace_floats < sstem::conv3x3_bf16_packed_floats((int)Cin, (int)Cout))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 bf16io masked: workspace too small (see sstem_conv3x3_forward_workspace_floats_algo)");
    hipError_t e = sstem::launch_conv3x3_bf16_mfma_io(input, input_bf16 ? 1 : 0, weight, bias, scale, shift, output, output_bf16 ? 1 : 0,
                                                      workspace, workspace_floats, (int)N, (int)Cin, (int)H, (int)W, (int)Cout, act, slope,
                                                      weight_flags, static_cast<hipStream_t>(stream), input_mask, output_mask);
    if (e != hipSuccess) return hip_fail("conv3x3 bf16io masked launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_backward_weight_bf16_masked(const void* input, int input_bf16, const float* grad_output, const uint8_t* grad_mask,
                                              float* grad_weight, float* grad_bias, float* workspace, int64_t workspace_floats, int64_t N,
                                              int64_t Cin, int64_t H, int64_t W, int64_t Cout, int accumulate, void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad bf16 masked: bad shape");
    if (!input || !grad_output || !grad_weight) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 wgrad bf16 masked: null tensor pointer");
    if (W % 4 != 0 || ((reinterpret_cast<uintptr_t>(input) | reinterpret_cast<uintptr_t>(grad_output)) & 15) != 0)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 wgrad bf16 masked: needs W % 4 == 0 and 16-byte aligned tensors");
    if (Cin * Cout >= ((int64_t)1 << 31)) return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad bf16 masked: Cin*Cout too large");
    const int64_t need = sstem::conv3x3_wgrad_bf16_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if (!workspace || workspace_floats < need)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad bf16 masked: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
    hipError_t e = sstem::launch_conv3x3_wgrad_bf16_mfma_in(input, input_bf16 ? 1 : 0, grad_output, grad_weight, grad_bias, workspace, (int)N,
                                                            (int)Cin, (int)H, (int)W, (int)Cout, static_cast<hipStream_t>(stream),
                                                            wgrad_flags(accumulate), grad_mask);
    if (e != hipSuccess) return hip_fail("conv3x3 wgrad bf16 masked launch", e);
    return SSTEM_OK;
}

int sstem_conv_transpose3x3s2_forward_f32(const float* input, const float* weight, const float* bias,
                                          const float* scale, const float* shift, float* output,
                                          int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                          int act, float slope, void* stream)
{
    if (!conv_sizes_ok(N, Cin, 2 * H, 2 * W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: unknown activation id");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose: null tensor pointer");
    hipError_t e = sstem::launch_convT3x3s2_direct(input, weight, bias, scale, shift, output, (int)N, (int)Cin,
                                                   (int)H, (int)W, (int)Cout, act, slope,
                                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("conv_transpose launch", e);
    return SSTEM_OK;
}

int64_t sstem_conv_transpose3x3s2_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int which)
{
    if (!conv_sizes_ok(N, Cin, 2 * H, 2 * W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    const int64_t f = sstem::convT3x3s2_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    const int64_t d = sstem::convT3x3s2_dgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    const int64_t g = sstem::convT3x3s2_wgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if (which == 0) return f;
    if (which == 1) return d;
    if (which == 2) return g;
    if (which == 3) return d > g ? d : g;
    return 0;
}

int sstem_conv_transpose3x3s2_forward_ex_f32(const float* input, const float* weight, const float* bias,
                                             const float* scale, const float* shift, const float* residual, float residual_scale,
                                             float* output, float* bn_partials, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                             int weight_flags, int act, float slope, void* stream)
{
    if (!conv_sizes_ok(N, Cin, 2 * H, 2 * W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: unknown activation id");
    if (weight_flags != 0 && weight_flags != SSTEM_CONV_WEIGHT_PREPACKED) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: unknown weight flags");
    if (N == 0 || Cout == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !weight || !output) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose: null tensor pointer");
    if (Cin == 0) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: no input channels");
    if (8 * H * W * 4 >= ((int64_t)1 << 32)) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: plane too large for the MFMA kernel's 32-bit offsets");
    const int64_t need = sstem::convT3x3s2_forward_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    const int64_t packed = sstem::conv3x3_workspace_floats((int)Cin, (int)Cout);
    if (!workspace || workspace_floats < (bn_partials ? need : packed))
        return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose: workspace too small (see sstem_conv_transpose3x3s2_workspace_floats)");
    if (bn_partials && (scale || shift || act != 0 || residual))
        return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: bn_partials are the statistics of the raw output (no affine, activation or residual)");
    if (bn_partials && sstem::convT3x3s2_bn_partials((int)N, (int)Cin, (int)H, (int)W, (int)Cout) == 0)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose: this launch is split over K and writes no statistics (sstem_conv_bn_partials == 0)");
    const sstem::ConvExtra ex{residual, residual_scale, bn_partials, 0};
    hipError_t e = sstem::launch_convT3x3s2_mfma(input, weight, bias, scale, shift, output, workspace, workspace_floats, (int)N, (int)Cin,
                                                 (int)H, (int)W, (int)Cout, act, slope, weight_flags ? 1 : 0,
                                                 static_cast<hipStream_t>(stream), ex);
    if (e != hipSuccess) return hip_fail("conv_transpose launch", e);
    return SSTEM_OK;
}

int sstem_conv_transpose3x3s2_backward_ex_f32(const float* input, const float* weight, const float* grad_output,
                                              float* grad_input, float* grad_weight, float* grad_bias,
                                              float* workspace, int64_t workspace_floats,
                                              int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                              int accumulate, void* stream)
{
    if (!conv_sizes_ok(N, Cin, 2 * H, 2 * W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose backward: bad shape");
    if (Cin == 0 || Cout == 0) return SSTEM_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0 || H == 0 || W == 0) {
        if (!accumulate) {
            hipError_t e = grad_weight ? hipMemsetAsync(grad_weight, 0, (size_t)Cout * Cin * 9 * sizeof(float), s) : hipSuccess;
            if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), s);
            if (e != hipSuccess) return hip_fail("conv_transpose backward memset", e);
        }
        return SSTEM_OK;
    }
    if (!grad_output) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null grad_output");
    if (grad_bias && !grad_weight) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose backward: the bias gradient rides along with the weight gradient");
    if (8 * 4 * H * W * 4 >= ((int64_t)1 << 32)) return fail(SSTEM_ERR_UNSUPPORTED, "conv_transpose backward: plane too large for 32-bit offsets");
    if (grad_input) {
        if (!weight) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null weight");
        if (!workspace || workspace_floats < sstem::conv3x3_workspace_floats((int)Cout, (int)Cin))
            return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose backward: workspace too small (see sstem_conv_transpose3x3s2_workspace_floats)");
        hipError_t e = sstem::launch_convT3x3s2_dgrad_mfma(grad_output, weight, grad_input, workspace, workspace_floats, (int)N, (int)Cin,
                                                           (int)H, (int)W, (int)Cout, s);
        if (e != hipSuccess) return hip_fail("conv_transpose dgrad launch", e);
    }
    if (grad_weight) {
        if (!input) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null input");
        if (!workspace || workspace_floats < sstem::convT3x3s2_wgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
            return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose backward: workspace too small (see sstem_conv_transpose3x3s2_workspace_floats)");
        hipError_t e = sstem::launch_convT3x3s2_wgrad_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin, (int)H,
                                                           (int)W, (int)Cout, s, wgrad_flags(accumulate));
        if (e != hipSuccess) return hip_fail("conv_transpose wgrad launch", e);
    }
    return SSTEM_OK;
}

int64_t sstem_conv3x3_wgrad_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N == 0 || Cin == 0 || H == 0 || W == 0 || Cout == 0) return 0;
    return sstem::conv3x3_wgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
}

int64_t sstem_conv3x3_wgrad_workspace_floats_algo(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N == 0 || Cin == 0 || H == 0 || W == 0 || Cout == 0) return 0;
    if (algo == SSTEM_CONV_MFMA_BF16) return sstem::conv3x3_wgrad_bf16_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if ((split_pieces_of(algo) || algo == SSTEM_CONV_MFMA_F16X3) && sstem::conv3x3_wgrad_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout))
        return sstem::conv3x3_wgrad_split_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if (algo == SSTEM_CONV_DIRECT) return 0;
    return sstem::conv3x3_wgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
}

int sstem_conv2d_backward_weight_f32(const float* input, const float* grad_output, float* grad_weight,
                                     float* workspace, int64_t workspace_floats,
                                     int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                     int KH, int KW, int pad_h, int pad_w, void* stream, int algo)
{
    return sstem_conv2d_backward_weight_bias_f32(input, grad_output, grad_weight, nullptr, workspace, workspace_floats,
                                                 N, Cin, H, W, Cout, KH, KW, pad_h, pad_w, stream, algo);
}

int sstem_conv2d_backward_weight_bias_f32(const float* input, const float* grad_output, float* grad_weight,
                                          float* grad_bias, float* workspace, int64_t workspace_floats,
                                          int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                          int KH, int KW, int pad_h, int pad_w, void* stream, int algo)
{
    return sstem_conv2d_backward_weight_bias_ex_f32(input, grad_output, grad_weight, grad_bias, workspace, workspace_floats,
                                                    N, Cin, H, W, Cout, KH, KW, pad_h, pad_w, 0, stream, algo);
}

int sstem_conv2d_backward_weight_bias_ex_f32(const float* input, const float* grad_output, float* grad_weight,
                                             float* grad_bias, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                             int KH, int KW, int pad_h, int pad_w, int accumulate, void* stream, int algo)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || KH <= 0 || KW <= 0 || KH > 5 || KW > 5)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv2d wgrad: bad shape (kernel up to 5x5)");
    if (2 * pad_h != KH - 1 || 2 * pad_w != KW - 1)
        return fail(SSTEM_ERR_UNSUPPORTED, "conv2d wgrad: only stride-1 'same' padding is supported");
    if (Cin == 0 || Cout == 0) return SSTEM_OK;
    if (!grad_weight) return fail(SSTEM_ERR_NULL_POINTER, "conv2d wgrad: null grad_weight");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0 || H == 0 || W == 0) {
        if (accumulate) return SSTEM_OK;
        hipError_t e = hipMemsetAsync(grad_weight, 0, (size_t)Cout * Cin * KH * KW * sizeof(float), s);
        if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), s);
        if (e != hipSuccess) return hip_fail("conv2d wgrad memset", e);
        return SSTEM_OK;
    }
    if (!input || !grad_output) return fail(SSTEM_ERR_NULL_POINTER, "conv2d wgrad: null tensor pointer");
    if (Cin * Cout >= ((int64_t)1 << 31)) return fail(SSTEM_ERR_BAD_SHAPE, "conv2d wgrad: Cin*Cout too large");
    const bool is3x3 = (KH == 3 && KW == 3);
    if (algo == SSTEM_CONV_AUTO) algo = is3x3 ? SSTEM_CONV_MFMA : SSTEM_CONV_DIRECT;
    if (split_pieces_of(algo) && (!is3x3 || !sstem::conv3x3_wgrad_split_supported((int)N, (int)Cin, (int)H, (int)W, (int)Cout)))
        algo = SSTEM_CONV_MFMA;                                 // outside the split kernel's range: the fp32 MFMA kernel
    hipError_t e;
    if (algo == SSTEM_CONV_MFMA) {
        if (!is3x3) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d wgrad: the MFMA kernel is 3x3 only");
        const int64_t need = sstem::conv3x3_wgrad_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
        if (!workspace || workspace_floats < need)
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d wgrad: workspace too small (see sstem_conv3x3_wgrad_workspace_floats)");
        e = sstem::launch_conv3x3_wgrad_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin, (int)H,
                                             (int)W, (int)Cout, s, wgrad_flags(accumulate));
    } else if (split_pieces_of(algo)) {
        const int64_t need = sstem::conv3x3_wgrad_split_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
        if (!workspace || workspace_floats < need)
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d wgrad: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
        e = sstem::launch_conv3x3_wgrad_split_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin, (int)H,
                                                   (int)W, (int)Cout, split_pieces_of(algo), s, wgrad_flags(accumulate));
    } else if (algo == SSTEM_CONV_MFMA_BF16) {
        if (!is3x3) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d wgrad: the bf16 MFMA kernel is 3x3 only");
        const int64_t need = sstem::conv3x3_wgrad_bf16_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
        if (!workspace || workspace_floats < need)
            return fail(SSTEM_ERR_BAD_SHAPE, "conv2d wgrad: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
        e = sstem::launch_conv3x3_wgrad_bf16_mfma(input, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin, (int)H,
                                                  (int)W, (int)Cout, s, wgrad_flags(accumulate));
    } else if (algo == SSTEM_CONV_DIRECT) {
        if (grad_bias) return fail(SSTEM_ERR_UNSUPPORTED, "conv2d wgrad: the fused bias gradient needs the 3x3 MFMA kernel");
        e = sstem::launch_conv2d_wgrad_direct(input, grad_output, grad_weight, (int)N, (int)Cin, (int)H, (int)W,
                                              (int)Cout, KH, KW, pad_h, pad_w, s, accumulate ? 1 : 0);
    } else {
        return fail(SSTEM_ERR_UNSUPPORTED, "conv2d wgrad: unknown algorithm id");
    }
    if (e != hipSuccess) return hip_fail("conv2d wgrad launch", e);
    return SSTEM_OK;
}

int sstem_conv3x3_backward_weight_bf16in(const void* input_bf16, const float* grad_output, float* grad_weight, float* grad_bias,
                                         float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                         int64_t Cout, void* stream)
{
    return sstem_conv3x3_backward_weight_bf16in_ex(input_bf16, grad_output, grad_weight, grad_bias, workspace, workspace_floats,
                                                   N, Cin, H, W, Cout, 0, stream);
}

int sstem_conv3x3_backward_weight_bf16in_ex(const void* input_bf16, const float* grad_output, float* grad_weight, float* grad_bias,
                                            float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                            int64_t Cout, int accumulate, void* stream)
{
    if (!conv_sizes_ok(N, Cin, H, W, Cout) || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad bf16in: bad shape");
    if (!input_bf16 || !grad_output || !grad_weight) return fail(SSTEM_ERR_NULL_POINTER, "conv3x3 wgrad bf16in: null tensor pointer");
    if (W % 4 != 0) return fail(SSTEM_ERR_UNSUPPORTED, "conv3x3 wgrad bf16in: needs W % 4 == 0");
    const int64_t need = sstem::conv3x3_wgrad_bf16_workspace_floats((int)N, (int)Cin, (int)H, (int)W, (int)Cout);
    if (!workspace || workspace_floats < need)
        return fail(SSTEM_ERR_BAD_SHAPE, "conv3x3 wgrad bf16in: workspace too small (see sstem_conv3x3_wgrad_workspace_floats_algo)");
    hipError_t e = sstem::launch_conv3x3_wgrad_bf16_mfma_in(input_bf16, 1, grad_output, grad_weight, grad_bias, workspace, (int)N, (int)Cin,
                                                            (int)H, (int)W, (int)Cout, static_cast<hipStream_t>(stream), wgrad_flags(accumulate));
    if (e != hipSuccess) return hip_fail("conv3x3 wgrad bf16in launch", e);
    return SSTEM_OK;
}

int sstem_conv_transpose3x3s2_backward_f32(const float* input, const float* weight,
                                           const float* grad_output, float* grad_input,
                                           float* grad_weight,
                                           int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                           void* stream)
{
    if (!conv_sizes_ok(N, Cin, 2 * H, 2 * W, Cout)) return fail(SSTEM_ERR_BAD_SHAPE, "conv_transpose backward: bad shape");
    if (Cin == 0 || Cout == 0) return SSTEM_OK;
    if (!grad_output && N > 0 && H > 0 && W > 0) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null grad_output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (grad_input && N > 0 && H > 0 && W > 0) {
        if (!weight) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null weight");
        hipError_t e = sstem::launch_convT3x3s2_dgrad_direct(grad_output, weight, grad_input, (int)N, (int)Cin,
                                                             (int)H, (int)W, (int)Cout, s);
        if (e != hipSuccess) return hip_fail("conv_transpose dgrad launch", e);
    }
    if (grad_weight) {
        if (!input && N > 0 && H > 0 && W > 0) return fail(SSTEM_ERR_NULL_POINTER, "conv_transpose backward: null input");
        hipError_t e = sstem::launch_convT3x3s2_wgrad_direct(input, grad_output, grad_weight, (int)N, (int)Cin,
                                                             (int)H, (int)W, (int)Cout, s);
        if (e != hipSuccess) return hip_fail("conv_transpose wgrad launch", e);
    }
    return SSTEM_OK;
}

// ---- bilinear back-warp (include/sstem_warp.h) -------------------------------------------------
int sstem_warp_bilinear_f32(const float* image, const float* flow, float* output,
                            int64_t B, int64_t C, int64_t H, int64_t W, void* stream)
{
    if (!conv_sizes_ok(B, C, H, W, 2)) return fail(SSTEM_ERR_BAD_SHAPE, "warp: bad shape");
    if (B == 0 || C == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!image || !flow || !output) return fail(SSTEM_ERR_NULL_POINTER, "warp: null tensor pointer");
    hipError_t e = sstem::launch_warp_bilinear(image, flow, output, (int)B, (int)C, (int)H, (int)W,
                                               static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("warp launch", e);
    return SSTEM_OK;
}

// ---- uint8 edge + flat Adam (include/sstem_io.h) -----------------------------------------------
int sstem_gray_u8_to_f32(const uint8_t* image, float* output, int64_t npix, int64_t replicas, void* stream)
{
    if (npix < 0 || replicas < 0 || replicas > 1024 || npix > ((int64_t)1 << 40)) return fail(SSTEM_ERR_BAD_SHAPE, "u8->f32: bad size");
    if (npix == 0 || replicas == 0) return SSTEM_OK;
    if (!image || !output) return fail(SSTEM_ERR_NULL_POINTER, "u8->f32: null pointer");
    hipError_t e = sstem::launch_gray_u8_to_f32(image, output, npix, (int)replicas, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("u8->f32 launch", e);
    return SSTEM_OK;
}

static bool bn_sizes_ok(int64_t N, int64_t C, int64_t HW)
{
    return N >= 0 && C >= 0 && HW >= 0 && N <= (1 << 20) && C <= 65535 && HW <= ((int64_t)1 << 31) &&
           (__int128)N * C * HW < ((__int128)1 << 40);
}

int64_t sstem_batchnorm_workspace_floats(int64_t N, int64_t C, int64_t HW)
{
    if (!bn_sizes_ok(N, C, HW)) return 0;
    return sstem::bn_workspace_floats(N, C, HW);
}

int sstem_batchnorm_train_forward_f32(const float* x, const float* weight, const float* bias,
                                      float* running_mean, float* running_var, float* y,
                                      float* save_mean, float* save_invstd,
                                      float* workspace, int64_t workspace_floats,
                                      int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                      int act, float slope, void* stream)
{
    return sstem_batchnorm_train_forward_ex_f32(x, weight, bias, running_mean, running_var, nullptr, y, save_mean, save_invstd,
                                                nullptr, 0, workspace, workspace_floats, N, C, HW, momentum, eps, act, slope, stream);
}

int sstem_batchnorm_train_forward_ex_f32(const float* x, const float* weight, const float* bias,
                                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                         float* save_mean, float* save_invstd,
                                         const float* partials, int64_t n_partials,
                                         float* workspace, int64_t workspace_floats,
                                         int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                         int act, float slope, void* stream)
{
    return sstem_batchnorm_train_forward_amax_f32(x, weight, bias, running_mean, running_var, num_batches_tracked, y, nullptr, save_mean,
                                                  save_invstd, partials, n_partials, workspace, workspace_floats, N, C, HW, momentum, eps,
                                                  act, slope, stream);
}

int sstem_batchnorm_train_forward_amax_f32(const float* x, const float* weight, const float* bias,
                                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                           float* y_amax, float* save_mean, float* save_invstd,
                                           const float* partials, int64_t n_partials,
                                           float* workspace, int64_t workspace_floats,
                                           int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                           int act, float slope, void* stream)
{
    if (!bn_sizes_ok(N, C, HW)) return fail(SSTEM_ERR_BAD_SHAPE, "batchnorm: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "batchnorm: unknown activation id");
    if (N == 0 || C == 0 || HW == 0) return SSTEM_OK;
    if (!x || !y || !save_mean || !save_invstd) return fail(SSTEM_ERR_NULL_POINTER, "batchnorm: null tensor pointer");
    if (partials && n_partials <= 0) return fail(SSTEM_ERR_BAD_SHAPE, "batchnorm: partials without a count");
    if (!partials && (!workspace || workspace_floats < sstem::bn_workspace_floats(N, C, HW)))
        return fail(SSTEM_ERR_BAD_SHAPE, "batchnorm: workspace too small (see sstem_batchnorm_workspace_floats)");
    hipError_t e = sstem::launch_bn_train_forward(x, weight, bias, running_mean, running_var, y, save_mean, save_invstd,
                                                  workspace, (int)N, (int)C, HW, momentum, eps, act, slope,
                                                  static_cast<hipStream_t>(stream), partials, n_partials,
                                                  reinterpret_cast<long long*>(num_batches_tracked), y_amax);
    if (e != hipSuccess) return hip_fail("batchnorm forward launch", e);
    return SSTEM_OK;
}

int sstem_batchnorm_train_backward_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                       const float* save_mean, const float* save_invstd,
                                       float* dx, float* dweight, float* dbias,
                                       float* workspace, int64_t workspace_floats,
                                       int64_t N, int64_t C, int64_t HW, int act, float slope, void* stream)
{
    return sstem_batchnorm_train_backward_ex_f32(dy, x, weight, bias, save_mean, save_invstd, dx, dweight, dbias, workspace,
                                                 workspace_floats, N, C, HW, act, slope, 0, stream);
}

int sstem_batchnorm_train_backward_ex_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                          const float* save_mean, const float* save_invstd,
                                          float* dx, float* dweight, float* dbias,
                                          float* workspace, int64_t workspace_floats,
                                          int64_t N, int64_t C, int64_t HW, int act, float slope, int accumulate, void* stream)
{
    return sstem_batchnorm_train_backward_amax_f32(dy, x, weight, bias, save_mean, save_invstd, dx, nullptr, dweight, dbias, workspace,
                                                   workspace_floats, N, C, HW, act, slope, accumulate, stream);
}

int sstem_batchnorm_train_backward_amax_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                            const float* save_mean, const float* save_invstd,
                                            float* dx, float* dx_amax, float* dweight, float* dbias,
                                            float* workspace, int64_t workspace_floats,
                                            int64_t N, int64_t C, int64_t HW, int act, float slope, int accumulate, void* stream)
{
    if (!bn_sizes_ok(N, C, HW)) return fail(SSTEM_ERR_BAD_SHAPE, "batchnorm: bad shape");
    if (act < 0 || act > 2) return fail(SSTEM_ERR_UNSUPPORTED, "batchnorm: unknown activation id");
    if (N == 0 || C == 0 || HW == 0) return SSTEM_OK;
    if (!dy || !x || !dx || !save_mean || !save_invstd) return fail(SSTEM_ERR_NULL_POINTER, "batchnorm: null tensor pointer");
    if (!workspace || workspace_floats < sstem::bn_workspace_floats(N, C, HW))
        return fail(SSTEM_ERR_BAD_SHAPE, "batchnorm: workspace too small (see sstem_batchnorm_workspace_floats)");
    hipError_t e = sstem::launch_bn_train_backward(dy, x, weight, bias, save_mean, save_invstd, dx, dweight, dbias, workspace,
                                                   (int)N, (int)C, HW, act, slope, static_cast<hipStream_t>(stream), accumulate ? 1 : 0, dx_amax);
    if (e != hipSuccess) return hip_fail("batchnorm backward launch", e);
    return SSTEM_OK;
}

int sstem_upsample_bilinear2x_f32(const float* input, float* output, int64_t planes, int64_t H, int64_t W, void* stream)
{
    if (planes < 0 || H < 0 || W < 0 || H > (1 << 14) || W > (1 << 14) || planes > ((int64_t)1 << 30))
        return fail(SSTEM_ERR_BAD_SHAPE, "upsample: bad shape");
    if (W % 2 != 0) return fail(SSTEM_ERR_UNSUPPORTED, "upsample: the input width must be even (16-byte output stores)");
    if (planes == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !output) return fail(SSTEM_ERR_NULL_POINTER, "upsample: null pointer");
    if ((reinterpret_cast<uintptr_t>(output) & 15) != 0) return fail(SSTEM_ERR_UNSUPPORTED, "upsample: output must be 16-byte aligned");
    hipError_t e = sstem::launch_upsample_bilinear2x(input, output, planes, (int)H, (int)W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("upsample launch", e);
    return SSTEM_OK;
}

int sstem_upsample_bilinear2x_backward_f32(const float* grad_output, float* grad_input, int64_t planes, int64_t H, int64_t W, void* stream)
{
    if (planes < 0 || H < 0 || W < 0 || H > (1 << 14) || W > (1 << 14) || planes > ((int64_t)1 << 31) - 1)
        return fail(SSTEM_ERR_BAD_SHAPE, "upsample backward: bad shape");
    if (planes == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!grad_output || !grad_input) return fail(SSTEM_ERR_NULL_POINTER, "upsample backward: null pointer");
    hipError_t e = sstem::launch_upsample_bilinear2x_backward(grad_output, grad_input, planes, (int)H, (int)W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("upsample backward launch", e);
    return SSTEM_OK;
}

int sstem_pool2x2_forward_f32(const float* input, float* output, uint8_t* argmax, int64_t planes, int64_t H, int64_t W, int is_max, void* stream)
{
    if (planes < 0 || H < 0 || W < 0 || H > (1 << 15) || W > (1 << 15) || planes > ((int64_t)1 << 31) - 1)
        return fail(SSTEM_ERR_BAD_SHAPE, "pool2x2: bad shape");
    if (planes == 0 || H < 2 || W < 2) return SSTEM_OK;
    if (!input || !output) return fail(SSTEM_ERR_NULL_POINTER, "pool2x2: null pointer");
    hipError_t e = sstem::launch_pool2x2_forward(input, output, is_max ? argmax : nullptr, planes, (int)H, (int)W, is_max ? 1 : 0,
                                                 static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("pool2x2 launch", e);
    return SSTEM_OK;
}

int sstem_pool2x2_backward_f32(const float* grad_output, const uint8_t* argmax, float* grad_input, int64_t planes, int64_t H, int64_t W,
                               int is_max, void* stream)
{
    if (planes < 0 || H < 0 || W < 0 || H > (1 << 15) || W > (1 << 15) || planes > ((int64_t)1 << 31) - 1)
        return fail(SSTEM_ERR_BAD_SHAPE, "pool2x2 backward: bad shape");
    if (planes == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!grad_input || ((H >= 2 && W >= 2) && !grad_output)) return fail(SSTEM_ERR_NULL_POINTER, "pool2x2 backward: null pointer");
    if (is_max && H >= 2 && W >= 2 && !argmax) return fail(SSTEM_ERR_NULL_POINTER, "pool2x2 backward: the maximum needs its argmax bytes");
    hipError_t e = sstem::launch_pool2x2_backward(grad_output, argmax, grad_input, planes, (int)H, (int)W, is_max ? 1 : 0,
                                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("pool2x2 backward launch", e);
    return SSTEM_OK;
}

int sstem_f32_to_gray_u8(const float* pred, uint8_t* output, int64_t npix, int clamp01, void* stream)
{
    if (npix < 0 || npix > ((int64_t)1 << 40)) return fail(SSTEM_ERR_BAD_SHAPE, "f32->u8: bad size");
    if (npix == 0) return SSTEM_OK;
    if (!pred || !output) return fail(SSTEM_ERR_NULL_POINTER, "f32->u8: null pointer");
    hipError_t e = sstem::launch_f32_to_gray_u8(pred, output, npix, clamp01 ? 1 : 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("f32->u8 launch", e);
    return SSTEM_OK;
}

int sstem_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                        void* stream)
{
    if (n < 0 || n > ((int64_t)1 << 40) || step < 1) return fail(SSTEM_ERR_BAD_SHAPE, "adam: bad size or step < 1");
    if (n == 0) return SSTEM_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(SSTEM_ERR_NULL_POINTER, "adam: null pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipError_t e = sstem::launch_adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                                           (float)bc1, (float)sqrt(bc2), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("adam launch", e);
    return SSTEM_OK;
}

// ---- any filter length (the reference's cupy spelling: sff_scripts_interp/model/sepconv.py:8-31, SIZE_1(vertical)) -----------------
int sstem_sepconv_forward_taps_f32(const float* input, const float* vertical, const float* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W, int taps, void* stream)
{
    if (taps == SSTEM_SEPCONV_FILTER) return sstem_sepconv_forward_f32(input, vertical, horizontal, output, B, C, H, W, stream);
    if (taps < 1 || taps > 1024 || B < 0 || C < 0 || H < 0 || W < 0 ||
        (__int128)B * (C > taps ? C : taps) * (H + taps) * (W + taps) >= ((__int128)1 << 46))
        return fail(SSTEM_ERR_BAD_SHAPE, "forward (any filter length): bad shape or filter length");
    if (B == 0 || C == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !vertical || !horizontal || !output) return fail(SSTEM_ERR_NULL_POINTER, "forward (any filter length): null tensor pointer");
    hipError_t e = sstem::launch_fwd_direct(input, vertical, horizontal, output, B, C, H, W, taps, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("sepconv forward (any filter length) launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_backward_taps_f32(const float* grad_output, const float* input, const float* vertical, const float* horizontal,
                                    float* grad_vertical, float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W, int taps, void* stream)
{
    if (taps < 1 || taps > 1024 || B < 0 || C < 0 || H < 0 || W < 0 ||
        (__int128)B * (C > taps ? C : taps) * (H + taps) * (W + taps) >= ((__int128)1 << 46))
        return fail(SSTEM_ERR_BAD_SHAPE, "backward (any filter length): bad shape or filter length");
    if (B == 0 || C == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!grad_output || !input || !vertical || !horizontal || !grad_vertical || !grad_horizontal)
        return fail(SSTEM_ERR_NULL_POINTER, "backward (any filter length): null tensor pointer");
    hipError_t e = sstem::launch_bwd_direct(grad_output, input, vertical, horizontal, grad_vertical, grad_horizontal, B, C, H, W, taps,
                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("sepconv backward (any filter length) launch", e);
    return SSTEM_OK;
}

// ---- bf16 coefficient tensors (BASELINE config 5: "bf16 activations with fp32 sepconv accumulate") -------------------
int64_t sstem_sepconv_forward_bytes_bf16coef(int64_t B, int64_t C, int64_t H, int64_t W)
{
    return 4 * (B * C * (H + 50) * (W + 50) + B * C * H * W) + 2 * (2 * B * 51 * H * W);
}

int64_t sstem_sepconv_backward_bytes_bf16coef(int64_t B, int64_t C, int64_t H, int64_t W)
{
    return 4 * (B * C * H * W + B * C * (H + 50) * (W + 50) + 2 * B * 51 * H * W) + 2 * (2 * B * 51 * H * W);
}

int64_t sstem_sepconv_interp_apply_bytes_bf16coef(int64_t B, int64_t H, int64_t W, int frame_planes)
{
    return 4 * (2 * B * frame_planes * H * W + B * H * W) + 2 * (4 * B * 51 * H * W);
}

int sstem_sepconv_forward_bf16coef(const float* input, const uint16_t* vertical, const uint16_t* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, C, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "forward (bf16 coefficients): negative or oversized shape");
    if (B == 0 || C == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!input || !vertical || !horizontal || !output) return fail(SSTEM_ERR_NULL_POINTER, "forward (bf16 coefficients): null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "forward (bf16 coefficients): grid too large");
    hipError_t e = sstem::launch_fwd_bf16coef(input, vertical, horizontal, output, B, C, H, W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("sepconv forward (bf16 coefficients) launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_backward_bf16coef(const float* grad_output, const float* input, const uint16_t* vertical, const uint16_t* horizontal,
                                    float* grad_input, float* grad_vertical, float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W, void* stream)
{
    (void)grad_input;      // never written, as in the reference (kernel.cu:152-206) and in sstem_sepconv_backward_f32
    if (!sizes_ok(B, C, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "backward (bf16 coefficients): negative or oversized shape");
    if (C != 3) return fail(SSTEM_ERR_BAD_SHAPE, "backward: the gradient kernels are defined for 3 channels (kernel.cu:100-108)");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!grad_output || !input || !vertical || !horizontal || !grad_vertical || !grad_horizontal)
        return fail(SSTEM_ERR_NULL_POINTER, "backward (bf16 coefficients): null tensor pointer");
    if (!sstem::mfma_grid_ok(B, H, W)) return fail(SSTEM_ERR_UNSUPPORTED, "backward (bf16 coefficients): grid too large");
    hipError_t e = sstem::launch_bwd_bf16coef(grad_output, input, vertical, horizontal, grad_vertical, grad_horizontal, B, C, H, W,
                                              static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("sepconv backward (bf16 coefficients) launch", e);
    return SSTEM_OK;
}

int sstem_sepconv_interp_apply_gray_bf16coef_supported(int64_t B, int64_t H, int64_t W)
{
    return (sizes_ok(B, 3, H, W) && B > 0 && H > 0 && W > 0 && sstem::mfma_grid_ok(B, H, W) && sstem::interp_fused_gray_bf16coef_ok(H, W)) ? 1 : 0;
}

int sstem_sepconv_interp_apply_gray_bf16coef(const float* g1, const float* g2, const uint16_t* k1v, const uint16_t* k1h,
                                             const uint16_t* k2v, const uint16_t* k2h, float* output,
                                             int64_t B, int64_t H, int64_t W, void* stream)
{
    if (!sizes_ok(B, 3, H, W)) return fail(SSTEM_ERR_BAD_SHAPE, "gray interp apply (bf16 coefficients): negative or oversized shape");
    if (B == 0 || H == 0 || W == 0) return SSTEM_OK;
    if (!g1 || !g2 || !k1v || !k1h || !k2v || !k2h || !output)
        return fail(SSTEM_ERR_NULL_POINTER, "gray interp apply (bf16 coefficients): null tensor pointer");
    if (!sstem_sepconv_interp_apply_gray_bf16coef_supported(B, H, W))
        return fail(SSTEM_ERR_UNSUPPORTED, "gray interp apply (bf16 coefficients): grid too large or 51*H*W*4 bytes per image not below 4 GiB");
    hipError_t e = sstem::launch_interp_fused_gray_bf16coef(g1, g2, k1v, k1h, k2v, k2h, output, B, H, W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("gray interp apply (bf16 coefficients) launch", e);
    return SSTEM_OK;
}

int64_t sstem_l1_workspace_floats(void) { return 1024 + 1; }

int sstem_l1_mean_forward_grad_f32(const float* pred, const float* target, int64_t n, float* loss, float* grad, float* workspace,
                                   void* stream)
{
    if (n < 1 || n > ((int64_t)1 << 40)) return fail(SSTEM_ERR_BAD_SHAPE, "l1: bad size");
    if (!pred || !target || !loss || !grad || !workspace) return fail(SSTEM_ERR_NULL_POINTER, "l1: null pointer");
    hipError_t e = sstem::launch_l1_mean_fwd_grad(pred, target, n, loss, grad, workspace, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("l1 launch", e);
    return SSTEM_OK;
}

}  // extern "C"
