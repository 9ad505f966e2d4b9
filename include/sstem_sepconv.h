/*
 * sstem_sepconv.h -- C-ABI of the MI355X (gfx950) separable-convolution library
 * (libsstem_hip.so).  Plain pointers and sizes only; no torch / THC types.
 *
 * This is the drop-in boundary for the hot path of sydeng99/ssTEM-restoration.
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference repo root):
 *
 *   sstem_sepconv_forward_f32   <- int SeparableConvolution_cuda_forward(
 *                                      THCudaTensor* input, vertical, horizontal, output)
 *                                  libs/sepconv/src/SeparableConvolution_cuda.h:6-11,
 *                                  libs/sepconv/src/SeparableConvolution_cuda.c:13-28,
 *                                  launcher kernel.cu:54-73, kernel kernel.cu:25-52
 *   sstem_sepconv_backward_f32  <- int SeparableConvolution_cuda_backward(
 *                                      THCudaTensor* gradLoss, input, vertical, horizontal,
 *                                      gradInput, gradVertical, gradHorizontal)
 *                                  ..._cuda.h:13-21, ..._cuda.c:31-52,
 *                                  launcher kernel.cu:152-206, kernels :77-112, :115-150
 *
 * Differences from the reference ABI, all forced by the platform:
 *   - THCudaTensor* (PyTorch 0.4 TH structs, gone in torch >= 1.0) become raw
 *     device pointers + the four output sizes; tensors must be contiguous NCHW
 *     fp32 (the reference asserts contiguity, SeparableConvolution.py:33-35).
 *   - the global `extern THCState* state` / THCState_getCurrentStream
 *     (_cuda.c:11, kernel.cu:64) becomes an explicit hipStream_t argument.
 *   - the reference always returns 1 and reports launch errors through
 *     THCudaCheck -> THError.  Here: 0 = success, non-zero = sstem_status code;
 *     sstem_status_string() names it and sstem_last_error() adds detail.
 *
 * Ownership (same as the reference, SeparableConvolution.py:37,60-62): the
 * caller allocates every output; the library allocates and frees nothing, is
 * asynchronous on `stream`, keeps no global mutable state besides the
 * thread-local last-error string, and is safe to call from several host threads.
 *
 * Shapes (K = 51 filter taps):
 *   input       [B, C, H+50, W+50]   replication-padded by the caller
 *   vertical    [B, 51, H, W]
 *   horizontal  [B, 51, H, W]
 *   output      [B, C, H, W]
 *   out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] * V[b,fy,y,x] * H[b,fx,y,x]
 */
#ifndef SSTEM_SEPCONV_H
#define SSTEM_SEPCONV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSTEM_SEPCONV_FILTER 51

typedef enum sstem_status {
    SSTEM_OK = 0,
    SSTEM_ERR_NULL_POINTER = 1,
    SSTEM_ERR_BAD_SHAPE = 2,      /* negative size, or sizes that overflow int64 math */
    SSTEM_ERR_UNSUPPORTED = 3,    /* e.g. backward with C > 3, unknown algorithm id */
    SSTEM_ERR_HIP = 4,            /* a HIP runtime call failed; see sstem_last_error() */
    SSTEM_ERR_NO_DEVICE = 5       /* no gfx950 device / code object not loadable */
} sstem_status;

/* Kernel selection for the *_algo entry points (A/B measurement and tests).
 * AUTO is what the plain entry points use. */
typedef enum sstem_sepconv_algo {
    SSTEM_SEPCONV_AUTO = 0,
    SSTEM_SEPCONV_DIRECT = 1,     /* one lane per output pixel, fp32 VALU; any C */
    SSTEM_SEPCONV_MFMA = 2        /* banded 4x4x1 fp32-MFMA formulation, LDS-tiled */
} sstem_sepconv_algo;

/* Forward.  Replaces SeparableConvolution_cuda_forward (_cuda.h:6-11).
 * `stream` is a hipStream_t (NULL = the null stream).  Empty tensors
 * (B, C, H or W == 0) are a successful no-op. */
int sstem_sepconv_forward_f32(const float* input, const float* vertical,
                              const float* horizontal, float* output,
                              int64_t B, int64_t C, int64_t H, int64_t W,
                              void* stream);

int sstem_sepconv_forward_f32_algo(const float* input, const float* vertical,
                                   const float* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W,
                                   void* stream, int algo);

/* Backward.  Replaces SeparableConvolution_cuda_backward (_cuda.h:13-21).
 *   grad_vertical[b,fy,y,x]   = sum_fx sum_c g[b,c,y,x] in[b,c,y+fy,x+fx] H[b,fx,y,x]
 *   grad_horizontal[b,fx,y,x] = sum_fy sum_c g[b,c,y,x] in[b,c,y+fy,x+fx] V[b,fy,y,x]
 * `grad_input` is accepted for signature parity and is NEVER written, exactly
 * like the reference (kernel.cu:152-206); it may be NULL.
 * The reference hard-codes three channels (kernel.cu:100-108,138-146): C == 3
 * reproduces it; C < 3 (out-of-bounds reads in the reference) sums the C
 * channels that exist; C > 3 (silently truncated by the reference) is refused
 * with SSTEM_ERR_UNSUPPORTED. */
int sstem_sepconv_backward_f32(const float* grad_output, const float* input,
                               const float* vertical, const float* horizontal,
                               float* grad_input, float* grad_vertical,
                               float* grad_horizontal,
                               int64_t B, int64_t C, int64_t H, int64_t W,
                               void* stream);

int sstem_sepconv_backward_f32_algo(const float* grad_output, const float* input,
                                    const float* vertical, const float* horizontal,
                                    float* grad_input, float* grad_vertical,
                                    float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W,
                                    void* stream, int algo);

/* Fused interpolation apply.  Replaces, in ONE launch, the epilogue of IFNet.forward
 * (sff_scripts_interp/model/model_interp.py:90-97; sp_scripts_train/networks.py:116-124 per output channel):
 *     padded_i2 = ReplicationPad2d(25)(i2);  padded_i1 = ReplicationPad2d(25)(i1)
 *     y   = SeparableConvolution(padded_i2, k2v, k2h) + SeparableConvolution(padded_i1, k1v, k1h)
 *     out = torch.mean(y, dim=1, keepdim=True)
 * i1, i2 [B,3,H,W] UNPADDED; k* [B,51,H,W]; out [B,1,H,W].  Forward only (inference); training keeps the
 * separate op so autograd sees it. */
int sstem_sepconv_interp_apply_f32(const float* i1, const float* i2,
                                   const float* k1v, const float* k1h,
                                   const float* k2v, const float* k2h, float* output,
                                   int64_t B, int64_t H, int64_t W, void* stream);

/* The same apply for callers that KNOW their frames are grayscale -- every caller of the reference builds the IFNet
 * input by replicating one plane three times (sff_scripts_interp/inference_singleImage.py:55-61,
 * sp_scripts_test/test_fusion.py:105-106, sp_scripts_train/main_fusion.py:210-211):
 *     g1, g2 [B,1,H,W] UNPADDED planes;  i_k = g_k repeated x3 is implied, never materialised
 *     out[B,1,H,W] = exactly (bit for bit) what sstem_sepconv_interp_apply_f32 returns on the replicated frames.
 * One launch of the identical-channel kernel: no channel comparison, no device-side dispatch, 2/3 less frame traffic.
 * 51*H*W*4 bytes must stay below 4 GiB (SSTEM_ERR_UNSUPPORTED otherwise; ..._supported() answers beforehand). */
int sstem_sepconv_interp_apply_gray_f32(const float* g1, const float* g2,
                                        const float* k1v, const float* k1h,
                                        const float* k2v, const float* k2h, float* output,
                                        int64_t B, int64_t H, int64_t W, void* stream);
int sstem_sepconv_interp_apply_gray_supported(int64_t B, int64_t H, int64_t W);

/* Blocked coefficients.  The apply reads, per 64-pixel row segment, the 51 taps of each of its four coefficient tensors; in
 * NCHW ([B,51,H,W], the reference's layout: model_interp.py:86-89 -> SeparableConvolution.py:29-35) those are 256-byte pieces of 51
 * planes a whole plane apart.  The ROW-SEGMENT layout
 *     blocked[b][y][tx][f][l] = coef[b][f][y][tx*64 + l],   tx < ceil(W/64), l < 64   (elements with tx*64 + l >= W: padding)
 * puts them in 51 consecutive 256-byte runs (13 KB per row segment).  The kernel heads of the IFNet can store it directly
 * (sstem_conv.h: sstem_conv3x3_forward_scaled_f32 with SSTEM_LAYOUT_ROW_SEGMENTS); sstem_sepconv_coef_to_blocked_f32 converts an NCHW tensor (tests, foreign
 * producers).  sstem_sepconv_interp_apply_gray_blocked_f32 is sstem_sepconv_interp_apply_gray_f32 on four tensors of that layout:
 * the same values through the same instruction sequence, i.e. the same bits.  H * ceil(W/64) * 51 * 256 bytes must stay below
 * 4 GiB (..._supported() answers beforehand).  The operator API (SeparableConvolution.apply) stays NCHW. */
int64_t sstem_sepconv_coef_blocked_floats(int64_t B, int64_t H, int64_t W);
int sstem_sepconv_coef_to_blocked_f32(const float* coef, float* blocked, int64_t B, int64_t H, int64_t W, void* stream);
int sstem_sepconv_interp_apply_gray_blocked_f32(const float* g1, const float* g2,
                                                const float* k1v, const float* k1h,
                                                const float* k2v, const float* k2h, float* output,
                                                int64_t B, int64_t H, int64_t W, void* stream);
int sstem_sepconv_interp_apply_gray_blocked_supported(int64_t B, int64_t H, int64_t W);

/* The same apply with the uint8 image stored by the launch that holds the values (round 5): besides output [B,1,H,W]
 * fp32, output_u8 [B,H,W] = (output * 255).astype(np.uint8) as numpy computes it on x86-64 -- fp32 multiply, truncation toward zero, the
 * low 8 bits, NO clamp (sff_scripts_interp/inference_singleImage.py:76, sp_scripts_test/utils/gray2tensor.py:14-20).
 * blocked_coefficients: 0 = NCHW coefficient tensors, 1 = the row-segment layout.  Same bits in `output` as the _f32 entries. */
int sstem_sepconv_interp_apply_gray_u8_f32(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                           const float* k2v, const float* k2h, float* output, uint8_t* output_u8,
                                           int64_t B, int64_t H, int64_t W, int blocked_coefficients, void* stream);

/* Algorithmic HBM bytes of one fused apply: 4*(2*B*frame_planes*H*W + 4*B*51*H*W + B*H*W); frame_planes = 3 for
 * sstem_sepconv_interp_apply_f32, 1 for the gray entry point. */
int64_t sstem_sepconv_interp_apply_bytes(int64_t B, int64_t H, int64_t W, int frame_planes);

/* Algorithmic HBM bytes of one call:
 *   forward : 4*(B*C*(H+50)*(W+50) + 2*B*51*H*W + B*C*H*W)
 *   backward: 4*(B*C*H*W + B*C*(H+50)*(W+50) + 4*B*51*H*W) */
int64_t sstem_sepconv_forward_bytes(int64_t B, int64_t C, int64_t H, int64_t W);
int64_t sstem_sepconv_backward_bytes(int64_t B, int64_t C, int64_t H, int64_t W);

/* Any filter length (round 5).  The reference's alternate, cupy-string spelling of the forward (sff_scripts_interp/model/sepconv.py:8-31)
 * takes the filter length from its tensors (SIZE_1(vertical), :85-90) where the compiled op fixes 51 (kernel.cu:9):
 *     input [B,C,H+taps-1,W+taps-1], vertical / horizontal [B,taps,H,W] -> output [B,C,H,W]
 * taps == 51 is sstem_sepconv_forward_f32; any other length runs on the one-lane-per-element kernels (same loop order, fy outer / fx
 * inner, fp32 fma).  The gradient entry is this library's addition (the reference's backward raises, sepconv.py:140-144): any C, the
 * reference's formulas with C instead of 3 channels. */
int sstem_sepconv_forward_taps_f32(const float* input, const float* vertical, const float* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W, int taps, void* stream);
int sstem_sepconv_backward_taps_f32(const float* grad_output, const float* input, const float* vertical, const float* horizontal,
                                    float* grad_vertical, float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W, int taps, void* stream);

/* bf16 COEFFICIENT tensors (round 5).  BASELINE config 5 runs the SFF interpolation training (sff_scripts_interp/main_ms.py:187-211)
 * with "bf16 activations with fp32 sepconv accumulate"; SURVEY.md 8(b) asks for "+ bf16-coefficient variants" of the two entry points and
 * 8(d) prices them: the two 51*H*W terms of the byte model halved.  The kernel heads' outputs -- vertical / horizontal, [B,51,H,W] -- are
 * handed over as bf16 (the upper 16 bits of an fp32, round-to-nearest-even done by the producer); frames, gradients and every sum stay
 * fp32: a bf16 value is widened exactly, so these entry points compute bit for bit what the _f32 entry points compute on fp32 tensors
 * holding the same (rounded) values (tests/test_sepconv_bf16coef_gpu.py pins exactly that, and the oracle on the rounded values).
 * grad_vertical / grad_horizontal come back in fp32 (the data gradient of the head's last convolution consumes them).
 * x3-replicated grayscale frames -- what every caller of the reference feeds -- run on the streaming kernels (device-side dispatch as
 * in the _f32 entries); other inputs on the one-lane-per-element kernels (correct for any C, slow).
 *   forward  bytes: 4*(B*C*(H+50)*(W+50) + B*C*H*W) + 2*(2*B*51*H*W)
 *   backward bytes: 4*(B*C*H*W + B*C*(H+50)*(W+50) + 2*B*51*H*W) + 2*(2*B*51*H*W)
 *   fused apply   : 4*(2*B*frame_planes*H*W + B*H*W) + 2*(4*B*51*H*W) */
int sstem_sepconv_forward_bf16coef(const float* input, const uint16_t* vertical, const uint16_t* horizontal, float* output,
                                   int64_t B, int64_t C, int64_t H, int64_t W, void* stream);
int sstem_sepconv_backward_bf16coef(const float* grad_output, const float* input, const uint16_t* vertical, const uint16_t* horizontal,
                                    float* grad_input, float* grad_vertical, float* grad_horizontal,
                                    int64_t B, int64_t C, int64_t H, int64_t W, void* stream);
int sstem_sepconv_interp_apply_gray_bf16coef(const float* g1, const float* g2, const uint16_t* k1v, const uint16_t* k1h,
                                             const uint16_t* k2v, const uint16_t* k2h, float* output,
                                             int64_t B, int64_t H, int64_t W, void* stream);
int sstem_sepconv_interp_apply_gray_bf16coef_supported(int64_t B, int64_t H, int64_t W);
int64_t sstem_sepconv_forward_bytes_bf16coef(int64_t B, int64_t C, int64_t H, int64_t W);
int64_t sstem_sepconv_backward_bytes_bf16coef(int64_t B, int64_t C, int64_t H, int64_t W);
int64_t sstem_sepconv_interp_apply_bytes_bf16coef(int64_t B, int64_t H, int64_t W, int frame_planes);

/* Library / error reporting. */
int sstem_version(void);                       /* MAJOR*10000 + MINOR*100 + PATCH */
const char* sstem_status_string(int status);   /* static string, never NULL */
const char* sstem_last_error(void);            /* thread-local detail of the last failure */

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_SEPCONV_H */
