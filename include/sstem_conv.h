/*
 * sstem_conv.h -- C-ABI of the dense convolution blocks of libsstem_hip.so (MI355X / gfx950).
 *
 * The reference has no native code for these: every layer is a torch.nn module executed by
 * cuDNN/ATen.  The entry points below replace, for the U-Net / kernel-prediction blocks on the
 * hot path, the module sequences
 *
 *   nn.Conv2d(k=3,s=1,p=1) [+ nn.BatchNorm2d in eval mode] [+ nn.ReLU | nn.LeakyReLU(0.2)]
 *       sff_scripts_interp/model/model_interp.py:121-143  (IFNet _conv/_kernel/_upsample modules)
 *       sp_scripts_train/networks.py:179-186              (DoubleConv)
 *       sff_scripts_fusion/model/model_unet.py:11-48      (contracting/expansive/final blocks)
 *       sff_scripts_fusion/model/model_fusionnet.py:12-43 (conv_block, conv_block_3)
 *   nn.Conv2d(k=1)                                        sp_scripts_train/networks.py:238 (OutConv)
 *   nn.ConvTranspose2d(k=3,s=2,p=1,output_padding=1) [+BN eval][+act]
 *       model_unet.py:32,70; model_fusionnet.py:21-27
 *
 * with one fused launch:   out = act( (conv(in, w) + bias) * scale + shift )
 * where scale/shift are the folded BatchNorm affine (gamma/sqrt(var+eps), beta - mean*that), NULL = 1/0.
 * Tensors: contiguous NCHW fp32 device pointers.  Same status codes / error reporting / stream and
 * ownership rules as sstem_sepconv.h.  Arithmetic: exact fp32 (fmaf chains on the matrix cores).
 */
#ifndef SSTEM_CONV_H
#define SSTEM_CONV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSTEM_ACT_NONE 0
#define SSTEM_ACT_RELU 1
#define SSTEM_ACT_LEAKY_RELU 2

#define SSTEM_CONV_WEIGHT_TRANSPOSED 1
#define SSTEM_CONV_WEIGHT_PREPACKED 2

#define SSTEM_CONV_AUTO 0
#define SSTEM_CONV_DIRECT 1   /* one lane per output element, any kernel size */
#define SSTEM_CONV_MFMA 2     /* 3x3/s1/p1 implicit GEMM on fp32 MFMA */
#define SSTEM_CONV_MFMA_BF16 3 /* opt-in, never chosen by AUTO: the same 3x3 GEMM with both operands rounded to bf16 (RNE) as they are
                                * staged, exact products, fp32 sums (v_mfma_f32_32x32x16_bf16).  Tensors stay fp32 in memory.
                                * BASELINE config 5 ("bf16 activations") -- results differ from the fp32 ids by ~2^-9 relative
                                * per operand; parity with the reference's fp32 path is NOT claimed for this id. */
#define SSTEM_CONV_MFMA_BF16X3 4 /* opt-in: fp32 operands split into TWO bf16 pieces each (x = h + m, exact subtraction), three exact
                                  * products hh + hm + mh summed in fp32 on the bf16 matrix cores: per product the dropped terms are
                                  * <= 3 * 2^-18 = 1.1e-5 relative (about 200 x finer than SSTEM_CONV_MFMA_BF16, 3/16 of the fp32
                                  * MFMA's pipe time).  Forward, data gradient and weight gradient (the bias gradient is summed from the fp32 values). */
#define SSTEM_CONV_MFMA_BF16X6 5 /* explicit id (SSTEM_CONV_AUTO of this C-ABI never resolves to it; the shipped Python binding's own AUTO
                                  * uses it for the layers where it is the faster kernel): THREE bf16 pieces per operand (x = h + m + l exactly), the six products of order <= 2^-16:
                                  * every product is x * y to 2^-26 relative, below half an fp32 ulp -- the arithmetic of the fp32 ids
                                  * (exact products, fp32 sums in another order) at 6/16 of the fp32 MFMA's pipe time.  Same tests and
                                  * tolerances as SSTEM_CONV_MFMA.  Forward, data gradient and weight gradient. */

#define SSTEM_CONV_MFMA_F16X3 6 /* inference launches through sstem_conv3x3_forward_scaled_f32 only (never chosen by this C-ABI's AUTO; the shipped
                                 * Python binding's AUTO uses it when no backward can follow): fp32 operands as TWO fp16 pieces each, scaled by a
                                 * per-tensor power of two taken from an upper bound of the tensor's largest magnitude (fp16 has 5 exponent
                                 * bits): x*s = h0 + h1 to 2^-22, three exact products h0g0 + h0g1 + h1g0 per term summed in fp32 on
                                 * v_mfma_f32_32x32x16_f16 -- x*y to 2^-22 relative per product (45x finer than _BF16X3) at half the matrix
                                 * instructions of _BF16X6.  Scales are exact and leave the sums by an exponent shift.  22 of fp32's 24 bits: a
                                 * one-hot weight does NOT copy its input bit for bit; values more than 18 binades below the tensor's bound fade
                                 * out (absolute error 2^-25 of the bound).  Same tests and tolerance as SSTEM_CONV_MFMA.  Range
                                 * (sstem_conv3x3_algo_supported): that of the bf16 ids, and with W % 4 == 0 either eight channel planes or
                                 * the whole image below 2^31 bytes. */

/* Scratch floats the 3x3 MFMA path needs for its packed weights (caller-allocated, device): the minimum. */
int64_t sstem_conv3x3_workspace_floats(int64_t Cin, int64_t Cout);

/* Scratch floats to pass for full speed: packed weights + the partial-sum slices of the split-K form the kernel
 * uses on small grids (deep layers at small batch).  With less than this (but at least the minimum above) the
 * forward runs unsplit.  Results are deterministic for a given workspace size. */
int64_t sstem_conv3x3_forward_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout);
/* The same query for an explicit algorithm id (SSTEM_CONV_MFMA_BF16 packs its weights differently); 0 for ids without workspace. */
int64_t sstem_conv3x3_forward_workspace_floats_algo(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo);

/* Packed weights on their own (training: ONE launch per layer and step writes the packing the forward needs and the transposed +
 * flipped one its data gradient needs; both are then passed with SSTEM_CONV_WEIGHT_PREPACKED).  sstem_conv3x3_packed_floats(Cin,
 * Cout, algo) = floats of one packing for a convolution with Cin inputs and Cout outputs under an explicit MFMA id (0 otherwise).
 * weight [Cout,Cin,3,3]; packed_forward: >= packed_floats(Cin, Cout) floats, the head of the forward workspace; packed_transposed:
 * >= packed_floats(Cout, Cin) floats, the head of the workspace of the (Cout -> Cin) data-gradient call; either may be NULL. */
int64_t sstem_conv3x3_packed_floats(int64_t Cin, int64_t Cout, int algo);
int sstem_conv3x3_pack_weights_f32(const float* weight, int64_t Cin, int64_t Cout, int algo, float* packed_forward,
                                   float* packed_transposed, void* stream);

/* The same for MANY layers in one launch (after an optimiser step every layer's weights have changed).  `table` is a DEVICE array of
 * n_entries x 16 int64: [0] weight pointer, [1] packed_forward pointer, [2] packed_transposed pointer, [3..12] the layout numbers that
 * sstem_conv3x3_pack_group_entry(Cin, Cout, algo, entry16) writes for that layer (it returns the 256-thread blocks the entry needs),
 * [13] the first block of the entry = the running sum of those returns over the preceding entries, [14..15] zero.  total_blocks = the
 * sum over all entries.  hipnn.PackGroup builds the table once and FlatAdam.step launches it after its update. */
int64_t sstem_conv3x3_pack_group_entry(int64_t Cin, int64_t Cout, int algo, int64_t* entry16);
int sstem_conv3x3_pack_weights_group_f32(const int64_t* table, int64_t n_entries, int64_t total_blocks, int algo, void* stream);
/* The fp16 two-piece id (SSTEM_CONV_MFMA_F16X3) packs under each layer's own bound: sstem_conv3x3_pack_weights_f32 measures it (three
 * launches per layer); the group form is ONE clear + ONE bound launch + ONE pack launch for all layers: table entries as above from
 * sstem_conv3x3_pack_group_entry(.., SSTEM_CONV_MFMA_F16X3, entry16) (its return counts one thread per weight slot: both pieces of a
 * weight are made by one thread), which also leaves in entry16[14] the entry's blocks of the bound
 * launch; the caller replaces [14] by the running sum of those counts over the preceding entries and sets [15] = the address of
 * bounds[entry index]; bound_blocks = their total; `bounds`: n_entries device floats (cleared and written here). */
int sstem_conv3x3_pack_weights_group_f16(const int64_t* table, int64_t n_entries, int64_t total_blocks, int64_t bound_blocks, float* bounds,
                                         void* stream);

/* Conv2d, stride 1, "same" zero padding pad_h/pad_w, weight [Cout,Cin,KH,KW].
 * weight_transposed != 0: weight is [Cin,Cout,3,3] and is applied transposed with flipped taps
 * (the data-gradient of a 3x3 convolution: grad_in = conv(grad_out, W^T flipped)); 3x3 only.
 * weight_transposed bit 1 (SSTEM_CONV_WEIGHT_PREPACKED, MFMA ids only): the head of `workspace` still holds the packed weights
 * an earlier call with the same weight values, orientation, algorithm id and sizes wrote there -- the call skips its packing
 * launch (frozen / inference weights: one launch per layer less; hipnn keeps such workspaces on the module).
 * workspace may be NULL when the direct algorithm is forced. */
int sstem_conv2d_forward_f32(const float* input, const float* weight, const float* bias,
                             const float* scale, const float* shift, float* output,
                             float* workspace, int64_t workspace_floats,
                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                             int KH, int KW, int pad_h, int pad_w, int weight_transposed,
                             int act, float slope, void* stream, int algo);

/* The same launch with the two things the reference's blocks do right after a convolution folded into its store:
 *   residual (nullable, [N,Cout,H,W]):  output = (act(affine(conv + bias)) + residual) * residual_scale
 *       -- `conv_1 + conv_2` of Conv_residual_conv (model_fusionnet.py:57-61; residual_scale 1) and the skip average
 *       `(deconv + down) / 2` (model_fusionnet.py:129-138; residual_scale 0.5): one elementwise launch and one pass less each;
 *   bn_partials (nullable): the convolution feeds a train-mode nn.BatchNorm2d (networks.py:179-186, model_unet.py:11-48): every
 *       workgroup also writes (count, mean, M2) of its tile's raw conv + bias values per output channel, M2 = sum of squared
 *       deviations from that tile's own mean (two passes over registers: no cancellation), to
 *       bn_partials[(co * P + tile) * 3 ..], P = sstem_conv_bn_partials(...).  sstem_batchnorm_train_forward_ex_f32 takes them
 *       instead of making its own statistics pass over the tensor.  Needs scale = shift = residual = NULL, act = NONE, the
 *       fp32 3x3 MFMA id and the full workspace.
 * sstem_conv_bn_partials: partial triplets per channel that launch writes (0: no statistics from this configuration, pass NULL);
 * transposed != 0 asks for sstem_conv_transpose3x3s2_forward_ex_f32 (H, W = its INPUT size). */
int64_t sstem_conv_bn_partials(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int KH, int KW, int transposed, int algo);
int sstem_conv2d_forward_ex_f32(const float* input, const float* weight, const float* bias,
                                const float* scale, const float* shift, const float* residual, float residual_scale,
                                float* output, float* bn_partials, float* workspace, int64_t workspace_floats,
                                int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                int KH, int KW, int pad_h, int pad_w, int weight_transposed,
                                int act, float slope, void* stream, int algo);

/* Split-bf16 ids only (SSTEM_CONV_MFMA_BF16X6 / _BF16X3): the 3x3 launch of sstem_conv2d_forward_f32 with the ReLU bookkeeping of a
 * training step folded into it (the reference's nn.Conv2d + nn.ReLU pairs, model_interp.py:121-143: autograd keeps the activation's
 * output and multiplies the incoming gradient by (output > 0) in a pass of its own).
 *   output_mask (nullable, [N,Cout,H,W] bytes): receives 1 where the stored activation output is > 0, else 0 -- written by the forward
 *       launch instead of a compare pass over the output;
 *   input_mask (nullable, [N,Cin,H,W] bytes): an input element counts as 0 where its byte is 0 -- the data-gradient launch
 *       (weight_flags = SSTEM_CONV_WEIGHT_TRANSPOSED [| _PREPACKED], input = the gradient wrt the activation's output) applies the mask
 *       while it stages the gradient instead of a select pass; sstem_conv3x3_backward_weight_masked_f32 does the same for the weight
 *       and bias gradient (grad_mask, [N,Cout,H,W] bytes).
 * Workspaces as for the unmasked entries under the same id. */
int sstem_conv3x3_forward_masked_f32(const float* input, const uint8_t* input_mask, const float* weight, const float* bias,
                                     const float* scale, const float* shift, float* output, uint8_t* output_mask,
                                     float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                     int64_t Cout, int weight_flags, int act, float slope, void* stream, int algo);
int sstem_conv3x3_backward_weight_masked_f32(const float* input, const float* grad_output, const uint8_t* grad_mask, float* grad_weight,
                                             float* grad_bias, float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin,
                                             int64_t H, int64_t W, int64_t Cout, int accumulate, void* stream, int algo);

/* Amax words and the scaled forward.  An amax word is sstem_amax_word_floats() (= 1024) floats in device memory (16-byte aligned), zeroed by
 * whoever allocates it; its meaning is "no element of the tensor is larger in magnitude than the largest of these 1024" (many slots: a
 * launch adds one atomic per workgroup, and atomics on one 64-byte line execute one after the other).  Producers only ever
 * raise slots (atomic max): sstem_amax_f32 makes one pass over a tensor; the split kernels add the largest value they STORE when given
 * output_amax (after bias, folded BatchNorm, activation and residual), so a chain of layers needs no extra pass.  A bound that is too
 * large costs precision only beyond 18 binades; a bound that is too small overflows fp16 -- never reuse a word for another tensor.
 * sstem_conv3x3_forward_scaled_f32 = the 3x3 launch of sstem_conv2d_forward_ex_f32 (same weight flags, workspace rules -- sized by
 * sstem_conv3x3_forward_workspace_floats_algo --, residual store) for the ids SSTEM_CONV_MFMA_F16X3 (input_amax required), _BF16X6,
 * _BF16X3 (input_amax ignored), with the optional output bound and the output layout (SSTEM_LAYOUT_*).  The reference's blocks it serves: every Conv3x3 [+ BatchNorm eval]
 * [+ ReLU | LeakyReLU] of model_interp.py:121-143, networks.py:179-186, model_unet.py:11-48, model_fusionnet.py:12-43 at inference. */
#define SSTEM_LAYOUT_NCHW 0
#define SSTEM_LAYOUT_ROW_SEGMENTS 1   /* [N][H][ceil(W/64)][Cout][64]: the blocked coefficient layout of sstem_sepconv.h -- what the last convolution
                                       * of an IFNet kernel head (model_interp.py:129-137, Conv 51 -> 51) stores for the fused apply to read.
                                       * Same values; columns beyond W inside the last segment are not written.  No residual with it; the launch
                                       * is never split over K. */
#define SSTEM_LAYOUT_CONVT_PARITY 2   /* The launch is the SUB-PIXEL FORM of nn.ConvTranspose2d(k3, s2, p1, output_padding 1) (model_fusionnet.py:21-27,
                                       * model_unet.py:32,70) on the fp16 two-piece id: output pixel (2y + py, 2x + px) of the transposed convolution
                                       * receives 1, 2, 2 or 4 taps, all from in[y .. y+1][x .. x+1] -- a 2 x 2 convolution with Cout = 4 C channels
                                       * in parity-major order (co' = (2 py + px) C + co).  `weight` is [4 C, Cin, 3, 3] with the window in taps
                                       * (ky, kx) in {1, 2}^2:  W'[(py,px) co][ci][1 + dy][1 + dx] = wT[ci][co][kyT(py, dy)][kxT(px, dx)],
                                       * kyT(0,0) = 1, kyT(1,0) = 2, kyT(1,1) = 0, no tap for (0,1) (zero); taps with ky = 0 or kx = 0 are never
                                       * read.  bias / scale / shift: 4 C values (the real channel's, four times).  `output` (and `residual`) are
                                       * [N, C, 2H, 2W]; output_amax as for the other layouts.  Needs SSTEM_CONV_MFMA_F16X3, C % 32 == 0,
                                       * Cin % 16 == 0, W % 4 == 0; never split over K.  9 of the 16 issued taps are real: a third of the fp32
                                       * MFMA kernel's matrix-pipe time (sstem_conv_transpose3x3s2_forward_ex_f32 stays the exact-fp32 form). */
int64_t sstem_amax_word_floats(void);
int sstem_amax_f32(const float* x, int64_t n, float* word, void* stream);
int sstem_conv3x3_forward_scaled_f32(const float* input, const float* input_amax, const float* weight, const float* bias,
                                     const float* scale, const float* shift, const float* residual, float residual_scale,
                                     float* output, float* output_amax, float* workspace, int64_t workspace_floats,
                                     int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                     void* stream, int algo, int output_layout);
/* The same launch with two more things the networks do right behind a convolution folded into its store (round 4):
 *  - storing into a channel block of a LARGER tensor: output_image_stride = floats between the images of `output` (0 or Cout*H*W: back
 *    to back), e.g. the [N, 2C, H, W] tensor a U-Net decoder concatenates (model_unet.py:86, `torch.cat((up, skip), 1)`) with `output`
 *    pointing at channel 0 or C of image 0 -- the producers store straight into it and the concatenation disappears.
 *    SSTEM_LAYOUT_NCHW and SSTEM_LAYOUT_CONVT_PARITY only;
 *  - pooled_output (nullable) = [N, Cout, H/2, W/2], contiguous: receives the 2 x 2 pooling of the stored values as well (pool_kind
 *    SSTEM_POOL_MAX: nn.MaxPool2d(2), model_fusionnet.py / model_unet.py; SSTEM_POOL_AVG: nn.AvgPool2d(2), model_interp.py:60-70) with
 *    the arithmetic of sstem_pool2x2_forward_f32, bit for bit -- the pooling launch and its read of the full-resolution tensor disappear.
 *    SSTEM_CONV_MFMA_F16X3, SSTEM_LAYOUT_NCHW, no residual, H % 8 == 0, W % 32 == 0.  With a pooled_output, `output` may be NULL: the
 *    pooled copy alone is stored (the IFNet's first block, model_interp.py:60-61: nothing but the pooling reads its result).
 * Such a launch is never split over K.
 * algo = SSTEM_CONV_DIRECT (round 4): the streaming fp32 kernel for layers with 1, 2, 3, 4, 6 or 8 OUTPUT channels and W % 4 == 0
 * (sstem_conv3x3_stream_small_supported) -- the last layers of the SFF nets (32 -> 2 flow, 32 -> 1 restored section at full resolution,
 * model_fusionnet.py / model_unet.py) and the IFNet's first block (6 -> 6, model_interp.py:121-127), which occupy 3-20 % of the matrix
 * kernels' 32-channel output block: exact fp32 products summed in (ci, ky, kx) order, plain [Cout, Cin, 3, 3] weights (weight_flags 0),
 * no workspace, input_amax ignored (may be NULL), plain NCHW store without residual / stride / pooled copy; output_amax is filled as
 * under the split ids, so the layer can sit inside an fp16 chain. */
int sstem_conv3x3_stream_small_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout);
#define SSTEM_POOL_NONE 0
#define SSTEM_POOL_MAX 1
#define SSTEM_POOL_AVG 2
int sstem_conv3x3_forward_scaled_strided_f32(const float* input, const float* input_amax, const float* weight, const float* bias,
                                             const float* scale, const float* shift, const float* residual, float residual_scale,
                                             float* output, float* output_amax, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                             void* stream, int algo, int output_layout, int64_t output_image_stride, float* pooled_output,
                                             int pool_kind);

/* The same bookkeeping for the bf16-operand id (BASELINE config 5): sstem_conv3x3_forward_bf16io / sstem_conv3x3_backward_weight_bf16in_ex
 * with the masks of sstem_conv3x3_forward_masked_f32.  input_mask needs an fp32 input tensor (input_bf16 = 0; the incoming gradient
 * always is); both need W % 4 == 0 and 16-byte aligned tensors; output_mask describes the value that is stored (after the bf16 rounding of a
 * bf16 output). */
int sstem_conv3x3_forward_bf16io_masked(const void* input, int input_bf16, const uint8_t* input_mask, const float* weight, const float* bias,
                                        const float* scale, const float* shift, void* output, int output_bf16, uint8_t* output_mask,
                                        float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                        int64_t Cout, int weight_flags, int act, float slope, void* stream);
int sstem_conv3x3_backward_weight_bf16_masked(const void* input, int input_bf16, const float* grad_output, const uint8_t* grad_mask,
                                              float* grad_weight, float* grad_bias, float* workspace, int64_t workspace_floats, int64_t N,
                                              int64_t Cin, int64_t H, int64_t W, int64_t Cout, int accumulate, void* stream);

/* Can the 3x3 forward / data-gradient launch of this size run under `algo`?  (SSTEM_CONV_MFMA_BF16 needs W % 4 == 0 or an image
 * below 2 GiB; the MFMA ids a grid the launch can index.)  hipnn asks before every layer and falls back to SSTEM_CONV_MFMA for the
 * layers a forced bf16 id cannot take, instead of failing the whole model. */
int sstem_conv3x3_algo_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo);

/* The bf16-operand 3x3 convolution (SSTEM_CONV_MFMA_BF16) with bf16 ACTIVATION TENSORS on either side: input_bf16 / output_bf16 != 0
 * mean the tensor is bf16 NCHW instead of fp32.  For the convolutions inside one block (the reference's Conv-ReLU-Conv-ReLU-Conv
 * nn.Sequential, model_interp.py:121-127) when no backward can follow: numerically free -- the consumer rounds the same fp32 value
 * to bf16 with the same instruction -- and half the traffic.  Same weight / workspace / flag conventions as sstem_conv2d_forward_f32;
 * sstem_conv3x3_bf16io_supported: W % 4 == 0 and, for a bf16 output, a launch that is not split over K. */
int sstem_conv3x3_bf16io_supported(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int output_bf16);
int sstem_conv3x3_forward_bf16io(const void* input, int input_bf16, const float* weight, const float* bias, const float* scale,
                                 const float* shift, void* output, int output_bf16, float* workspace, int64_t workspace_floats,
                                 int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope,
                                 void* stream);

/* ConvTranspose2d(k=3, s=2, p=1, output_padding=1), weight [Cin,Cout,3,3], output [N,Cout,2H,2W]. */
int sstem_conv_transpose3x3s2_forward_f32(const float* input, const float* weight, const float* bias,
                                          const float* scale, const float* shift, float* output,
                                          int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                          int act, float slope, void* stream);

/* ConvTranspose2d(k=3, s=2, p=1, output_padding=1) on the fp32 matrix cores by output-parity decomposition (csrc/convt_kernels.hip):
 * each of the four output parities is a 1-, 2-, 2- or 4-tap convolution of the INPUT-resolution tensor -- 9 multiply-adds per
 * 2x2 output block and channel pair, no zero-inserted tensor (the earlier route ran a 3x3 convolution over a 4x-sized tensor of
 * 3/4 zeros).  Same epilogue (bias, folded BatchNorm affine, activation, residual, bn_partials) as sstem_conv2d_forward_ex_f32.
 * weight [Cin,Cout,3,3]; output / residual [N,Cout,2H,2W]; weight_flags: 0 or SSTEM_CONV_WEIGHT_PREPACKED.
 * sstem_conv_transpose3x3s2_workspace_floats(.., which): 0 forward, 1 data gradient, 2 weight gradient, 3 = max(1, 2) for the
 * backward entry (its two launches use the workspace one after the other). */
int64_t sstem_conv_transpose3x3s2_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int which);
int sstem_conv_transpose3x3s2_forward_ex_f32(const float* input, const float* weight, const float* bias,
                                             const float* scale, const float* shift, const float* residual, float residual_scale,
                                             float* output, float* bn_partials, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                             int weight_flags, int act, float slope, void* stream);
/* Its gradients on the matrix cores: grad_input [N,Cin,H,W] = a stride-2 3x3 convolution of grad_output (every computed value is
 * used), grad_weight [Cin,Cout,3,3] and grad_bias [Cout] (nullable; rides along with grad_weight) summed in a fixed order.
 * accumulate != 0: grad_weight / grad_bias are ADDED to (they are the parameters' .grad buffers). */
int sstem_conv_transpose3x3s2_backward_ex_f32(const float* input, const float* weight, const float* grad_output,
                                              float* grad_input, float* grad_weight, float* grad_bias,
                                              float* workspace, int64_t workspace_floats,
                                              int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                              int accumulate, void* stream);

/* Weight gradient of the stride-1 "same" Conv2d above (KH,KW <= 5):
 *   grad_weight[co,ci,ky,kx] = sum_{n,y,x} grad_output[n,co,y,x] * input[n,ci,y+ky-pad_h,x+kx-pad_w]
 * (the bias gradient is a plain reduction of grad_output and is left to the caller).  The data
 * gradient is sstem_conv2d_forward_f32(grad_output, weight, ..., weight_transposed = 1).
 * algo: SSTEM_CONV_AUTO / _DIRECT / _MFMA (3x3 only; needs the workspace below; sums in a fixed order) /
 * _MFMA_BF16 (opt-in, 3x3 only: input and grad_output rounded to bf16 while staged, fp32 sums in a fixed order; the bias
 * gradient is summed from the fp32 values). */
int sstem_conv2d_backward_weight_f32(const float* input, const float* grad_output, float* grad_weight,
                                     float* workspace, int64_t workspace_floats,
                                     int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                     int KH, int KW, int pad_h, int pad_w, void* stream, int algo);

/* The same with the bias gradient grad_bias[Cout] = sum over batch and pixels of grad_output (nullable) computed in the
 * same launches: the kernel has the grad_output tiles in LDS anyway, so one reduction launch and one full re-read of
 * grad_output per layer disappear.  3x3 MFMA algorithm only (SSTEM_ERR_UNSUPPORTED otherwise when grad_bias != NULL);
 * fixed summation order. */
int sstem_conv2d_backward_weight_bias_f32(const float* input, const float* grad_output, float* grad_weight,
                                          float* grad_bias, float* workspace, int64_t workspace_floats,
                                          int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                          int KH, int KW, int pad_h, int pad_w, void* stream, int algo);

/* The same with accumulate != 0: the results are ADDED to grad_weight / grad_bias, which then are the parameters' .grad buffers
 * (hipnn hands over the views of its flat gradient bucket): autograd's one add launch per parameter and step disappears.
 * One read-modify-write per element in stream order: bitwise reproducible. */
int sstem_conv2d_backward_weight_bias_ex_f32(const float* input, const float* grad_output, float* grad_weight,
                                             float* grad_bias, float* workspace, int64_t workspace_floats,
                                             int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                             int KH, int KW, int pad_h, int pad_w, int accumulate, void* stream, int algo);

/* Scratch floats of the 3x3 MFMA weight-gradient path (split-K partial slabs + bias partial sums; device memory). */
int64_t sstem_conv3x3_wgrad_workspace_floats(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout);
/* The same query for an explicit algorithm id (SSTEM_CONV_MFMA_BF16 splits the pixel tiles differently). */
int64_t sstem_conv3x3_wgrad_workspace_floats_algo(int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int algo);

/* Weight (+ bias) gradient of a 3x3 convolution under SSTEM_CONV_MFMA_BF16 whose saved input is a bf16 NCHW tensor (a block run as
 * one autograd function keeps the tensors between its convolutions in bf16: hipnn's conv chain).  Same results as the fp32-tensor
 * entry on the fp32 values those bf16 numbers were rounded from.  Workspace: sstem_conv3x3_wgrad_workspace_floats_algo(.., _MFMA_BF16). */
int sstem_conv3x3_backward_weight_bf16in(const void* input_bf16, const float* grad_output, float* grad_weight, float* grad_bias,
                                         float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                         int64_t Cout, void* stream);

int sstem_conv3x3_backward_weight_bf16in_ex(const void* input_bf16, const float* grad_output, float* grad_weight, float* grad_bias,
                                            float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H, int64_t W,
                                            int64_t Cout, int accumulate, void* stream);

/* Gradients of the ConvTranspose2d(k=3,s=2,p=1,op=1) above; input [N,Cin,H,W], grad_output
 * [N,Cout,2H,2W], weight / grad_weight [Cin,Cout,3,3].  Either output pointer may be NULL. */
int sstem_conv_transpose3x3s2_backward_f32(const float* input, const float* weight,
                                           const float* grad_output, float* grad_input,
                                           float* grad_weight,
                                           int64_t N, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                           void* stream);

/* Recorded (training) launches on the fp16 two-piece id (round 5): sstem_conv3x3_forward_scaled_strided_f32's arithmetic
 * (SSTEM_CONV_MFMA_F16X3: x s = h0 + h1, three exact products per term, fp32 sums; input_amax = the input's amax word) with the ReLU
 * bookkeeping of sstem_conv3x3_forward_masked_f32: input_mask (nullable, [N,Cin,H,W] bytes) zeroes input elements whose byte is 0 while
 * they are staged -- the data gradient of a layer behind a ReLU (weight_flags bit 0: transposed + flipped weights) --, output_mask
 * (nullable, [N,Cout,H,W] bytes) receives (stored value > 0).  W % 4 == 0, 16-byte aligned input; plain NCHW store.  output_amax
 * (nullable) receives the largest stored magnitude.  Workspace: sstem_conv3x3_forward_workspace_floats_algo(.., SSTEM_CONV_MFMA_F16X3). */
int sstem_conv3x3_forward_scaled_masked_f32(const float* input, const float* input_amax, const uint8_t* input_mask, const float* weight,
                                            const float* bias, const float* scale, const float* shift, float* output, float* output_amax,
                                            uint8_t* output_mask, float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin,
                                            int64_t H, int64_t W, int64_t Cout, int weight_flags, int act, float slope, void* stream);

/* Weight + bias gradient on the fp16 two-piece id: sstem_conv3x3_backward_weight_masked_f32's sums (same slabs, same fixed-order
 * reduce, same accumulate flags, grad_mask nullable) with input and grad_output split into two fp16 pieces under the scales of their
 * amax words (both required); three exact products per term, 2^-22 per product.  The bias gradient is an fp32 sum of grad_output.
 * Workspace: sstem_conv3x3_wgrad_workspace_floats_algo(.., SSTEM_CONV_MFMA_F16X3) (= _BF16X6's: the split kernels share one plan). */
int sstem_conv3x3_backward_weight_scaled_masked_f32(const float* input, const float* input_amax, const float* grad_output,
                                                    const float* grad_amax, const uint8_t* grad_mask, float* grad_weight, float* grad_bias,
                                                    float* workspace, int64_t workspace_floats, int64_t N, int64_t Cin, int64_t H,
                                                    int64_t W, int64_t Cout, int accumulate, void* stream);

/* Grouped weight-gradient reduce (round 5).  The weight-gradient entry points (sstem_conv2d_backward_weight_bias_ex_f32,
 * sstem_conv3x3_backward_weight_masked_f32, sstem_conv3x3_backward_weight_bf16in_ex / _bf16_masked, sstem_conv_transpose3x3s2_backward_ex_f32)
 * write partial slabs and then add them in a fixed order -- one small launch per layer.  accumulate == 3 (add into the gradient buffers
 * AND defer) makes them record that second launch instead of issuing it; sstem_wgrad_deferred_flush(stream) then runs every recorded
 * job as ONE launch (the per-layer kernels' bodies on their workgroup shapes: the same sums in the same order, bit for bit) on the
 * stream the slab launches ran on (or one that waits for them).  The caller keeps the workspaces alive until the flush;
 * sstem_wgrad_deferred_drop() forgets recorded jobs (a backward pass that raised).  One list per process. */
int sstem_wgrad_deferred_count(void);
void sstem_wgrad_deferred_drop(void);
int sstem_wgrad_deferred_flush(void* stream);

/* The IFNet's first convolution straight from the uint8 frames (round 5).  The reference reads two 8-bit grayscale
 * PNGs, divides by 255 in float32, replicates each plane x3 and concatenates ([1,6,H,W]: sff_scripts_interp/inference_singleImage.py:55-66);
 * the IFNet's first layer is Conv2d(6 -> 6, 3x3, padding 1) + ReLU (model_interp.py:121-127).  Here:
 *     frames [N,2,H,W] uint8 (frame 1, frame 2);  weight [6,6,3,3], bias [6] (nullable);  output [N,6,H,W] fp32 = act(conv(x) + bias)
 *     with x[n, c] = float32(frames[n, c / 3]) / float32(255)  -- never materialised;
 *     planes [2,N,H,W] fp32 (nullable; FRAME-major: each frame's planes are one contiguous [N,1,H,W] tensor) = the normalised planes,
 *     left behind for the fused apply at the network's other end;
 *     output_amax (nullable): an amax word (sstem_amax_word_floats) that receives the largest magnitude stored.
 * The products are summed in the order of the streaming fp32 kernel for few output channels (ci, ky, kx ascending, fp32 fma): the bits
 * of sstem_conv3x3_forward_scaled_strided_f32(SSTEM_CONV_DIRECT) on the materialised input.  Cout must be 6, W % 4 == 0. */
int sstem_conv3x3_first_layer_u8_supported(int64_t N, int64_t H, int64_t W, int64_t Cout);
int sstem_conv3x3_first_layer_u8(const uint8_t* frames, const float* weight, const float* bias, float* output, float* planes,
                                 float* output_amax, int64_t N, int64_t H, int64_t W, int64_t Cout, int act, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_CONV_H */
