/*
 * sstem_norm.h -- C-ABI of the train-mode BatchNorm2d (+ ReLU / LeakyReLU) of libsstem_hip.so (MI355X / gfx950).
 *
 * Replaces torch's kernels behind the train-mode nn.BatchNorm2d (+ activation) runs inside the reference's blocks
 * (sp_scripts_train/networks.py:179-186 DoubleConv; sff_scripts_fusion/model/model_unet.py:11-48;
 * sff_scripts_fusion/model/model_fusionnet.py:12-43 conv_block / conv_trans_block) with two streaming passes forward
 * (batch statistics as per-chunk (count, mean, M2) triplets merged with Chan's formula -- no E[x^2] - E[x]^2 cancellation;
 * normalise + affine + activation) and two backward.  PyTorch semantics: biased batch variance for the
 * normalisation, running_var updated with the unbiased one, running = (1 - momentum) * running + momentum * batch.
 *
 *   x, y, dy, dx           [N, C, HW]  fp32 contiguous (NCHW with HW = H*W)
 *   weight, bias           [C] or NULL (no affine)
 *   running_mean / _var    [C] or NULL (not tracked); updated in place by the forward
 *   save_mean, save_invstd [C], written by the forward, read by the backward
 *   act                    0 none, 1 ReLU, 2 LeakyReLU(slope) -- applied after the affine; the backward recomputes the
 *                          activation mask from x, so nothing else is kept from the forward
 *   dweight, dbias         [C] or NULL
 *   workspace              sstem_batchnorm_workspace_floats(N, C, HW) floats, device, caller-allocated
 * Fixed summation order (bitwise reproducible).  Same status codes / error reporting / stream and ownership rules as
 * sstem_sepconv.h.
 */
#ifndef SSTEM_NORM_H
#define SSTEM_NORM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t sstem_batchnorm_workspace_floats(int64_t N, int64_t C, int64_t HW);

int sstem_batchnorm_train_forward_f32(const float* x, const float* weight, const float* bias,
                                      float* running_mean, float* running_var, float* y,
                                      float* save_mean, float* save_invstd,
                                      float* workspace, int64_t workspace_floats,
                                      int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                      int act, float slope, void* stream);

int sstem_batchnorm_train_backward_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                       const float* save_mean, const float* save_invstd,
                                       float* dx, float* dweight, float* dbias,
                                       float* workspace, int64_t workspace_floats,
                                       int64_t N, int64_t C, int64_t HW, int act, float slope, void* stream);

/* Forward with the statistics pass optional and torch's bookkeeping inside:
 *   partials (nullable) [C][n_partials][3] = (count, mean, M2) triplets written by the producing convolution
 *       (sstem_conv2d_forward_ex_f32 / sstem_conv_transpose3x3s2_forward_ex_f32, bn_partials): the statistics pass over x is
 *       skipped and the forward is ONE streaming pass; NULL: the statistics launch runs first (workspace needed).  Either way the
 *       triplets of a channel are merged with Chan's formula in double, in a fixed order.
 *   num_batches_tracked (nullable, one int64 on the device): incremented by the launch (nn.BatchNorm2d's counter). */
int sstem_batchnorm_train_forward_ex_f32(const float* x, const float* weight, const float* bias,
                                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                         float* save_mean, float* save_invstd,
                                         const float* partials, int64_t n_partials,
                                         float* workspace, int64_t workspace_floats,
                                         int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                         int act, float slope, void* stream);

/* Backward with accumulate != 0: dweight / dbias are ADDED to (the parameters' .grad buffers). */
int sstem_batchnorm_train_backward_ex_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                          const float* save_mean, const float* save_invstd,
                                          float* dx, float* dweight, float* dbias,
                                          float* workspace, int64_t workspace_floats,
                                          int64_t N, int64_t C, int64_t HW, int act, float slope, int accumulate, void* stream);

/* The two _ex entries with the bound of what they store left behind (round 5: the recorded fp16 convolution launches of a training
 * step scale their operands by it, sstem_conv.h "amax word"): y_amax / dx_amax (nullable) = an amax word -- 1024 device floats, zeroed
 * by the caller before its first use; every workgroup raises one slot to the largest |y| (|dx|) it stored.  Everything else as _ex. */
int sstem_batchnorm_train_forward_amax_f32(const float* x, const float* weight, const float* bias,
                                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                           float* y_amax, float* save_mean, float* save_invstd,
                                           const float* partials, int64_t n_partials,
                                           float* workspace, int64_t workspace_floats,
                                           int64_t N, int64_t C, int64_t HW, float momentum, float eps,
                                           int act, float slope, void* stream);
int sstem_batchnorm_train_backward_amax_f32(const float* dy, const float* x, const float* weight, const float* bias,
                                            const float* save_mean, const float* save_invstd,
                                            float* dx, float* dx_amax, float* dweight, float* dbias,
                                            float* workspace, int64_t workspace_floats,
                                            int64_t N, int64_t C, int64_t HW, int act, float slope, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_NORM_H */
