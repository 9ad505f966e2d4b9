/*
 * sstem_io.h -- C-ABI of the uint8 <-> fp32 image edge and the flat Adam update (libsstem_hip.so).
 *
 * sstem_gray_u8_to_f32    replaces  inputs.astype(np.float32) / 255.0 with np.repeat(img, 3, 0)
 *                                   (sff_scripts_interp/inference_singleImage.py:55-66) and Gray2Tensor
 *                                   (sp_scripts_test/utils/gray2tensor.py:7-12): out[r*npix + i] = img[i]/255
 *                                   for r < replicas (the replicated channels are identical copies).
 * sstem_f32_to_gray_u8    replaces  (pred * 255).astype(np.uint8)  (inference_singleImage.py:76,
 *                                   gray2tensor.py:14-24): fp32 multiply, truncation toward zero, low 8 bits --
 *                                   NO clamp, exactly like numpy on x86-64 (256.0 -> 0, -1.0 -> 255);
 *                                   clamp01 != 0 first clamps pred to [0,1] (TrainTensor2mask, :26-31).
 * sstem_adam_step_f32     replaces  torch.optim.Adam.step (sff_scripts_interp/main_ms.py:211,315;
 *                                   betas (0.9, 0.999), eps 1e-8) for parameters and gradients that live in
 *                                   one flat buffer each: one launch instead of ~10 per tensor.
 *                                   `step` is the 1-based update count (bias correction 1 - beta^step).
 * sstem_l1_mean_forward_grad_f32  replaces  nn.L1Loss()(pred, target) + its backward (sff_scripts_fusion/main_fusion.py:252,
 *                                   sff_scripts_interp/main_ms.py:205, sp_scripts_train/main_fusion.py:237-249): *loss = mean |pred -
 *                                   target| and grad = sign(pred - target) / n (sign(0) = 0) in ONE launch, deterministic
 *                                   (fixed-order sums).  workspace: sstem_l1_workspace_floats() floats, zero before the first
 *                                   call, reusable afterwards without another fill (the launch leaves it clean), one per stream.
 * Device pointers; same status codes / stream / ownership rules as sstem_sepconv.h.
 */
#ifndef SSTEM_IO_H
#define SSTEM_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int sstem_gray_u8_to_f32(const uint8_t* image, float* output, int64_t npix, int64_t replicas, void* stream);
int sstem_f32_to_gray_u8(const float* pred, uint8_t* output, int64_t npix, int clamp01, void* stream);
int sstem_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                        void* stream);
int64_t sstem_l1_workspace_floats(void);
int sstem_l1_mean_forward_grad_f32(const float* pred, const float* target, int64_t n, float* loss, float* grad, float* workspace,
                                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_IO_H */
