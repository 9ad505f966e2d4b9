/*
 * sstem_resize.h -- C-ABI of the bilinear x2 up-sampling of libsstem_hip.so (MI355X / gfx950).
 *
 * Replaces torch's kernel behind the reference's nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
 * (sff_scripts_interp/model/model_interp.py:17 -- used by _upsample_module :139-143 and inside every kernel head
 * _kernel_module :129-137; sp_scripts_train/networks.py:27 and Up :213) on the inference path: at C2 the four kernel
 * heads of an IFNet each write a [8,51,1024,1024] tensor through it.  Same formula as PyTorch
 * (src = dst * (in-1)/(out-1), two-tap interpolation per axis), fp32.
 *
 *   input  [planes, H, W]      fp32 contiguous (planes = N*C), W even
 *   output [planes, 2H, 2W]    fp32 contiguous, 16-byte aligned
 * Forward only (training keeps torch's differentiable op).  Same status codes / error reporting / stream and
 * ownership rules as sstem_sepconv.h.
 */
#ifndef SSTEM_RESIZE_H
#define SSTEM_RESIZE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int sstem_upsample_bilinear2x_f32(const float* input, float* output, int64_t planes, int64_t H, int64_t W,
                                  void* stream);

/* Gradient of the above with respect to its input: grad_output [planes, 2H, 2W] -> grad_input [planes, H, W] (fully overwritten).
 * A gather with the forward kernel's own weights: deterministic (no atomics), any H, W.  Replaces autograd's
 * upsample_bilinear2d_backward under the reference's nn.Upsample in training (model_interp.py:17, networks.py:27). */
int sstem_upsample_bilinear2x_backward_f32(const float* grad_output, float* grad_input, int64_t planes, int64_t H, int64_t W,
                                           void* stream);

/* 2 x 2 / stride 2 pooling of [planes, H, W] -> [planes, H/2, W/2] (floor, like torch): is_max != 0 nn.MaxPool2d(2)
 * (model_unet.py:35-39, model_fusionnet.py:24, networks.py:134), else nn.AvgPool2d((2,2),(2,2)) (model_interp.py:27, networks.py:33).
 * torch's arithmetic: the average is ((a + b) + c) + d in row-major window order, times 0.25; the maximum is the first largest
 * element in that order (NaN propagates).  argmax (is_max only; nullable in the forward when no backward follows): one byte per
 * output = the winner's position 0..3 in the window; the backward routes grad_output through it (average: grad / 4 to all four) and
 * writes EVERY element of grad_input (rows / columns a floor-ed window does not cover get 0). */
int sstem_pool2x2_forward_f32(const float* input, float* output, uint8_t* argmax, int64_t planes, int64_t H, int64_t W, int is_max,
                              void* stream);
int sstem_pool2x2_backward_f32(const float* grad_output, const uint8_t* argmax, float* grad_input, int64_t planes, int64_t H, int64_t W,
                               int is_max, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_RESIZE_H */
