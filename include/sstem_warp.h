/*
 * sstem_warp.h -- C-ABI of the bilinear back-warp of libsstem_hip.so (MI355X / gfx950).
 *
 * Replaces the reference's pure-torch module SpatialTransformation.forward / .interpolate
 * (sff_scripts_fusion/utils/image_warp_torch.py:35-112), the step between the frozen flow network and
 * the fusion UNet (sff_scripts_fusion/main_fusion.py:229-235, inference.py:150), with one gather kernel.
 *
 *   image  [B, C, H, W]  fp32 contiguous (the reference permutes to NHWC and zero-pads by 1 pixel; neither
 *                        copy is materialised here)
 *   flow   [B, 2, H, W]  fp32 contiguous; channel 0 = dx (columns), channel 1 = dy (rows) -- the layout the
 *                        flow network produces; the reference's deformation_matrix[b,y,x,k] is flow[b,k,y,x]
 *   output [B, C, H, W]
 * Same status codes / error reporting / stream and ownership rules as sstem_sepconv.h.
 */
#ifndef SSTEM_WARP_H
#define SSTEM_WARP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int sstem_warp_bilinear_f32(const float* image, const float* flow, float* output,
                            int64_t B, int64_t C, int64_t H, int64_t W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSTEM_WARP_H */
