#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {
hipError_t launch_upsample_bilinear2x(const float* in, float* out, int64_t planes, int H, int W, hipStream_t s);
hipError_t launch_upsample_bilinear2x_backward(const float* g, float* gin, int64_t planes, int H, int W, hipStream_t s);
hipError_t launch_pool2x2_forward(const float* in, float* out, uint8_t* idx, int64_t planes, int H, int W, int is_max, hipStream_t s);
hipError_t launch_pool2x2_backward(const float* g, const uint8_t* idx, float* gin, int64_t planes, int H, int W, int is_max, hipStream_t s);
hipError_t launch_gray_u8_to_f32(const uint8_t* img, float* out, int64_t npix, int replicas, hipStream_t s);
hipError_t launch_f32_to_gray_u8(const float* pred, uint8_t* out, int64_t npix, int clamp01, hipStream_t s);
hipError_t launch_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, hipStream_t s);
hipError_t launch_l1_mean_fwd_grad(const float* pred, const float* target, int64_t n, float* loss, float* grad, float* ws, hipStream_t s);
}
