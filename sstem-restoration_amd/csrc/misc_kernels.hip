// HBM-bound edge kernels of the hot path's callers:
//   * uint8 image <-> fp32 tensor conversions of the inference scripts
//       sff_scripts_interp/inference_singleImage.py:55-66,76  (x3 channel replicate, /255, *255 truncation)
//       sp_scripts_test/utils/gray2tensor.py:7-20
//   * Adam update over one flat parameter / gradient buffer (torch.optim.Adam semantics as used by
//       sff_scripts_interp/main_ms.py:315, sff_scripts_fusion/main_fusion.py, betas 0.9/0.999, eps 1e-8)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "misc_kernels.h"

namespace sstem {

__global__ __launch_bounds__(256) void gray_u8_to_f32(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                      int64_t npix, int replicas)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = __fdiv_rn((float)img[i], 255.0f);     // == numpy float32(k) / float32(255)
        for (int r = 0; r < replicas; ++r) out[(int64_t)r * npix + i] = v;
    }
}

__global__ __launch_bounds__(256) void f32_to_gray_u8(const float* __restrict__ pred, uint8_t* __restrict__ out,
                                                      int64_t npix, int clamp01)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        float p = pred[i];
        if (clamp01) p = p > 1.f ? 1.f : (p < 0.f ? 0.f : p);          // TrainTensor2mask, gray2tensor.py:26-31
        const float v = __fmul_rn(p, 255.0f);
        // numpy's float -> uint8 astype on x86-64: truncate toward zero to a wide integer, keep the low 8 bits
        // (no clamp: 256.0 -> 0, -1.0 -> 255); NaN and |v| >= 2^63 -> 0
        long long w = 0;
        if (v == v && fabsf(v) < 9.0e18f) w = (long long)v;
        out[i] = (uint8_t)(w & 0xFF);
    }
}

// p, g, m, v: flat fp32 buffers of n elements.  torch.optim.Adam (no amsgrad, L2 weight decay added to the
// gradient): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_step(float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                                 float bc1, float bc2_sqrt)
{
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);            // lerp form, as torch does
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

static inline int grid_1d(int64_t n)
{
    int64_t g = (n + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_gray_u8_to_f32(const uint8_t* img, float* out, int64_t npix, int replicas, hipStream_t s)
{
    hipLaunchKernelGGL(gray_u8_to_f32, dim3(grid_1d(npix)), dim3(256), 0, s, img, out, npix, replicas);
    return hipGetLastError();
}

hipError_t launch_f32_to_gray_u8(const float* pred, uint8_t* out, int64_t npix, int clamp01, hipStream_t s)
{
    hipLaunchKernelGGL(f32_to_gray_u8, dim3(grid_1d(npix)), dim3(256), 0, s, pred, out, npix, clamp01);
    return hipGetLastError();
}

hipError_t launch_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, hipStream_t s)
{
    hipLaunchKernelGGL(adam_step, dim3(grid_1d(n)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2_sqrt);
    return hipGetLastError();
}

}  // namespace sstem
