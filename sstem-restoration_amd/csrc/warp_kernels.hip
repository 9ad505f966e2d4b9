// Bilinear back-warp of an image by a dense flow field (one gather kernel, HBM-bound).
//
// Replaces the ~20 torch ops of the reference's SpatialTransformation
// (sff_scripts_fusion/utils/image_warp_torch.py:5-112: NHWC permute, 1-px zero pad, meshgrid, floor,
// clamp, four gathers on a flattened copy, weighted sum) used between the frozen flow network and the
// fusion UNet (sff_scripts_fusion/main_fusion.py:229-235).  Semantics kept exactly:
//   x = dx + col + 1, y = dy + row + 1                      (:104-105, :44-45; +1 = the zero border)
//   x0 = floor(x), x1 = x0 + 1, both clamped to [0, W+1]; same for y         (:50-58)
//   weights from the CLAMPED x1, y1:  wx = x1 - x, wy = y1 - y                (:87-93)
//   out = wx*wy*I(y0,x0) + wx*(1-wy)*I(y1,x0) + (1-wx)*wy*I(y0,x1) + (1-wx)*(1-wy)*I(y1,x1)   (:95-98)
//   I(.) reads the zero-padded image: indices 0 and W+1 / H+1 are the border.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "warp_kernels.h"

namespace sstem {

__global__ __launch_bounds__(256) void warp_bilinear(
    const float* __restrict__ img, const float* __restrict__ flow, float* __restrict__ out,
    int B, int C, int H, int W)
{
    const int64_t plane = (int64_t)H * W;
    const int64_t n = (int64_t)B * plane;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / plane;
        const int64_t yx = p - b * plane;
        const int row = (int)(yx / W), col = (int)(yx - (int64_t)row * W);
        const float x = (flow[(b * 2 + 0) * plane + yx] + (float)col) + 1.0f;
        const float y = (flow[(b * 2 + 1) * plane + yx] + (float)row) + 1.0f;
        // floor -> int64 like torch's .long(); clamp to the padded image
        const float fx0 = floorf(x), fy0 = floorf(y);
        const float lim = 4.0e18f;   // keep the float->int64 conversion defined for absurd flows
        int64_t x0 = (int64_t)fminf(fmaxf(fx0, -lim), lim), y0 = (int64_t)fminf(fmaxf(fy0, -lim), lim);
        int64_t x1 = x0 + 1, y1 = y0 + 1;
        const int64_t max_x = W + 1, max_y = H + 1;
        x0 = x0 < 0 ? 0 : (x0 > max_x ? max_x : x0);
        x1 = x1 < 0 ? 0 : (x1 > max_x ? max_x : x1);
        y0 = y0 < 0 ? 0 : (y0 > max_y ? max_y : y0);
        y1 = y1 < 0 ? 0 : (y1 > max_y ? max_y : y1);
        const float wx = (float)x1 - x, wy = (float)y1 - y;
        const float wa = wx * wy, wb = wx * (1.0f - wy), wc = (1.0f - wx) * wy, wd = (1.0f - wx) * (1.0f - wy);
        const bool x0in = x0 >= 1 && x0 <= W, x1in = x1 >= 1 && x1 <= W;
        const bool y0in = y0 >= 1 && y0 <= H, y1in = y1 >= 1 && y1 <= H;
        const int64_t oa = (y0 - 1) * W + (x0 - 1), ob = (y1 - 1) * W + (x0 - 1);
        const int64_t oc = (y0 - 1) * W + (x1 - 1), od = (y1 - 1) * W + (x1 - 1);
        for (int c = 0; c < C; ++c) {
            const float* ip = img + (b * C + c) * plane;
            const float ia = (y0in && x0in) ? ip[oa] : 0.f;
            const float ib = (y1in && x0in) ? ip[ob] : 0.f;
            const float ic = (y0in && x1in) ? ip[oc] : 0.f;
            const float id = (y1in && x1in) ? ip[od] : 0.f;
            // separate products, then summed in stack order (torch.sum over the 4 stacked terms)
            const float s = ((__fmul_rn(wa, ia) + __fmul_rn(wb, ib)) + __fmul_rn(wc, ic)) + __fmul_rn(wd, id);
            out[(b * C + c) * plane + yx] = s;
        }
    }
}

hipError_t launch_warp_bilinear(const float* img, const float* flow, float* out, int B, int C, int H, int W,
                                hipStream_t s)
{
    int64_t n = (int64_t)B * H * W;
    int64_t g = (n + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(warp_bilinear, dim3((unsigned)g), dim3(256), 0, s, img, flow, out, B, C, H, W);
    return hipGetLastError();
}

}  // namespace sstem
