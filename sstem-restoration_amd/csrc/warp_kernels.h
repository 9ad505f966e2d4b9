#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {
hipError_t launch_warp_bilinear(const float* img, const float* flow, float* out, int B, int C, int H, int W,
                                hipStream_t s);
}
