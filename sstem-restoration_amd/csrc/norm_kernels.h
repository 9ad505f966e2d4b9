// Internal launcher interface between the C-ABI (sstem_capi.hip) and the train-mode BatchNorm kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {
int64_t bn_workspace_floats(int64_t N, int64_t C, int64_t HW);
hipError_t launch_bn_train_forward(const float* x, const float* weight, const float* bias, float* running_mean,
                                   float* running_var, float* y, float* save_mean, float* save_invstd, float* workspace,
                                   int N, int C, int64_t HW, float momentum, float eps, int act, float slope, hipStream_t s,
                                   const float* partials = nullptr, int64_t n_partials = 0, long long* num_batches_tracked = nullptr,
                                   float* y_amax = nullptr);
hipError_t launch_bn_train_backward(const float* dy, const float* x, const float* weight, const float* bias,
                                    const float* save_mean, const float* save_invstd, float* dx, float* dweight,
                                    float* dbias, float* workspace, int N, int C, int64_t HW, int act, float slope,
                                    hipStream_t s, int accumulate = 0, float* dx_amax = nullptr);
}  // namespace sstem
