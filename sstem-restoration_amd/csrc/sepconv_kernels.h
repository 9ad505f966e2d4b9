// Internal launcher interface between the C-ABI (sstem_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sstem {

hipError_t launch_fwd_direct(const float* in, const float* ver, const float* hor, float* out,
                             int64_t B, int64_t C, int64_t H, int64_t W, int filt, hipStream_t s);
hipError_t launch_bwd_direct(const float* g, const float* in, const float* ver, const float* hor,
                             float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                             int filt, hipStream_t s);
hipError_t launch_fwd_mfma(const float* in, const float* ver, const float* hor, float* out,
                           int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s);
hipError_t launch_bwd_mfma(const float* g, const float* in, const float* ver, const float* hor,
                           float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W,
                           hipStream_t s);
bool mfma_grid_ok(int64_t B, int64_t H, int64_t W);
hipError_t launch_interp_fused(const float* i1, const float* i2, const float* k1v, const float* k1h,
                               const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                               hipStream_t s);

bool interp_fused_gray_ok(int64_t H, int64_t W);
hipError_t launch_interp_fused_gray(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                    const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                    hipStream_t s, uint8_t* out_u8 = nullptr);

// the same apply with the four coefficient tensors in the row-segment layout [B][H][ceil(W/64)][51][64]
int64_t coef_blocked_floats(int64_t B, int64_t H, int64_t W);
bool interp_fused_gray_blocked_ok(int64_t H, int64_t W);
hipError_t launch_coef_to_blocked(const float* src, float* dst, int64_t B, int64_t H, int64_t W, hipStream_t s);
hipError_t launch_interp_fused_gray_blocked(const float* g1, const float* g2, const float* k1v, const float* k1h,
                                            const float* k2v, const float* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                            hipStream_t s, uint8_t* out_u8 = nullptr);

// bf16 coefficient tensors ([B,51,H,W] bf16), fp32 frames, gradients and sums
hipError_t launch_fwd_bf16coef(const float* in, const uint16_t* ver, const uint16_t* hor, float* out,
                               int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s);
hipError_t launch_bwd_bf16coef(const float* g, const float* in, const uint16_t* ver, const uint16_t* hor,
                               float* gv, float* gh, int64_t B, int64_t C, int64_t H, int64_t W, hipStream_t s);
bool interp_fused_gray_bf16coef_ok(int64_t H, int64_t W);
hipError_t launch_interp_fused_gray_bf16coef(const float* g1, const float* g2, const uint16_t* k1v, const uint16_t* k1h,
                                             const uint16_t* k2v, const uint16_t* k2h, float* out, int64_t B, int64_t H, int64_t W,
                                             hipStream_t s);

}  // namespace sstem
