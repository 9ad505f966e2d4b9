// Micro-benchmark: how fast can 256 CUs stream the sepconv coefficient tensors ([B,51,H,W] fp32, two of them) with
//   (a) one dword per lane (64 consecutive pixels of one tap per wave-instruction -- what sepconv_gray_mfma does),
//   (b) 16 bytes per lane (4 pixels x 16 lanes x 4 taps per wave-instruction),
// summing into a register (one store per thread at the end).  Build: hipcc --offload-arch=gfx950 -O3 coef_stream.hip -o coef_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

constexpr int F = 51;
typedef float f4 __attribute__((ext_vector_type(4)));

// grid: (W/64, H/ROWS, B); block 256 = 4 waves, each wave owns ROWS/4 rows of a 64-pixel column strip
template <int ROWS>
__global__ __launch_bounds__(256) void stream_dword(const float* __restrict__ v, const float* __restrict__ h, float* __restrict__ out, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = (int64_t)H * W;
    const int x = blockIdx.x * 64 + lane;
    float acc = 0.f;
    for (int r = 0; r < ROWS / 4; ++r) {
        const int y = blockIdx.y * ROWS + wave * (ROWS / 4) + r;
        const float* pv = v + (int64_t)blockIdx.z * F * plane + (int64_t)y * W + x;
        const float* ph = h + (int64_t)blockIdx.z * F * plane + (int64_t)y * W + x;
        float tv[F], th[F];
#pragma unroll
        for (int t = 0; t < F; ++t) { tv[t] = pv[t * plane]; th[t] = ph[t * plane]; }
#pragma unroll
        for (int t = 0; t < F; ++t) acc += tv[t] * th[t];
    }
    out[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = acc;
}

// same bytes, 16 B per lane: lane = (tap sub-index t4 = lane >> 4, pixel group q = lane & 15) -> 4 taps x 64 pixels per instruction
template <int ROWS>
__global__ __launch_bounds__(256) void stream_x4(const float* __restrict__ v, const float* __restrict__ h, float* __restrict__ out, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = (int64_t)H * W;
    const int q = lane & 15, t4 = lane >> 4;
    const int x = blockIdx.x * 64 + 4 * q;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < ROWS / 4; ++r) {
        const int y = blockIdx.y * ROWS + wave * (ROWS / 4) + r;
        const float* pv = v + (int64_t)blockIdx.z * F * plane + (int64_t)y * W + x;
        const float* ph = h + (int64_t)blockIdx.z * F * plane + (int64_t)y * W + x;
        f4 tv[13], th[13];
#pragma unroll
        for (int g = 0; g < 13; ++g) {
            int t = 4 * g + t4; if (t > F - 1) t = F - 1;         // 52nd slot: re-reads tap 50 (same bytes as padding would cost)
            tv[g] = *reinterpret_cast<const f4*>(pv + t * plane);
            th[g] = *reinterpret_cast<const f4*>(ph + t * plane);
        }
#pragma unroll
        for (int g = 0; g < 13; ++g) acc += tv[g] * th[g];
    }
    out[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    const int B = 8, H = 1024, W = 1024;
    const size_t n = (size_t)B * F * H * W;
    float *v, *h, *out;
    hipMalloc(&v, n * 4); hipMalloc(&h, n * 4); hipMalloc(&out, (size_t)B * H * W * 4);
    hipMemset(v, 0, n * 4); hipMemset(h, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 2.0 * n * 4;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    run("dword/lane, 16 rows per workgroup", [&] { hipLaunchKernelGGL(stream_dword<16>, dim3(W / 64, H / 16, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 64 rows per workgroup", [&] { hipLaunchKernelGGL(stream_dword<64>, dim3(W / 64, H / 64, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("16 B/lane,  16 rows per workgroup", [&] { hipLaunchKernelGGL(stream_x4<16>, dim3(W / 64, H / 16, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("16 B/lane,  64 rows per workgroup", [&] { hipLaunchKernelGGL(stream_x4<64>, dim3(W / 64, H / 64, B), dim3(256), 0, 0, v, h, out, H, W); });
    return 0;
}
