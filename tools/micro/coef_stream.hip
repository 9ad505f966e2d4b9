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

// 64-pixel strips x ROWS rows as stream_dword, but the linear workgroup id is re-mapped so that the workgroups that land on one
// XCD (id % 8) own horizontally adjacent strips: MAP 1: strip = (id % 8) * (tiles_x / 8) + (id / 8) % (tiles_x / 8)
template <int ROWS, int MAP>
__global__ __launch_bounds__(256) void stream_dword_remap(const float* __restrict__ v, const float* __restrict__ h, float* __restrict__ out, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = (int64_t)H * W;
    const int tiles_x = W / 64, tiles_y = H / ROWS;
    int id = blockIdx.x;                         // 1-D grid: tiles_x * tiles_y * B
    const int per_img = tiles_x * tiles_y;
    const int b = id / per_img; id -= b * per_img;
    int tx, ty;
    if (MAP == 1) {                              // XCD k owns strips [k * tiles_x/8, (k+1) * tiles_x/8)
        const int k = id % 8, rest = id / 8, sub = tiles_x / 8;
        tx = k * sub + rest % sub; ty = rest / sub;
    } else if (MAP == 2) {                       // XCD k owns rows of tiles: ty = k + 8 * (...), all strips of a row consecutive in time
        const int k = id % 8, rest = id / 8;
        tx = rest % tiles_x; ty = (rest / tiles_x) * 8 + k;
    } else { tx = id % tiles_x; ty = id / tiles_x; }
    const int x = tx * 64 + lane;
    float acc = 0.f;
    for (int r = 0; r < ROWS / 4; ++r) {
        const int y = ty * ROWS + wave * (ROWS / 4) + r;
        const float* pv = v + (int64_t)b * F * plane + (int64_t)y * W + x;
        const float* ph = h + (int64_t)b * F * plane + (int64_t)y * W + x;
        float tv[F], th[F];
#pragma unroll
        for (int t = 0; t < F; ++t) { tv[t] = pv[t * plane]; th[t] = ph[t * plane]; }
#pragma unroll
        for (int t = 0; t < F; ++t) acc += tv[t] * th[t];
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// WX waves side by side x WY = 4 / WX waves interleaved in y (wave (wx, wy) owns rows wy, wy + WY, ... of strip wx): WX = 1 is the
// geometry of sepconv_gray_mfma (a wave's rows are WAVES apart), WX = 4 the "wide" one
template <int WX, int RPW>
__global__ __launch_bounds__(256) void stream_geom(const float* __restrict__ v, const float* __restrict__ h, float* __restrict__ out, int H, int W)
{
    constexpr int WY = 4 / WX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wx = wave % WX, wy = wave / WX;
    const int64_t plane = (int64_t)H * W;
    const int x = blockIdx.x * 64 * WX + wx * 64 + lane;
    float acc = 0.f;
    for (int r = 0; r < RPW; ++r) {
        const int y = blockIdx.y * (RPW * WY) + wy + r * WY;
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

// waves side by side: workgroup = 256 pixels wide x ROWS rows, each wave walks all ROWS rows of its 64-pixel strip
template <int ROWS>
__global__ __launch_bounds__(256) void stream_dword_wide(const float* __restrict__ v, const float* __restrict__ h, float* __restrict__ out, int H, int W)
{
    const int64_t plane = (int64_t)H * W;
    const int x = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < ROWS; ++r) {
        const int y = blockIdx.y * ROWS + r;
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
    run("dword/lane, 64 rows, 1-D grid plain", [&] { hipLaunchKernelGGL((stream_dword_remap<64, 0>), dim3(W / 64 * H / 64 * B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 64 rows, XCD owns strips", [&] { hipLaunchKernelGGL((stream_dword_remap<64, 1>), dim3(W / 64 * H / 64 * B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 64 rows, XCD owns tile rows", [&] { hipLaunchKernelGGL((stream_dword_remap<64, 2>), dim3(W / 64 * H / 64 * B), dim3(256), 0, 0, v, h, out, H, W); });
    run("geom 1x4 (kernel today), 16 rows/wave", [&] { hipLaunchKernelGGL((stream_geom<1, 16>), dim3(W / 64, H / 64, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("geom 2x2, 16 rows/wave", [&] { hipLaunchKernelGGL((stream_geom<2, 16>), dim3(W / 128, H / 32, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("geom 4x1, 16 rows/wave", [&] { hipLaunchKernelGGL((stream_geom<4, 16>), dim3(W / 256, H / 16, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("geom 2x2, 32 rows/wave", [&] { hipLaunchKernelGGL((stream_geom<2, 32>), dim3(W / 128, H / 64, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 256 px wide x 4 rows", [&] { hipLaunchKernelGGL(stream_dword_wide<4>, dim3(W / 256, H / 4, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 256 px wide x 16 rows", [&] { hipLaunchKernelGGL(stream_dword_wide<16>, dim3(W / 256, H / 16, B), dim3(256), 0, 0, v, h, out, H, W); });
    run("dword/lane, 256 px wide x 32 rows", [&] { hipLaunchKernelGGL(stream_dword_wide<32>, dim3(W / 256, H / 32, B), dim3(256), 0, 0, v, h, out, H, W); });
    return 0;
}
