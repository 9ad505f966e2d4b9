// Micro-benchmark: the store phase of a convolution tile (64 channels x 8 rows x 32 pixels, fp32 NCHW) on its own -- how fast a
// workgroup gets its 64 KB out as a function of the plane size (the distance between the 64 channel planes a tile writes to).
//   hipcc -O3 --offload-arch=gfx950 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) float gfloat_t;

// MODE 0: the kernel's order (channel outer, row inner); 1: nontemporal stores; 2: one dword load per channel first (TLB warm-up);
// 3: row outer, channel inner; 4: blocked layout [H][W/64][C][64]
template <int MODE>
__global__ __launch_bounds__(256) void store_tiles(float* out, int C, int H, int W, int tiles_per_wg, float val)
{
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int wco = wave & 1, wr = wave >> 1;
    const int tx = W / 32, ty = H / 8, ncb = C / 64;
    const int64_t plane = (int64_t)H * W;
    const int total = tx * ty * ncb;
    float acc = val;
    for (int t = 0; t < tiles_per_wg; ++t) {
        // XCD-aware: workgroup b walks tiles b, b + G, ... of its XCD's run (as the convolution does)
        uint32_t lin = blockIdx.x + (uint32_t)t * gridDim.x;
        if (lin >= (uint32_t)total) break;
        const uint32_t k8 = lin & 7u, q8 = total >> 3;
        uint32_t u = k8 * q8 + (lin >> 3);
        const int cb = u % ncb; u /= ncb;
        const int bx = u % tx; const int by = u / tx;
        const int co0 = cb * 64 + wco * 32, X0 = bx * 32, Y0 = by * 8 + wr * 4;
        if (MODE == 2) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (r == 0) s += out[(int64_t)co * plane + (int64_t)Y0 * W + X0];
            }
            if (s == 1234.5f) acc += 1.f;
        }
        if (MODE == 3) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    out[(int64_t)co * plane + (int64_t)(Y0 + rr) * W + X0 + r] = acc + q;
                }
        } else if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    out[(((int64_t)(Y0 + rr) * (W / 64) + (X0 >> 6)) * C + co) * 64 + (X0 & 63) + r] = acc + q;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float* p = out + (int64_t)co * plane + (int64_t)(Y0 + rr) * W + X0 + r;
                    if (MODE == 1) __builtin_nontemporal_store(acc + q, p); else *p = acc + q;
                }
            }
        }
    }
}

template <int MODE>
static void run(float* out, int C, int HW, int wgs_per_cu)
{
    const int tiles = (HW / 32) * (HW / 8) * (C / 64);
    const int G = 256 * wgs_per_cu;
    const int per = (tiles + G - 1) / G;
    const int lds = wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(store_tiles<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_tiles<MODE>, dim3(G), dim3(256), lds, 0, out, C, HW, HW, per, 1.f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_tiles<MODE>, dim3(G), dim3(256), lds, 0, out, C, HW, HW, per, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double bytes = (double)C * HW * HW * 4;
    printf("mode %d  C %4d  %4dx%-4d (plane %6.0f KB)  %d wg/CU: %.3f ms  %.2f TB/s  %.2f us per tile per workgroup\n", MODE, C, HW, HW,
           HW * HW * 4 / 1024.0, wgs_per_cu, ms, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / per);
}

int main()
{
    float* out;
    const size_t bytes = (size_t)1 << 30;           // 1 GiB
    hipMalloc(&out, bytes);
    hipMemset(out, 0, bytes);
    for (int w = 1; w <= 2; ++w) {
        run<0>(out, 2048, 128, w); run<0>(out, 1024, 256, w); run<0>(out, 256, 512, w); run<0>(out, 64, 1024, w); run<0>(out, 256, 1024, w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<1>(out, 256, 512, w); run<1>(out, 64, 1024, w);
        run<2>(out, 256, 512, w); run<2>(out, 64, 1024, w);
        run<3>(out, 256, 512, w); run<3>(out, 64, 1024, w);
        run<4>(out, 256, 512, w); run<4>(out, 64, 1024, w);
    }
    return 0;
}
