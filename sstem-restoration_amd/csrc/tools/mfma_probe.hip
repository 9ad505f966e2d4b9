// Hardware probe (developer tool, not part of the library): prints the lane/register layout of
// v_mfma_f32_4x4x1_16b_f32 on the device and measures its issue rate, alone and with VALU FMAs
// interleaved.  The sepconv kernels assume: A lane l -> (block l>>2, row l&3); B lane l ->
// (block l>>2, col l&3); D reg i of lane l -> (block l>>2, row i, col l&3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void probe(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

template <int NV>
__global__ __launch_bounds__(512) void rate(float* out, int iters, float seed)
{
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, acc2, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k % 8] = fmaf(v[k % 8], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
static void run_rate(float* dout, const char* name)
{
    const int iters = 2000, blocks = 256 * 2, threads = 512;  // 2 WG x 8 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate<NV>, dim3(blocks), dim3(threads), 0, 0, dout, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<NV>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * threads / 64;
    const double mfma = waves * iters * 24.0;
    const double fl = mfma * 512 + waves * iters * 8.0 * NV * 128;
    printf("%-28s %8.3f ms  mfma/s %.3e  (%.1f cyc/mfma/SIMD @2.4GHz)  total %.1f TFLOP/s\n", name, ms,
           mfma / (ms * 1e-3), (ms * 1e-3) * 2.4e9 / (mfma / 1024.0), fl / (ms * 1e-3) / 1e12);
}

int main()
{
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    printf("device %s  CUs %d  clock %d kHz  LDS/block %zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    float *da, *db, *dd; CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 1024));
    std::vector<float> a(64), b(64), d(256);
    int bad = 0;
    // A layout: A lane la = 1, B all = lane-dependent id
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) { a[l] = (l == la) ? 1.f : 0.f; b[l] = 100.f + l; }
        CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd); CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
        // expect: nonzero at lanes 4*(la>>2)+j, reg la&3, value 100 + that lane
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
            const float exp = ((l >> 2) == (la >> 2) && i == (la & 3)) ? 100.f + l : 0.f;
            if (d[l * 4 + i] != exp) { if (bad < 10) printf("MISMATCH la=%d lane=%d reg=%d got %g exp %g\n", la, l, i, d[l*4+i], exp); ++bad; }
        }
    }
    printf("4x4x1_16B layout check: %s (%d mismatches)\n", bad ? "DIFFERENT FROM ASSUMED" : "as assumed", bad);
    if (bad) {  // dump the raw map for A lane 0..7
        for (int la = 0; la < 8; ++la) {
            for (int l = 0; l < 64; ++l) { a[l] = (l == la) ? 1.f : 0.f; b[l] = 100.f + l; }
            CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd); CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
            printf("A lane %d ->", la);
            for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (d[l*4+i] != 0.f) printf(" (lane %d reg %d = %g)", l, i, d[l*4+i]);
            printf("\n");
        }
    }
    float* dout; CK(hipMalloc(&dout, 512 * 512 * 4));
    run_rate<0>(dout, "mfma only");
    run_rate<1>(dout, "mfma + 1 v_fma per 3 mfma");
    run_rate<3>(dout, "mfma + 3 v_fma per 3 mfma");
    run_rate<6>(dout, "mfma + 6 v_fma per 3 mfma");
    run_rate<12>(dout, "mfma + 12 v_fma per 3 mfma");
    return bad ? 2 : 0;
}
