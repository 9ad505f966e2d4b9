// Micro-test: is the SGPR offset of a raw buffer store part of the range check on gfx950?  A buffer resource of 1024 bytes over the first
// quarter of a 4096-byte allocation; lane i stores to voffset = 4 i with soffset = 0 / 512 / 1024 / 2048.  Prints which dwords changed.
//   hipcc -O3 --offload-arch=gfx950 -o buffer_soffset_range buffer_soffset_range.hip && ./buffer_soffset_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(float* p, int soff)
{
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 1024, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u, r, (int)(threadIdx.x * 4), soff, 0);
}
int main()
{
    float* d; float h[1024];
    hipMalloc(&d, 4096);
    for (int soff : {0, 512, 1024, 2048}) {
        hipMemset(d, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, soff);
        hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        int first = -1, last = -1, n = 0;
        for (int i = 0; i < 1024; ++i) if (h[i] != 0.f) { if (first < 0) first = i; last = i; ++n; }
        printf("soffset %4d (256 lanes, voffset 0..1020, num_records 1024): %d dwords written, dword %d .. %d\n", soff, n, first, last);
    }
    return 0;
}
