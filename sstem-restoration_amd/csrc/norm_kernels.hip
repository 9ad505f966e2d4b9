// Train-mode BatchNorm2d (+ ReLU / LeakyReLU) for MI355X: batch statistics, normalisation and activation in two streaming
// passes, and the matching backward in two passes.
//
// Replaces torch's kernels behind the train-mode nn.BatchNorm2d (+ activation) runs of the reference's blocks
//   sp_scripts_train/networks.py:179-186 (DoubleConv), sff_scripts_fusion/model/model_unet.py:11-48,
//   sff_scripts_fusion/model/model_fusionnet.py:12-43 (conv_block / conv_trans_block)
// Measured on MI355X (tools/bench_bn.py): torch's train-mode BatchNorm + ReLU takes 0.47 ms forward and 0.54 ms backward on a
// 16x32x256x256 activation (0.9-1.2 TB/s over the bytes a two-pass scheme moves); these kernels are HBM-streaming.
//
// Layout NCHW fp32.  Channel c of sample n is one contiguous plane of HW floats; a workgroup owns (channel, chunk), a chunk
// being up to CHUNK consecutive floats of one plane.  Forward pass 1 writes per-chunk (count, mean, M2) triplets -- M2 = sum of
// squared deviations from the chunk's own mean, computed around a pivot inside the chunk, so a channel with |mean| >> std does not
// cancel (torch / the reference use Welford; E[x^2] - E[x]^2 in fp32 does not survive mean/std ~ 1e3) -- or is skipped altogether
// when the producing convolution already wrote such triplets per tile (conv_kernels.hip, ConvExtra::bn_part).  Pass 2 first merges
// the triplets of its channel (Chan's formula, in double, fixed order: the result does not depend on the launch geometry) and then
// streams its chunk.  Semantics are torch's: biased variance for the normalisation, unbiased for running_var,
// running = (1 - momentum) * running + momentum * batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "norm_kernels.h"

namespace sstem {

constexpr int BN_THREADS = 256;
constexpr int BN_CHUNK = 16384;      // floats per (channel, chunk) workgroup at most: 64 per thread

struct BnGeom { int N, C; int64_t HW; int pieces, chunks, chunk_len; };   // pieces per plane, chunks per channel = N * pieces

// Chunk length: 16384 floats on large tensors; on small ones (the layers of a 2-sample training step: 0.5-8 MB) shorter chunks, so
// that the launch still has about a thousand workgroups -- with the fixed length a 2 x 64 x 128 x 128 tensor was 128 workgroups on a
// 256-CU chip and each of the four BatchNorm launches of a layer took 7-10 us (profiles/r02/g_*).  Multiples of 4 floats (16-byte
// loads), a pure function of the sizes (the workspace query and both passes of a direction see the same geometry).
__host__ __device__ inline int bn_chunk_len(int64_t N, int64_t C, int64_t HW)
{
    int len = BN_CHUNK;
    while (len > 1024 && N * C * ((HW + len - 1) / len) < 1024) len >>= 1;
    return len;
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope)
{
    return act == 1 ? (v > 0.f ? v : 0.f) : (act == 2 ? (v > 0.f ? v : v * slope) : v);
}
__device__ __forceinline__ float act_grad(float pre, int act, float slope)      // derivative at the pre-activation value
{
    return act == 1 ? (pre > 0.f ? 1.f : 0.f) : (act == 2 ? (pre > 0.f ? 1.f : slope) : 1.f);
}

// COH (the one-launch forms, round 5): partials written by workgroups on other XCDs are read inside the same launch -- the eight L2s are
// not coherent with one another within a launch, so those few floats go past them (agent-scope stores and loads)
template <bool COH> __device__ __forceinline__ float ld_part(const float* p)
{
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool COH> __device__ __forceinline__ void st_part(float* p, float v)
{
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

// fixed-shape block reduction of two doubles (tree over 256 threads): deterministic
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh)
{
    sh[threadIdx.x] = a; sh[BN_THREADS + threadIdx.x] = b;
    __syncthreads();
    for (int o = BN_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[threadIdx.x] += sh[threadIdx.x + o]; sh[BN_THREADS + threadIdx.x] += sh[BN_THREADS + threadIdx.x + o]; }
        __syncthreads();
    }
    a = sh[0]; b = sh[BN_THREADS];
    __syncthreads();
}

__device__ __forceinline__ void chunk_span(const BnGeom& gm, int chunk, int c, int64_t& base, int64_t& len)
{
    const int n = chunk / gm.pieces, pc = chunk % gm.pieces;
    const int64_t start = (int64_t)pc * gm.chunk_len;
    len = gm.HW - start < gm.chunk_len ? gm.HW - start : gm.chunk_len;
    base = ((int64_t)n * gm.C + c) * gm.HW + start;
}

// ---- forward pass 1: per-chunk (count, mean, M2) ---------------------------------------------------------------
template <bool COH>
__device__ __forceinline__ void bn_fwd_partial_body(const float* __restrict__ x, float* __restrict__ part, const BnGeom& gm, double* sh)
{
    const int c = blockIdx.y, chunk = blockIdx.x;
    int64_t base, len;
    chunk_span(gm, chunk, c, base, len);
    const float* p = x + base;
    const float pivot = p[0];                   // sums of (x - pivot): the pivot lies inside the data, nothing large cancels
    float s = 0.f, q = 0.f;
    if ((base & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int64_t i = threadIdx.x; i < len / 4; i += BN_THREADS) {
            float4 v = p4[i];
            v.x -= pivot; v.y -= pivot; v.z -= pivot; v.w -= pivot;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (int64_t i = (len / 4) * 4 + threadIdx.x; i < len; i += BN_THREADS) { const float v = p[i] - pivot; s += v; q += v * v; }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += BN_THREADS) { const float v = p[i] - pivot; s += v; q += v * v; }
    }
    double ds = s, dq = q;
    block_sum2(ds, dq, sh);
    if (threadIdx.x == 0) {
        const double n = (double)len, dm = ds / n;
        float* dst = part + ((int64_t)c * gm.chunks + chunk) * 3;
        st_part<COH>(dst + 0, (float)len);
        st_part<COH>(dst + 1, (float)((double)pivot + dm));
        double m2 = dq - ds * dm;               // sum (x - pivot)^2 - n * dm^2
        st_part<COH>(dst + 2, (float)(m2 > 0.0 ? m2 : 0.0));
    }
}
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_partial(const float* __restrict__ x, float* __restrict__ part, BnGeom gm)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_fwd_partial_body<false>(x, part, gm, sh);
}

// sum of this channel's partial pairs, by every workgroup of the channel the same way (backward: plain sums)
template <bool COH = false>
__device__ __forceinline__ void channel_totals(const float* __restrict__ part, int c, int chunks, double& t0, double& t1, double* sh)
{
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < chunks; k += BN_THREADS) {
        a += (double)ld_part<COH>(part + ((int64_t)c * chunks + k) * 2 + 0);
        b += (double)ld_part<COH>(part + ((int64_t)c * chunks + k) * 2 + 1);
    }
    block_sum2(a, b, sh);
    t0 = a; t1 = b;
}

// batch mean and biased variance of channel c from its (count, mean, M2) triplets: mean = sum n_i m_i / n,
// M2 = sum M2_i + sum n_i (m_i - mean)^2 (Chan et al.), two fixed-shape reductions in double
template <bool COH = false>
__device__ __forceinline__ void channel_moments(const float* __restrict__ part, int c, int nparts, double& mean, double& var, double& cnt, double* sh)
{
    double n = 0.0, nm = 0.0;
    for (int k = threadIdx.x; k < nparts; k += BN_THREADS) {
        const float* t = part + ((int64_t)c * nparts + k) * 3;
        const double t0 = (double)ld_part<COH>(t);
        n += t0; nm += t0 * (double)ld_part<COH>(t + 1);
    }
    block_sum2(n, nm, sh);
    mean = nm / n; cnt = n;
    double m2 = 0.0, dummy = 0.0;
    for (int k = threadIdx.x; k < nparts; k += BN_THREADS) {
        const float* t = part + ((int64_t)c * nparts + k) * 3;
        const double d = (double)ld_part<COH>(t + 1) - mean;
        m2 += (double)ld_part<COH>(t + 2) + (double)ld_part<COH>(t) * d * d;
    }
    block_sum2(m2, dummy, sh);
    var = m2 / n;
}

// ---- forward pass 2: statistics of the channel, then y = act((x - mean) * invstd * w + b) -------------------
// the workgroup's largest stored magnitude into an amax word (1024 float slots, include/sstem_conv.h: one atomic per workgroup;
// non-negative floats order like their bit patterns).  `sh` is the kernel's reduction scratch, free again by now.
__device__ __forceinline__ void bn_amax_update(float* __restrict__ word, float m, double* sh)
{
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float* red = reinterpret_cast<float*>(sh);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < BN_THREADS / 64; ++i) m = fmaxf(m, red[i]);
        atomicMax(reinterpret_cast<unsigned int*>(word) + ((blockIdx.y * gridDim.x + blockIdx.x) & 1023u), __builtin_bit_cast(unsigned int, m));
    }
}

template <bool COH>
__device__ __forceinline__ void bn_fwd_apply_body(const float* __restrict__ x, const float* __restrict__ part,
                                                  const float* __restrict__ weight, const float* __restrict__ bias,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  float* __restrict__ y, float* __restrict__ save_mean,
                                                  float* __restrict__ save_invstd, const BnGeom& gm, float momentum, float eps,
                                                  int act, float slope, int nparts, long long* __restrict__ num_batches_tracked,
                                                  float* __restrict__ y_amax, double* sh)
{
    const int c = blockIdx.y, chunk = blockIdx.x;
    double mean_d, var_d, cnt;
    channel_moments<COH>(part, c, nparts, mean_d, var_d, cnt, sh);
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (chunk == 0 && threadIdx.x == 0) {
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;       // torch's bookkeeping, without a launch of its own
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) {
            const double unbiased = cnt > 1.0 ? var_d * cnt / (cnt - 1.0) : var_d;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    const float sc = invstd * (weight ? weight[c] : 1.f);
    const float sf = (bias ? bias[c] : 0.f) - mean * sc;
    int64_t base, len;
    chunk_span(gm, chunk, c, base, len);
    const float* p = x + base;
    float* o = y + base;
    float vmax = 0.f;
    if ((base & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (int64_t i = threadIdx.x; i < len / 4; i += BN_THREADS) {
            const float4 v = p4[i];
            const float4 r = make_float4(act_fwd(v.x * sc + sf, act, slope), act_fwd(v.y * sc + sf, act, slope),
                                         act_fwd(v.z * sc + sf, act, slope), act_fwd(v.w * sc + sf, act, slope));
            o4[i] = r;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
        for (int64_t i = (len / 4) * 4 + threadIdx.x; i < len; i += BN_THREADS) { const float r = act_fwd(p[i] * sc + sf, act, slope); o[i] = r; vmax = fmaxf(vmax, fabsf(r)); }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += BN_THREADS) { const float r = act_fwd(p[i] * sc + sf, act, slope); o[i] = r; vmax = fmaxf(vmax, fabsf(r)); }
    }
    if (y_amax) bn_amax_update(y_amax, vmax, sh);
}
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_apply(const float* __restrict__ x, const float* __restrict__ part,
                                                           const float* __restrict__ weight, const float* __restrict__ bias,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ y, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, BnGeom gm, float momentum, float eps,
                                                           int act, float slope, int nparts, long long* __restrict__ num_batches_tracked,
                                                           float* __restrict__ y_amax = nullptr)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_fwd_apply_body<false>(x, part, weight, bias, running_mean, running_var, y, save_mean, save_invstd, gm, momentum, eps, act, slope, nparts,
                             num_batches_tracked, y_amax, sh);
}

// One launch for both forward passes (round 5; the small tensors of a 2-sample step: 68 BatchNorm launches of its 288): every workgroup
// writes its chunk's triplet, counts itself on its CHANNEL's arrival counter and waits until all chunks of the channel have -- the whole
// grid is resident (the launcher sees to that: at most BN_COOP_MAX_WGS workgroups of 256 threads, no workgroup needs another to leave
// before it can start) --, then merges the triplets and streams its chunk as bn_fwd_apply does: the same arithmetic in the same order,
// the same bits.  Two counters per channel (arrived, left), zero before the launch and zero after it (the last workgroup to leave a
// channel's barrier clears both).  A wait that does not end within ~0.2 s gives up (wrong numbers instead of a hung GPU; it has never been seen to).
// (a channel's two counters have a 64-byte line to themselves: memory-side atomics on one line run one after the other, ~11 ns each, and
//  with sixteen channels to a line the polls of 256 workgroups stood in front of the arrivals; the first poll comes after the time a
//  chunk takes, the later ones ~1 us apart)
constexpr int BN_COOP_STRIDE = 16;       // ints between the counters of two channels
__device__ __forceinline__ void bn_channel_barrier(int* __restrict__ counters, int c, int chunks)
{
    if (threadIdx.x == 0) {
        int* cnt = counters + (int64_t)c * BN_COOP_STRIDE;
        int* left = cnt + 8;
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // this thread's agent-scope stores of the partial are done
        const int before = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before != chunks - 1)                                 // (the last arrival knows without asking)
            for (int spins = 0; spins < (1 << 21); ++spins) {
                __builtin_amdgcn_s_sleep(6);
                if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chunks) break;
            }
        if (__hip_atomic_fetch_add(left, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == chunks - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(left, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_coop(const float* __restrict__ x, float* __restrict__ part,
                                                          const float* __restrict__ weight, const float* __restrict__ bias,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ y, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd, BnGeom gm, float momentum, float eps,
                                                          int act, float slope, long long* __restrict__ num_batches_tracked,
                                                          float* __restrict__ y_amax, int* __restrict__ counters)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_fwd_partial_body<true>(x, part, gm, sh);
    bn_channel_barrier(counters, blockIdx.y, gm.chunks);
    bn_fwd_apply_body<true>(x, part, weight, bias, running_mean, running_var, y, save_mean, save_invstd, gm, momentum, eps, act, slope, gm.chunks,
                            num_batches_tracked, y_amax, sh);
}

// ---- backward pass 1: per-chunk sums of dz and dz * xhat, dz = dy * act'(pre-activation) ----------------------
// The activation mask is recomputed from x (pre = xhat * w + b): nothing but x, the two saved statistics and the
// affine parameters is kept from the forward.
template <bool COH>
__device__ __forceinline__ void bn_bwd_partial_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                                    const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                    float* __restrict__ part, const BnGeom& gm, int act, float slope, double* sh)
{
    const int c = blockIdx.y, chunk = blockIdx.x;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
    int64_t base, len;
    chunk_span(gm, chunk, c, base, len);
    const float* px = x + base;
    const float* pg = dy + base;
    float s = 0.f, q = 0.f;
    auto one = [&](float xv, float gv) __attribute__((always_inline)) {
        const float xh = (xv - mean) * invstd;
        const float dz = gv * act_grad(xh * w + b, act, slope);
        s += dz; q += dz * xh;
    };
    if ((base & 3) == 0) {        // 16-byte loads (a first version with scalar loads had a 0.095 ms floor on small tensors)
        const float4* x4 = reinterpret_cast<const float4*>(px);
        const float4* g4 = reinterpret_cast<const float4*>(pg);
        for (int64_t i = threadIdx.x; i < len / 4; i += BN_THREADS) {
            const float4 xv = x4[i], gv = g4[i];
            one(xv.x, gv.x); one(xv.y, gv.y); one(xv.z, gv.z); one(xv.w, gv.w);
        }
        for (int64_t i = (len / 4) * 4 + threadIdx.x; i < len; i += BN_THREADS) one(px[i], pg[i]);
    } else {
        for (int64_t i = threadIdx.x; i < len; i += BN_THREADS) one(px[i], pg[i]);
    }
    double ds = s, dq = q;
    block_sum2(ds, dq, sh);
    if (threadIdx.x == 0) {
        st_part<COH>(part + ((int64_t)c * gm.chunks + chunk) * 2 + 0, (float)ds);
        st_part<COH>(part + ((int64_t)c * gm.chunks + chunk) * 2 + 1, (float)dq);
    }
}
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ weight, const float* __restrict__ bias,
                                                             const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                             float* __restrict__ part, BnGeom gm, int act, float slope)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_bwd_partial_body<false>(dy, x, weight, bias, save_mean, save_invstd, part, gm, act, slope, sh);
}

// ---- backward pass 2: dx = w * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); dweight, dbias ---------------
template <bool COH>
__device__ __forceinline__ void bn_bwd_apply_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                  const float* __restrict__ part, const float* __restrict__ weight,
                                                  const float* __restrict__ bias, const float* __restrict__ save_mean,
                                                  const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                  float* __restrict__ dweight, float* __restrict__ dbias, const BnGeom& gm,
                                                  int act, float slope, int accumulate, float* __restrict__ dx_amax, double* sh)
{
    const int c = blockIdx.y, chunk = blockIdx.x;
    double sdz, sdzx;
    channel_totals<COH>(part, c, gm.chunks, sdz, sdzx, sh);
    if (chunk == 0 && threadIdx.x == 0) {       // accumulate: dweight / dbias are the parameters' .grad buffers (+=, stream order)
        if (dbias) dbias[c] = accumulate ? dbias[c] + (float)sdz : (float)sdz;
        if (dweight) dweight[c] = accumulate ? dweight[c] + (float)sdzx : (float)sdzx;
    }
    const double cnt = (double)gm.N * (double)gm.HW;
    const float m_dz = (float)(sdz / cnt), m_dzx = (float)(sdzx / cnt);
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
    const float k = w * invstd;
    int64_t base, len;
    chunk_span(gm, chunk, c, base, len);
    const float* px = x + base;
    const float* pg = dy + base;
    float* po = dx + base;
    float vmax = 0.f;
    auto one = [&](float xv, float gv) __attribute__((always_inline)) -> float {
        const float xh = (xv - mean) * invstd;
        const float dz = gv * act_grad(xh * w + b, act, slope);
        return k * (dz - m_dz - xh * m_dzx);
    };
    if ((base & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(px);
        const float4* g4 = reinterpret_cast<const float4*>(pg);
        float4* o4 = reinterpret_cast<float4*>(po);
        for (int64_t i = threadIdx.x; i < len / 4; i += BN_THREADS) {
            const float4 xv = x4[i], gv = g4[i];
            const float4 r = make_float4(one(xv.x, gv.x), one(xv.y, gv.y), one(xv.z, gv.z), one(xv.w, gv.w));
            o4[i] = r;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
        for (int64_t i = (len / 4) * 4 + threadIdx.x; i < len; i += BN_THREADS) { const float r = one(px[i], pg[i]); po[i] = r; vmax = fmaxf(vmax, fabsf(r)); }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += BN_THREADS) { const float r = one(px[i], pg[i]); po[i] = r; vmax = fmaxf(vmax, fabsf(r)); }
    }
    if (dx_amax) bn_amax_update(dx_amax, vmax, sh);
}
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ part, const float* __restrict__ weight,
                                                           const float* __restrict__ bias, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                           float* __restrict__ dweight, float* __restrict__ dbias, BnGeom gm,
                                                           int act, float slope, int accumulate, float* __restrict__ dx_amax = nullptr)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_bwd_apply_body<false>(dy, x, part, weight, bias, save_mean, save_invstd, dx, dweight, dbias, gm, act, slope, accumulate, dx_amax, sh);
}
// both backward passes in one launch (see bn_fwd_coop)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_coop(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ part, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                          float* __restrict__ dweight, float* __restrict__ dbias, BnGeom gm,
                                                          int act, float slope, int accumulate, float* __restrict__ dx_amax,
                                                          int* __restrict__ counters)
{
    __shared__ double sh[2 * BN_THREADS];
    bn_bwd_partial_body<true>(dy, x, weight, bias, save_mean, save_invstd, part, gm, act, slope, sh);
    bn_channel_barrier(counters, blockIdx.y, gm.chunks);
    bn_bwd_apply_body<true>(dy, x, part, weight, bias, save_mean, save_invstd, dx, dweight, dbias, gm, act, slope, accumulate, dx_amax, sh);
}

// ---- host launchers ----------------------------------------------------------------------------------------
static BnGeom geom(int N, int C, int64_t HW)
{
    BnGeom g;
    g.N = N; g.C = C; g.HW = HW;
    g.chunk_len = bn_chunk_len(N, C, HW);
    g.pieces = (int)((HW + g.chunk_len - 1) / g.chunk_len);
    g.chunks = N * g.pieces;
    return g;
}

int64_t bn_workspace_floats(int64_t N, int64_t C, int64_t HW)
{
    const int64_t len = bn_chunk_len(N, C, HW);
    const int64_t pieces = (HW + len - 1) / len;
    return 3 * C * N * pieces;         // forward: (count, mean, M2) per (channel, chunk); backward uses 2 of the 3
}

// ---- the one-launch forms: when, and their arrival counters ------------------------------------------------------------------------
// Counter regions of BN_COOP_COUNTERS ints out of ONE pool per device, allocated and zeroed at the first use outside a capture (an
// allocation is not a capturable operation) and left zero by every launch that uses them.  Who shares a region must run one after the
// other: an eager stream keeps one region, every stream CAPTURE gets its own (keyed by the capture's id: two graphs captured on the
// same pool stream may be replayed side by side).  nullptr (no pool yet while capturing, pool used up, an API call failed): two launches.
constexpr int BN_COOP_COUNTERS = 16384, BN_COOP_REGIONS = 128;      // 64 KB per region: 1024 channels of 16 ints
static int* bn_counters_for(hipStream_t s)
{
    struct Pool { int* base = nullptr; int used = 0; std::map<std::pair<int, unsigned long long>, int> region; };
    static std::mutex mu;
    static std::map<int, Pool> pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cap_id = 0;
    if (hipStreamGetCaptureInfo(s, &st, &cap_id) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const bool capturing = st != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(mu);
    Pool& pool = pools[dev];
    if (!pool.base) {
        if (capturing) return nullptr;
        int* p = nullptr;
        const size_t bytes = (size_t)BN_COOP_REGIONS * BN_COOP_COUNTERS * sizeof(int);
        if (hipMalloc(reinterpret_cast<void**>(&p), bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
        pool.base = p;
    }
    const auto key = std::make_pair(capturing ? 1 : 0, capturing ? cap_id : (unsigned long long)reinterpret_cast<uintptr_t>(s));
    auto it = pool.region.find(key);
    if (it == pool.region.end()) {
        if (pool.used >= BN_COOP_REGIONS) return nullptr;
        it = pool.region.emplace(key, pool.used++).first;
    }
    return pool.base + (size_t)it->second * BN_COOP_COUNTERS;
}
// workgroups of `kernel` the chip holds at once (every one of a one-launch grid must be resident: they wait for one another)
static int64_t bn_resident_wgs(const void* kernel)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BN_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return (int64_t)per_cu * prop.multiProcessorCount;
}
// OPT-IN (SSTEM_BN_ONE_LAUNCH=1, read at every launch).  Measured (profiles/r05/m_*): bit-identical, 288 -> 220 launches of the 2-sample
// fusion step -- and the same time, 3.39 against 3.36 ms replayed, 10.82 against 10.73 ms at 16 samples: the wait for the channel's
// last chunk (memory-side atomics, ~1 us to see an arrival) costs what the second launch cost.  Tensors above BN_COOP_MAX_ELEMS stream
// from HBM in either form and have nothing to gain from a shorter launch chain.
constexpr int64_t BN_COOP_MAX_ELEMS = (int64_t)1 << 24;
static int* bn_one_launch_counters(const void* kernel, const BnGeom& g, hipStream_t s)
{
    const char* env = getenv("SSTEM_BN_ONE_LAUNCH");
    if (!env || atoi(env) == 0) return nullptr;
    if ((int64_t)g.N * g.C * g.HW > BN_COOP_MAX_ELEMS || (int64_t)g.C * BN_COOP_STRIDE > BN_COOP_COUNTERS) return nullptr;
    static std::mutex mu;
    static std::map<const void*, int64_t> resident;
    int64_t cap;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = resident.find(kernel);
        if (it == resident.end()) it = resident.emplace(kernel, bn_resident_wgs(kernel)).first;
        cap = it->second;
    }
    if ((int64_t)g.chunks * g.C > cap * 3 / 4) return nullptr;
    return bn_counters_for(s);
}

hipError_t launch_bn_train_forward(const float* x, const float* weight, const float* bias, float* running_mean,
                                   float* running_var, float* y, float* save_mean, float* save_invstd, float* workspace,
                                   int N, int C, int64_t HW, float momentum, float eps, int act, float slope, hipStream_t s,
                                   const float* partials, int64_t n_partials, long long* num_batches_tracked, float* y_amax)
{
    // partials (nullable): [C][n_partials][3] (count, mean, M2) triplets the producing convolution wrote -- the statistics pass
    // over x is skipped
    const BnGeom g = geom(N, C, HW);
    if (g.chunks > 0x7fffffff / 3 || C > 65535 || n_partials > 0x7fffffff / 3) return hipErrorInvalidValue;
    const dim3 grid((unsigned)g.chunks, (unsigned)C);
    int nparts = g.chunks;
    if (partials) {
        nparts = (int)n_partials;
    } else if (int* counters = bn_one_launch_counters(reinterpret_cast<const void*>(bn_fwd_coop), g, s)) {
        hipLaunchKernelGGL(bn_fwd_coop, grid, dim3(BN_THREADS), 0, s, x, workspace, weight, bias, running_mean, running_var, y, save_mean,
                           save_invstd, g, momentum, eps, act, slope, num_batches_tracked, y_amax, counters);
        return hipGetLastError();
    } else {
        hipLaunchKernelGGL(bn_fwd_partial, grid, dim3(BN_THREADS), 0, s, x, workspace, g);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        partials = workspace;
    }
    hipLaunchKernelGGL(bn_fwd_apply, grid, dim3(BN_THREADS), 0, s, x, partials, weight, bias, running_mean, running_var, y,
                       save_mean, save_invstd, g, momentum, eps, act, slope, nparts, num_batches_tracked, y_amax);
    return hipGetLastError();
}

hipError_t launch_bn_train_backward(const float* dy, const float* x, const float* weight, const float* bias,
                                    const float* save_mean, const float* save_invstd, float* dx, float* dweight,
                                    float* dbias, float* workspace, int N, int C, int64_t HW, int act, float slope,
                                    hipStream_t s, int accumulate, float* dx_amax)
{
    const BnGeom g = geom(N, C, HW);
    if (g.chunks > 0x7fffffff / 2 || C > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)g.chunks, (unsigned)C);
    if (int* counters = bn_one_launch_counters(reinterpret_cast<const void*>(bn_bwd_coop), g, s)) {
        hipLaunchKernelGGL(bn_bwd_coop, grid, dim3(BN_THREADS), 0, s, dy, x, workspace, weight, bias, save_mean, save_invstd, dx, dweight, dbias,
                           g, act, slope, accumulate, dx_amax, counters);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_bwd_partial, grid, dim3(BN_THREADS), 0, s, dy, x, weight, bias, save_mean, save_invstd, workspace, g,
                       act, slope);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bn_bwd_apply, grid, dim3(BN_THREADS), 0, s, dy, x, workspace, weight, bias, save_mean, save_invstd, dx,
                       dweight, dbias, g, act, slope, accumulate, dx_amax);
    return hipGetLastError();
}

}  // namespace sstem
