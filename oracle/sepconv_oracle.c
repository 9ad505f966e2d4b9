/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped operator (sstem-restoration_amd/libs/sepconv)
 * never calls it and raises when the HIP library is missing.
 *
 * What this is: a plain-C CPU restatement of the three CUDA kernels of the
 * reference's separable-convolution op, following them literally (same index
 * decode, same loop order, same float accumulator, one "thread" per element):
 *
 *   sepconv_oracle_forward       <- libs/sepconv/src/SeparableConvolution_kernel.cu:25-52
 *   sepconv_oracle_grad_vertical <- libs/sepconv/src/SeparableConvolution_kernel.cu:77-112
 *   sepconv_oracle_grad_horizontal <- ...kernel.cu:115-150
 *   sepconv_oracle_backward      <- ...kernel.cu:152-206 (gradInput is never written)
 *
 * Pinning status: the reference holds NO golden vectors, fixtures or tests for
 * this op, has no CPU implementation of it (SeparableConvolution.py:47-48
 * raises NotImplementedError) and its CUDA/THC sources cannot be compiled in
 * this image.  => "parity unpinned" by reference outputs.  The restatement is
 * pinned instead by (i) analytic known-answer tests (one-hot kernels give an
 * exact shifted crop, 1/51 kernels give a box filter), (ii) an independent
 * second restatement (oracle/sepconv_numpy.py, unfold/einsum in float64) and
 * (iii) the reference's own gradcheck shape (model_interp.py:109-119).
 *
 * Arithmetic notes (kept identical to the reference):
 *   - `float` running sum, product evaluated left to right as
 *     (in * V) * H then added (kernel.cu:47).  Build with -ffp-contract=off so
 *     the host compiler does not fuse what nvcc may or may not have fused;
 *     tests use a tolerance for GPU-vs-oracle anyway and exact equality only
 *     for the one-hot indexing KATs where every ordering gives the same bits.
 *   - the reference decodes the flat index in 32-bit int; here int64_t (the
 *     decode is identical for every size the reference could address).
 *   - the gradient kernels hard-code three channels (kernel.cu:100-108): the
 *     oracle does the same and REJECTS C != 3 instead of reading out of bounds.
 */
#include <stdint.h>
#include <stddef.h>

#define FILTER_LENGTH 51

#ifdef _OPENMP
#include <omp.h>
#endif

/* out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] * V[b,fy,y,x] * H[b,fx,y,x]
 * input  [B,C,H+50,W+50], vertical/horizontal [B,51,H,W], output [B,C,H,W],
 * all contiguous NCHW fp32.  Returns 0 on success. */
int sepconv_oracle_forward(const float* input, const float* vertical,
                           const float* horizontal, float* output,
                           int64_t B, int64_t C, int64_t H, int64_t W)
{
    if (B < 0 || C < 0 || H < 0 || W < 0) return 1;
    const int64_t Hin = H + FILTER_LENGTH - 1, Win = W + FILTER_LENGTH - 1;
    const int64_t n = B * C * H * W;
    const int64_t plane = H * W;
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n; ++idx) {
        /* index decode, kernel.cu:40-43 */
        const int64_t b = (idx / W / H / C) % B;
        const int64_t c = (idx / W / H) % C;
        const int64_t y = (idx / W) % H;
        const int64_t x = idx % W;
        const float* in_bc = input + ((b * C + c) * Hin) * Win;
        const float* v_b = vertical + b * FILTER_LENGTH * plane + y * W + x;
        const float* h_b = horizontal + b * FILTER_LENGTH * plane + y * W + x;
        float acc = 0.0f;
        /* fy outer, fx inner, kernel.cu:45-49 */
        for (int fy = 0; fy < FILTER_LENGTH; ++fy) {
            const float vv = v_b[(int64_t)fy * plane];
            const float* row = in_bc + (y + fy) * Win + x;
            for (int fx = 0; fx < FILTER_LENGTH; ++fx) {
                acc += row[fx] * vv * h_b[(int64_t)fx * plane];
            }
        }
        output[idx] = acc;
    }
    return 0;
}

/* gV[b,fy,y,x] = sum_fx sum_{c<3} g[b,c,y,x] * in[b,c,y+fy,x+fx] * H[b,fx,y,x] */
int sepconv_oracle_grad_vertical(const float* grad_out, const float* input,
                                 const float* horizontal, float* grad_vertical,
                                 int64_t B, int64_t C, int64_t H, int64_t W)
{
    if (C != 3) return 2; /* kernel.cu:100-108 hard-codes channels 0,1,2 */
    const int64_t Hin = H + FILTER_LENGTH - 1, Win = W + FILTER_LENGTH - 1;
    const int64_t plane = H * W;
    const int64_t n = B * FILTER_LENGTH * plane;
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n; ++idx) {
        const int64_t b = (idx / W / H / FILTER_LENGTH) % B;
        const int64_t fy = (idx / W / H) % FILTER_LENGTH;
        const int64_t y = (idx / W) % H;
        const int64_t x = idx % W;
        const float* g_b = grad_out + (b * 3) * plane + y * W + x;
        const float* in_b = input + (b * 3) * Hin * Win + (y + fy) * Win + x;
        const float* h_b = horizontal + b * FILTER_LENGTH * plane + y * W + x;
        float acc = 0.0f;
        for (int fx = 0; fx < FILTER_LENGTH; ++fx) {
            const float hh = h_b[(int64_t)fx * plane];
            acc += g_b[0] * in_b[fx] * hh
                 + g_b[plane] * in_b[Hin * Win + fx] * hh
                 + g_b[2 * plane] * in_b[2 * Hin * Win + fx] * hh;
        }
        grad_vertical[idx] = acc;
    }
    return 0;
}

/* gH[b,fx,y,x] = sum_fy sum_{c<3} g[b,c,y,x] * in[b,c,y+fy,x+fx] * V[b,fy,y,x] */
int sepconv_oracle_grad_horizontal(const float* grad_out, const float* input,
                                   const float* vertical, float* grad_horizontal,
                                   int64_t B, int64_t C, int64_t H, int64_t W)
{
    if (C != 3) return 2; /* kernel.cu:138-146 */
    const int64_t Hin = H + FILTER_LENGTH - 1, Win = W + FILTER_LENGTH - 1;
    const int64_t plane = H * W;
    const int64_t n = B * FILTER_LENGTH * plane;
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n; ++idx) {
        const int64_t b = (idx / W / H / FILTER_LENGTH) % B;
        const int64_t fx = (idx / W / H) % FILTER_LENGTH;
        const int64_t y = (idx / W) % H;
        const int64_t x = idx % W;
        const float* g_b = grad_out + (b * 3) * plane + y * W + x;
        const float* in_b = input + (b * 3) * Hin * Win + y * Win + x + fx;
        const float* v_b = vertical + b * FILTER_LENGTH * plane + y * W + x;
        float acc = 0.0f;
        for (int fy = 0; fy < FILTER_LENGTH; ++fy) {
            const float vv = v_b[(int64_t)fy * plane];
            acc += g_b[0] * in_b[(int64_t)fy * Win] * vv
                 + g_b[plane] * in_b[Hin * Win + (int64_t)fy * Win] * vv
                 + g_b[2 * plane] * in_b[2 * Hin * Win + (int64_t)fy * Win] * vv;
        }
        grad_horizontal[idx] = acc;
    }
    return 0;
}

/* kernel.cu:152-206: gradVertical launch, then gradHorizontal launch.
 * grad_input is accepted and NEVER written (it stays whatever the caller
 * put there -- the Python side zero-fills it, SeparableConvolution.py:60). */
int sepconv_oracle_backward(const float* grad_out, const float* input,
                            const float* vertical, const float* horizontal,
                            float* grad_input, float* grad_vertical,
                            float* grad_horizontal,
                            int64_t B, int64_t C, int64_t H, int64_t W)
{
    (void)grad_input;
    int rc = sepconv_oracle_grad_vertical(grad_out, input, horizontal,
                                          grad_vertical, B, C, H, W);
    if (rc) return rc;
    return sepconv_oracle_grad_horizontal(grad_out, input, vertical,
                                          grad_horizontal, B, C, H, W);
}

int sepconv_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
