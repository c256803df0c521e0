// Tail of the global layer module (reference model.py:537-541 with modal/msc_deeplab.py:42-48): the logits of the
// coarser scales are resized bilinearly (align_corners = False) to the scale-1 grid, the element-wise maximum over
// the scales is taken, then softmax over the classes, argmax, and [probs | argmax / 255] is written -- one pass,
// one wave per output pixel (lanes = classes), instead of two resizes, two maxima, softmax, argmax, a division, a
// concatenation and a layout copy.  The arithmetic per element is the reference's (ATen's upsample_bilinear2d:
// source index max(in/out * (dst + 0.5) - 0.5, 0), the lerp as h0*(w0*a + w1*b) + h1*(w0*c + w1*d); softmax as
// exp(v - max) / sum); compiled without contraction / fast math.
#include "common.h"

#define GLM_MAXLV 3
struct GlmPyr {
    const float *ptr[GLM_MAXLV];
    int h[GLM_MAXLV], w[GLM_MAXLV];
    long ps[GLM_MAXLV];        // pixel stride in floats
    int n;
};

__device__ __forceinline__ void glm_axis(int in, int out, int dst, int &i0, int &i1, float &l0, float &l1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

template <int NPL>     // classes per lane: C <= 64 * NPL
__global__ __launch_bounds__(256) void glm_tail_kernel(const float *__restrict__ logits, long lps, GlmPyr pyr, int B,
                                                       int C, int H, int W, float *__restrict__ probs,
                                                       long long *__restrict__ label) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long)B * H * W) return;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    float v[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < C ? logits[pix * lps + c] : -INFINITY;
    }
    for (int l = 0; l < pyr.n; ++l) {
        int y0, y1, x0, x1;
        float hl0, hl1, wl0, wl1;
        glm_axis(pyr.h[l], H, y, y0, y1, hl0, hl1);
        glm_axis(pyr.w[l], W, x, x0, x1, wl0, wl1);
        const float *base = pyr.ptr[l] + n * pyr.h[l] * pyr.w[l] * pyr.ps[l];
        const float *p00 = base + ((long)y0 * pyr.w[l] + x0) * pyr.ps[l], *p01 = base + ((long)y0 * pyr.w[l] + x1) * pyr.ps[l];
        const float *p10 = base + ((long)y1 * pyr.w[l] + x0) * pyr.ps[l], *p11 = base + ((long)y1 * pyr.w[l] + x1) * pyr.ps[l];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const float r = hl0 * (wl0 * p00[c] + wl1 * p01[c]) + hl1 * (wl0 * p10[c] + wl1 * p11[c]);
                v[k] = fmaxf(v[k], r);      // (torch.max propagates NaN, fmaxf does not: logits are finite)
            }
        }
    }
    float m = v[0];
#pragma unroll
    for (int k = 1; k < NPL; ++k) m = fmaxf(m, v[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e[NPL], s = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        e[k] = (lane + 64 * k) < C ? expf(v[k] - m) : 0.f;
        s += e[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float best = -1.f;
    int arg = 0x7fffffff;
    float *out = probs + pix * (long)(C + 1);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
            const float p = e[k] / s;
            out[c] = p;
            if (p > best) { best = p; arg = c; }      // (ascending c: the first maximum of this lane)
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {               // the maximum; among equals the lowest class
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        out[C] = (float)arg / 255.0f;
        label[pix] = arg;
    }
}

extern "C" int sln_msc_softmax_tail_f32(const float *logits, int64_t logits_pixel_stride, const float *const *pyramid,
                                        const int32_t *pyramid_hw, const int64_t *pyramid_pixel_stride, int n_pyramid,
                                        int B, int C, int H, int W, float *probs, int64_t *label,
                                        sln_stream_t stream) {
    sln_enter();
    if (B < 0 || C < 1 || C > 256 || H < 1 || W < 1 || n_pyramid < 0 || n_pyramid > GLM_MAXLV ||
        logits_pixel_stride < C)
        return SLN_ERR_INVALID_ARG;
    if (B == 0) return SLN_OK;
    if (!logits || !probs || !label || (n_pyramid > 0 && (!pyramid || !pyramid_hw || !pyramid_pixel_stride)))
        return SLN_ERR_INVALID_ARG;
    GlmPyr p;
    p.n = n_pyramid;
    for (int i = 0; i < n_pyramid; ++i) {
        p.ptr[i] = pyramid[i]; p.h[i] = pyramid_hw[2 * i]; p.w[i] = pyramid_hw[2 * i + 1];
        p.ps[i] = pyramid_pixel_stride[i];
        if (!p.ptr[i] || p.h[i] < 1 || p.w[i] < 1 || p.ps[i] < C) return SLN_ERR_INVALID_ARG;
    }
    const long npix = (long)B * H * W;
    const dim3 g((unsigned)((npix + 3) / 4)), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (C <= 64) hipLaunchKernelGGL(glm_tail_kernel<1>, g, b, 0, st, logits, (long)logits_pixel_stride, p, B, C, H, W, probs, (long long *)label);
    else if (C <= 128) hipLaunchKernelGGL(glm_tail_kernel<2>, g, b, 0, st, logits, (long)logits_pixel_stride, p, B, C, H, W, probs, (long long *)label);
    else if (C <= 192) hipLaunchKernelGGL(glm_tail_kernel<3>, g, b, 0, st, logits, (long)logits_pixel_stride, p, B, C, H, W, probs, (long long *)label);
    else hipLaunchKernelGGL(glm_tail_kernel<4>, g, b, 0, st, logits, (long)logits_pixel_stride, p, B, C, H, W, probs, (long long *)label);
    return sln_launch_status();
}
