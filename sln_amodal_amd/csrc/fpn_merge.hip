// FPN top-down merge (modal/modals.py:243-246): out = lateral + nearest-2x(top), NHWC fp32, as one pass
// (the reference runs F.upsample and an add: the upsampled map is written and read back), and its
// adjoint for the coarse input: gtop[n, y, x, c] = sum of the 2x2 fine-grid gradients above it.
#include "common.h"

__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float4 *__restrict__ lateral,
                                                             const float4 *__restrict__ top, int h, int w, int c4,
                                                             long total, float4 *__restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % c4);
        long t = e / c4;
        const int x = (int)(t % (2 * w));
        t /= 2 * w;
        const int y = (int)(t % (2 * h));
        const long n = t / (2 * h);
        const float4 a = lateral[e];
        const float4 b = top[((n * h + (y >> 1)) * w + (x >> 1)) * c4 + c];
        out[e] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float4 *__restrict__ g, int h, int w, int c4,
                                                         long total, float4 *__restrict__ gtop) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % c4);
        long t = e / c4;
        const int x = (int)(t % w);
        t /= w;
        const int y = (int)(t % h);
        const long n = t / h;
        const float4 *r0 = g + ((n * 2 * h + 2 * y) * 2 * w + 2 * x) * c4 + c;
        const float4 *r1 = r0 + (long)2 * w * c4;
        const float4 a = r0[0], b = r0[c4], cc = r1[0], d = r1[c4];
        gtop[e] = make_float4((a.x + b.x) + (cc.x + d.x), (a.y + b.y) + (cc.y + d.y), (a.z + b.z) + (cc.z + d.z),
                              (a.w + b.w) + (cc.w + d.w));
    }
}

static inline int fm_grid(long total) {
    long b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int sln_upsample2x_add_f32(const float *lateral, const float *top, int N, int h, int w, int C,
                                      float *out, sln_stream_t stream) {
    if (N < 0 || h < 1 || w < 1 || C < 4 || (C & 3)) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!lateral || !top || !out) return SLN_ERR_INVALID_ARG;
    sln_enter();
    const long total = (long)N * 2 * h * 2 * w * (C / 4);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(fm_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)lateral, (const float4 *)top, h, w, C / 4, total, (float4 *)out);
    return sln_launch_status();
}

extern "C" int sln_sumpool2x2_f32(const float *g, int N, int h, int w, int C, float *gtop, sln_stream_t stream) {
    if (N < 0 || h < 1 || w < 1 || C < 4 || (C & 3)) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!g || !gtop) return SLN_ERR_INVALID_ARG;
    sln_enter();
    const long total = (long)N * h * w * (C / 4);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(fm_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float4 *)g,
                       h, w, C / 4, total, (float4 *)gtop);
    return sln_launch_status();
}
