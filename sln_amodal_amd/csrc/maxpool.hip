// Max-pooling of the two stems (modal/modals.py:316-317: SamePad2d + MaxPool2d(3, 2); modal/resnet_deeplab.py:
// MaxPool2d(3, 2, 1, ceil_mode=True)), NHWC fp32, C % 4 == 0.  Windows are clipped to the map (implicit
// -inf padding / ceil mode: for the backbone's post-ReLU input a clipped window equals the reference's
// zero-padded one).  The arg-max follows torch: taps visited kh-major, a tap wins if it is greater than
// the running maximum or NaN, so among equal values (the many exact zeros behind a ReLU) the first one
// keeps the gradient.  Forward stores the winning tap (kh*KW + kw, relative to the unclipped window) as
// one byte per output; backward is a gather over the <= 4 windows that contain an input pixel: every
// input gradient is written exactly once -- no memset, no atomics.
#include "common.h"

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float4 *__restrict__ x, int H, int W, int c4,
                                                          int K, int S, int pt, int pl, int OH, int OW,
                                                          long total, float4 *__restrict__ y,
                                                          uchar4 *__restrict__ arg) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % c4);
        long t = e / c4;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const long n = t / OH;
        const int h0 = oh * S - pt, w0 = ow * S - pl;
        const float ninf = -__builtin_inff();
        float m[4] = {ninf, ninf, ninf, ninf};
        int a[4] = {-1, -1, -1, -1};
        for (int kh = 0; kh < K; ++kh) {
            const int ih = h0 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < K; ++kw) {
                const int iw = w0 + kw;
                if (iw < 0 || iw >= W) continue;
                const float4 v4 = x[((n * H + ih) * W + iw) * c4 + c];
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                const int tap = kh * K + kw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (a[j] < 0 || v[j] > m[j] || v[j] != v[j]) {   // first valid tap, greater, or NaN
                        m[j] = v[j];
                        a[j] = tap;
                    }
                }
            }
        }
        y[e] = make_float4(m[0], m[1], m[2], m[3]);
        arg[e] = make_uchar4((unsigned char)a[0], (unsigned char)a[1], (unsigned char)a[2], (unsigned char)a[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float4 *__restrict__ g,
                                                          const uchar4 *__restrict__ arg, int H, int W, int c4,
                                                          int K, int S, int pt, int pl, int OH, int OW,
                                                          long total, float4 *__restrict__ gx) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % c4);
        long t = e / c4;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H);
        const long n = t / H;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // windows containing (ih, iw): oh*S - pt <= ih <= oh*S - pt + K - 1
        const int oh_lo = max(0, (ih + pt - K + S) / S), oh_hi = min(OH - 1, (ih + pt) / S);
        const int ow_lo = max(0, (iw + pl - K + S) / S), ow_hi = min(OW - 1, (iw + pl) / S);
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const int kh = ih - (oh * S - pt);
            if (kh < 0 || kh >= K) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int kw = iw - (ow * S - pl);
                if (kw < 0 || kw >= K) continue;
                const long o = ((n * OH + oh) * OW + ow) * c4 + c;
                const uchar4 a = arg[o];
                const float4 gv = g[o];
                const unsigned tap = (unsigned)(kh * K + kw);
                if (a.x == tap) acc[0] += gv.x;
                if (a.y == tap) acc[1] += gv.y;
                if (a.z == tap) acc[2] += gv.z;
                if (a.w == tap) acc[3] += gv.w;
            }
        }
        gx[e] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

static inline int mp_grid(long total) {
    long b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    return (int)(b < 1 ? 1 : b);
}

static inline bool mp_args_ok(int N, int H, int W, int C, int K, int S, int pt, int pl, int OH, int OW) {
    return N >= 0 && H >= 1 && W >= 1 && C >= 4 && (C & 3) == 0 && K >= 1 && K <= 15 && S >= 1 && pt >= 0 &&
           pl >= 0 && pt < K && pl < K && OH >= 1 && OW >= 1 && (OH - 1) * S - pt < H && (OW - 1) * S - pl < W;
}

extern "C" int sln_maxpool_fwd_f32(const float *x, int N, int H, int W, int C, int K, int S, int pad_top,
                                   int pad_left, int OH, int OW, float *y, uint8_t *argmax, sln_stream_t stream) {
    if (!mp_args_ok(N, H, W, C, K, S, pad_top, pad_left, OH, OW)) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!x || !y || !argmax) return SLN_ERR_INVALID_ARG;
    sln_enter();
    const long total = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(mp_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x,
                       H, W, C / 4, K, S, pad_top, pad_left, OH, OW, total, (float4 *)y, (uchar4 *)argmax);
    return sln_launch_status();
}

extern "C" int sln_maxpool_bwd_f32(const float *g, const uint8_t *argmax, int N, int H, int W, int C, int K, int S,
                                   int pad_top, int pad_left, int OH, int OW, float *gx, sln_stream_t stream) {
    if (!mp_args_ok(N, H, W, C, K, S, pad_top, pad_left, OH, OW)) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!g || !argmax || !gx) return SLN_ERR_INVALID_ARG;
    sln_enter();
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(mp_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float4 *)g,
                       (const uchar4 *)argmax, H, W, C / 4, K, S, pad_top, pad_left, OH, OW, total, (float4 *)gx);
    return sln_launch_status();
}
