// Grouped 3x3 convolution, forward only: BASELINE.json configs[4] (ResNeXt-101 32 x 4d; the reference holds the
// backbone as dead code, modal/resnext.py:31-41: GroupBottleneck.conv2 = Conv2d(planes, planes, 3, stride,
// padding 1, groups 32, no bias) at 4 / 8 / 16 / 32 channels per group).  Per group the reduction is
// K = 9 * CG <= 288 with CG outputs: too narrow for the 256- and 128-wide MFMA tiles of conv.hip, and a
// block-diagonal dense weight would spend 8-32x the FLOPs, so this is a direct fp32 kernel: a block = one group x
// 256 output pixels, the group's weights in LDS as [tap][ci][co] (every lane of a wave reads the same address: a
// broadcast), a thread = one output pixel with its CG accumulators in registers, 16-B loads of the CG input
// channels of each tap (NHWC: a group's channels are contiguous).  The BN affine and the ReLU are applied before
// the store.  The grouped convolutions are ~4 % of the network's FLOPs (that is the point of the grouping).
#include "common.h"

template <int CG>
__global__ __launch_bounds__(256) void grouped_conv3x3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int relu, int N, int H,
                                                              int W, int C, int stride, int OH, int OW,
                                                              float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float s_w[9][CG][CG];      // [tap][ci][co]
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 9 * CG * CG; i += 256) {
        const int co = i % CG, ci = (i / CG) % CG, tap = i / (CG * CG);
        s_w[tap][ci][co] = w[((size_t)(g * CG + co) * CG + ci) * 9 + tap];          // torch layout [Cout][CG][3][3]
    }
    __syncthreads();
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)N * OH * OW) return;
    const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH);
    const long n = pix / ((long)OW * OH);
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int ih = oh * stride - 1 + tap / 3, iw = ow * stride - 1 + tap % 3;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;   // zero padding
        const float4 *src = (const float4 *)(x + ((n * H + ih) * (long)W + iw) * C + g * CG);
        float xin[CG];
#pragma unroll
        for (int q = 0; q < CG / 4; ++q) {
            const float4 v = src[q];
            xin[4 * q] = v.x; xin[4 * q + 1] = v.y; xin[4 * q + 2] = v.z; xin[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int q = 0; q < CG / 4; ++q) {
                const float4 wv = *(const float4 *)&s_w[tap][ci][4 * q];
                acc[4 * q] += xin[ci] * wv.x; acc[4 * q + 1] += xin[ci] * wv.y;
                acc[4 * q + 2] += xin[ci] * wv.z; acc[4 * q + 3] += xin[ci] * wv.w;
            }
    }
    float4 *dst = (float4 *)(y + pix * C + g * CG);
#pragma unroll
    for (int q = 0; q < CG / 4; ++q) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = g * CG + 4 * q + e;
            float v = acc[4 * q + e];
            if (scale) v = v * scale[c];
            if (shift) v += shift[c];
            o[e] = relu ? fmaxf(v, 0.f) : v;
        }
        dst[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int sln_grouped_conv3x3_f32(const float *x, int N, int H, int W, int C, int groups, const float *w,
                                       int stride, const float *scale, const float *shift, int relu, float *y,
                                       sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 1 || groups < 1 || C % groups || stride < 1 || stride > 2)
        return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    if (N == 0) return SLN_OK;
    if (!x || !w || !y || ((((size_t)x) | ((size_t)y)) & 15)) return SLN_ERR_INVALID_ARG;
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    const long npix = (long)N * OH * OW;
    if (npix > 2147483647L * 128) return SLN_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((npix + 255) / 256), (unsigned)groups), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SLN_GC(CGV) hipLaunchKernelGGL(grouped_conv3x3_kernel<CGV>, grid, block, 0, st, x, w, scale, shift, relu, N, H, W, C, stride, OH, OW, y)
    if (cg == 4) SLN_GC(4); else if (cg == 8) SLN_GC(8); else if (cg == 16) SLN_GC(16); else SLN_GC(32);
#undef SLN_GC
    return sln_launch_status();
}
