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

// MODE 0: forward.  MODE 1: data gradient -- a thread = one INPUT pixel, gx[ci] = sum over the taps whose output
// position exists (stride 2: every other one) and the group's co of g[co] * w[co][ci][tap], where g = gy * scale[c] on
// the units that were active in the forward pass (y > 0) when `y` is given: the ReLU mask and the BN scale of
// conv -> BN -> ReLU are applied while the gradient is read.
template <int CG, int MODE>
__global__ __launch_bounds__(256) void grouped_conv3x3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int relu, int N, int H,
                                                              int W, int C, int stride, int OH, int OW,
                                                              float *__restrict__ y, const float *__restrict__ ymask) {
    __shared__ __attribute__((aligned(16))) float s_w[9][CG][CG];      // [tap][reduced channel][produced channel]
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 9 * CG * CG; i += 256) {
        const int a = i % CG, b = (i / CG) % CG, tap = i / (CG * CG);      // torch layout [Cout][CG][3][3]
        s_w[tap][b][a] = MODE == 0 ? w[((size_t)(g * CG + a) * CG + b) * 9 + tap]       // [ci = b][co = a]
                                   : w[((size_t)(g * CG + b) * CG + a) * 9 + tap];      // [co = b][ci = a]
    }
    __syncthreads();
    // forward: x [N,H,W,C] -> y [N,OH,OW,C]; data gradient: "x" = gy [N,OH,OW,C] -> "y" = gx [N,H,W,C]
    const int PH = MODE == 0 ? OH : H, PW = MODE == 0 ? OW : W;          // the grid of produced pixels
    const int QH = MODE == 0 ? H : OH, QW = MODE == 0 ? W : OW;          // the grid that is read
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)N * PH * PW) return;
    const int pw = (int)(pix % PW), ph = (int)((pix / PW) % PH);
    const long n = pix / ((long)PW * PH);
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        int qh, qw;
        if (MODE == 0) {
            qh = ph * stride - 1 + tap / 3; qw = pw * stride - 1 + tap % 3;
        } else {
            const int th = ph + 1 - tap / 3, tw = pw + 1 - tap % 3;       // = oh * stride, ow * stride
            if (th < 0 || tw < 0 || th % stride || tw % stride) continue;
            qh = th / stride; qw = tw / stride;
        }
        if ((unsigned)qh >= (unsigned)QH || (unsigned)qw >= (unsigned)QW) continue;   // zero padding / no such output
        const long q = ((n * QH + qh) * (long)QW + qw) * C + g * CG;
        const float4 *src = (const float4 *)(x + q);
        float xin[CG];
#pragma unroll
        for (int k = 0; k < CG / 4; ++k) {
            const float4 v = src[k];
            xin[4 * k] = v.x; xin[4 * k + 1] = v.y; xin[4 * k + 2] = v.z; xin[4 * k + 3] = v.w;
        }
        if (MODE == 1 && ymask) {
            const float4 *ym = (const float4 *)(ymask + q);
#pragma unroll
            for (int k = 0; k < CG / 4; ++k) {
                const float4 m = ym[k];
                const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xin[4 * k + e] = mm[e] > 0.f ? xin[4 * k + e] * (scale ? scale[g * CG + 4 * k + e] : 1.f) : 0.f;
            }
        }
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int k = 0; k < CG / 4; ++k) {
                const float4 wv = *(const float4 *)&s_w[tap][ci][4 * k];
                acc[4 * k] += xin[ci] * wv.x; acc[4 * k + 1] += xin[ci] * wv.y;
                acc[4 * k + 2] += xin[ci] * wv.z; acc[4 * k + 3] += xin[ci] * wv.w;
            }
    }
    float4 *dst = (float4 *)(y + pix * C + g * CG);
#pragma unroll
    for (int k = 0; k < CG / 4; ++k) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = g * CG + 4 * k + e;
            float v = acc[4 * k + e];
            if (MODE == 0) {
                if (scale) v = v * scale[c];
                if (shift) v += shift[c];
                if (relu) v = fmaxf(v, 0.f);
            }
            o[e] = v;
        }
        dst[k] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Weight gradient: gw[co][ci][tap] = sum over output pixels of g[p][co] * x[p @ tap][ci] (g as in the data gradient).
// A block = one group x one range of output pixels; tiles of 32 pixels are staged in LDS (g [32][CG] and the nine
// shifted input rows [32][9][CG], zeros outside the image), a thread owns the (tap, co, ci) products
// t, t + 256, ... of the group's 9 * CG * CG and adds over the tile's pixels.  Partial sums per range go to
// `partial` [ranges][C * CG * 9] in the weight's own order; wgrad_reduce-style ordered sum afterwards (same bits
// on every run).
#define GW_TILE 32
template <int CG>
__global__ __launch_bounds__(256) void grouped_wgrad3x3_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                               const float *__restrict__ ymask,
                                                               const float *__restrict__ scale, int N, int H, int W,
                                                               int C, int stride, int OH, int OW, long pix_per_range,
                                                               float *__restrict__ partial) {
    constexpr int NPROD = 9 * CG * CG, NACC = CG >= 16 ? 1 : (NPROD + 255) / 256;
    constexpr int NP = CG * CG / 256;                // CG >= 16: (co, ci) pairs per thread, all nine taps each
    __shared__ float s_g[GW_TILE][CG];
    __shared__ float s_x[GW_TILE][9][CG + 1];        // (+1: the nine taps of a pixel on different banks)
    const int g = blockIdx.y, t = threadIdx.x;
    const long total = (long)N * OH * OW;
    const long p0 = (long)blockIdx.x * pix_per_range, p1 = min(total, p0 + pix_per_range);
    float acc[NACC];
    int e_tap[NACC], e_co[NACC], e_ci[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        acc[k] = 0.f;
        const int e = min(t + 256 * k, NPROD - 1);            // weight order: ((co * CG + ci) * 9 + tap)
        e_tap[k] = e % 9; e_ci[k] = (e / 9) % CG; e_co[k] = e / (9 * CG);
    }
    // CG >= 16: a thread owns the pairs (co0 + (256 / CG) k, ci), k < NP, with all nine taps: per pixel nine reads of
    // the shifted inputs and NP of the gradient feed 9 NP multiply-adds (the product-per-thread layout above needs
    // two reads per multiply-add and is LDS-bound)
    const int p_ci = t % CG, p_co0 = t / CG;
    float pacc[CG >= 16 ? NP : 1][9];
#pragma unroll
    for (int k = 0; k < (CG >= 16 ? NP : 1); ++k)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) pacc[k][tap] = 0.f;
    for (long base = p0; base < p1; base += GW_TILE) {
        __syncthreads();
        for (int i = t; i < GW_TILE * CG; i += 256) {
            const int c = i % CG, r = i / CG;
            const long p = base + r;
            float v = 0.f;
            if (p < p1) {
                const long q = p * C + g * CG + c;
                v = gy[q];
                if (ymask) v = ymask[q] > 0.f ? v * (scale ? scale[g * CG + c] : 1.f) : 0.f;
            }
            s_g[r][c] = v;
        }
        for (int i = t; i < GW_TILE * 9 * CG; i += 256) {
            const int c = i % CG, tap = (i / CG) % 9, r = i / (9 * CG);
            const long p = base + r;
            float v = 0.f;
            if (p < p1) {
                const int ow = (int)(p % OW), oh = (int)((p / OW) % OH);
                const long n = p / ((long)OW * OH);
                const int ih = oh * stride - 1 + tap / 3, iw = ow * stride - 1 + tap % 3;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = x[((n * H + ih) * (long)W + iw) * C + g * CG + c];
            }
            s_x[r][tap][c] = v;
        }
        __syncthreads();
        if (CG >= 16) {
#pragma unroll 2
            for (int r = 0; r < GW_TILE; ++r) {
                float xv[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) xv[tap] = s_x[r][tap][p_ci];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const float gv = s_g[r][p_co0 + (256 / CG) * k];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) pacc[k][tap] += gv * xv[tap];
                }
            }
        } else {
#pragma unroll 4
            for (int r = 0; r < GW_TILE; ++r)
#pragma unroll
                for (int k = 0; k < NACC; ++k) acc[k] += s_g[r][e_co[k]] * s_x[r][e_tap[k]][e_ci[k]];
        }
    }
    float *out = partial + (size_t)blockIdx.x * C * CG * 9 + (size_t)g * NPROD;     // group g's CG x CG x 9 slice
    if (CG >= 16) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                out[((size_t)(p_co0 + (256 / CG) * k) * CG + p_ci) * 9 + tap] = pacc[k][tap];
    } else {
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            if (t + 256 * k < NPROD) out[t + 256 * k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void grouped_wgrad_reduce_kernel(const float *__restrict__ partial, int ranges, long n,
                                                                   float *__restrict__ gw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int r = 0; r < ranges; ++r) a += partial[(size_t)r * n + i];      // fixed order
    gw[i] = a;
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
#define SLN_GC(CGV) hipLaunchKernelGGL((grouped_conv3x3_kernel<CGV, 0>), grid, block, 0, st, x, w, scale, shift, relu, N, H, W, C, stride, OH, OW, y, (const float *)nullptr)
    if (cg == 4) SLN_GC(4); else if (cg == 8) SLN_GC(8); else if (cg == 16) SLN_GC(16); else SLN_GC(32);
#undef SLN_GC
    return sln_launch_status();
}

// Data gradient of the layer above: gy [N,OH,OW,C] (the gradient w.r.t. the layer's OUTPUT) -> gx [N,H,W,C].  With
// y_out (the layer's forward output, post-ReLU) the gradient is first taken through the ReLU and the BN scale:
// g = gy * scale[c] where y_out > 0, else 0 (scale may be NULL: 1).
extern "C" int sln_grouped_conv3x3_dgrad_f32(const float *gy, const float *y_out, const float *scale, int N, int H,
                                             int W, int C, int groups, const float *w, int stride, float *gx,
                                             sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 1 || groups < 1 || C % groups || stride < 1 || stride > 2)
        return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    if (N == 0) return SLN_OK;
    if (!gy || !w || !gx || ((((size_t)gy) | ((size_t)gx) | ((size_t)y_out)) & 15)) return SLN_ERR_INVALID_ARG;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const long npix = (long)N * H * W;
    const dim3 grid((unsigned)((npix + 255) / 256), (unsigned)groups), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SLN_GD(CGV) hipLaunchKernelGGL((grouped_conv3x3_kernel<CGV, 1>), grid, block, 0, st, gy, w, scale, (const float *)nullptr, 0, N, H, W, C, stride, OH, OW, gx, y_out)
    if (cg == 4) SLN_GD(4); else if (cg == 8) SLN_GD(8); else if (cg == 16) SLN_GD(16); else SLN_GD(32);
#undef SLN_GD
    return sln_launch_status();
}

// Weight gradient gw [C][C/groups][3][3] (the parameter's order), same g as above.  workspace: at least
// sln_grouped_conv3x3_wgrad_workspace_bytes() bytes; two launches, no atomics, bit-reproducible.
static inline int gw_ranges(long npix) {
    // >= 256 pixels (eight LDS tiles) per range, at most 128 ranges per group: with 32 groups that is up to 4096
    // blocks.  (The first version asked for 4096 pixels per range: a 21 x 21 map of 8 images ran as 32 blocks on
    // 256 CUs, 1 ms per launch, 39 % of a ResNeXt train step.)
    long r = (npix + 255) / 256;
    return (int)(r < 1 ? 1 : (r > 128 ? 128 : r));
}
extern "C" size_t sln_grouped_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int C, int groups, int stride) {
    if (N < 1 || H < 1 || W < 1 || C < 1 || groups < 1 || C % groups || stride < 1) return 0;
    const long npix = (long)N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    return sizeof(float) * (size_t)gw_ranges(npix) * C * (C / groups) * 9;
}
extern "C" int sln_grouped_conv3x3_wgrad_f32(const float *x, const float *gy, const float *y_out, const float *scale,
                                             int N, int H, int W, int C, int groups, int stride, float *gw,
                                             void *workspace, size_t workspace_bytes, sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 1 || groups < 1 || C % groups || stride < 1 || stride > 2 || !gw)
        return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    const long n = (long)C * cg * 9;
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return hipMemsetAsync(gw, 0, sizeof(float) * n, st) == hipSuccess ? SLN_OK : SLN_ERR_LAUNCH;
    if (!x || !gy) return SLN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < sln_grouped_conv3x3_wgrad_workspace_bytes(N, H, W, C, groups, stride))
        return SLN_ERR_WORKSPACE;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const long npix = (long)N * OH * OW;
    const int ranges = gw_ranges(npix);
    const long per = ((npix + ranges - 1) / ranges + GW_TILE - 1) / GW_TILE * GW_TILE;
    const dim3 grid((unsigned)ranges, (unsigned)groups), block(256);
#define SLN_GWG(CGV) hipLaunchKernelGGL(grouped_wgrad3x3_kernel<CGV>, grid, block, 0, st, x, gy, y_out, scale, N, H, W, C, stride, OH, OW, per, (float *)workspace)
    if (cg == 4) SLN_GWG(4); else if (cg == 8) SLN_GWG(8); else if (cg == 16) SLN_GWG(16); else SLN_GWG(32);
#undef SLN_GWG
    hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const float *)workspace, ranges, n, gw);
    return sln_launch_status();
}
