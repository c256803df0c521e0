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

// =====================================================================================================================
// fp16 MFMA path (round 5): BASELINE.json configs[4] in its stated form -- "fp16 MFMA", fp16 storage.
//
// Operands are single scaled fp16 parts (conv.hip, P = 1: h = fp16(v * s), s a per-tensor power of two from the
// delayed-scaling table), accumulation is fp32 inside v_mfma_f32_16x16x32_f16, the epilogue multiplies by
// 1 / (sx * sw).  A grouped 3x3 with CG channels per group is block-diagonal: the GEMM runs on 16-channel tiles,
//   * CG = 32: a tile's 16 outputs reduce over their group's 32 channels -- one 16x16x32 instruction per tap (9);
//   * CG = 16: the tile IS a group; K = 32 holds TWO taps of its 16 channels per instruction (5, the last half empty);
//   * CG = 8 / 4: a tile spans 2 / 4 groups of the same 16 channels; the packed weights are block-diagonal with zeros
//     off the diagonal (2x / 4x the useful FLOPs: these layers are HBM-bound at 2 + 2 B per element anyway).
// The WEIGHTS are the A operand (D[channel][pixel]): a lane then holds four consecutive CHANNELS of one pixel -- 16
// contiguous bytes of an fp32 row, 8 of an fp16 row -- and the epilogue goes from registers to global memory with no
// LDS staging.  Weight fragments come pre-packed in fragment order (one coalesced 16-B load per lane and
// instruction, kept in registers for the block's four pixel tiles); activation fragments are 16-B gathers straight
// from the NHWC rows (a lane = one pixel's 8 consecutive channels of one tap; taps outside the image are zeros).
// A block = 64 channels (one tile per wave: the four waves read the same pixels' 128-B line) x 64 pixels.
// MODE 0: forward (BN affine + ReLU fused, fp32 and / or scaled-fp16 output with its running amax).
// MODE 1: data gradient: a pixel = an INPUT pixel, tap (kh, kw) reads the prepared gradient at
//         (ih + 1 - kh, iw + 1 - kw) / stride where that is an output position; fp32 out.
typedef __attribute__((ext_vector_type(8))) _Float16 gh16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 gh16x4;
typedef __attribute__((ext_vector_type(4))) float gf32x4;
#define SLN_GF16_MAX 65504.0f

struct GroupedParams {
    const _Float16 *x;        // [pixels read][C] scaled fp16
    const _Float16 *wpk;      // packed fragments [C/16][NM][64][8]
    const float *scale, *shift;      // [C] or NULL (MODE 0)
    float *y;                 // [pixels produced][C] fp32 or NULL
    _Float16 *y16;            // [pixels produced][C] scaled fp16 or NULL (MODE 0)
    const float *x_scale, *w_scale, *yq_scale;
    float *yq_amax;
    int32_t *yq_sat;
    // MODE 1 only: the data gradient handed on as the PREPARED gradient of the layer below (chained gradient
    // preparation, conv_hip._GroupedF16Fn): zero where that layer's ReLU was off (mask16 = its output's fp16 part,
    // sign test), times its BN scale (post_scale), stored as a scaled fp16 part (y16) with the running amax
    const _Float16 *mask16;
    const float *post_scale;
    int N, H, W, C, stride, OH, OW, relu;
    long M;                   // produced pixels
};

__device__ __forceinline__ void g_amax_commit(float m, bool sat, float *amax, int32_t *saturated, unsigned *s_word) {
    if (!amax && !saturated) return;
    if (threadIdx.x == 0) { s_word[0] = 0u; s_word[1] = 0u; }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const bool any_sat = __any(sat);
    if ((threadIdx.x & 63) == 0) {
        if (m > 0.f) atomicMax(&s_word[0], __float_as_uint(m));
        if (any_sat) s_word[1] = 1u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bits = s_word[0];
        if (amax && bits > __float_as_uint(*(volatile const float *)amax)) atomicMax((unsigned *)amax, bits);
        if (saturated && s_word[1]) atomicAdd(saturated, 1);
    }
}

template <int CG, int MODE>
__global__ __launch_bounds__(256) void grouped_mfma_kernel(const GroupedParams p) {
    constexpr int NM = CG == 32 ? 9 : 5;          // MFMA instructions per 16 x 16 output tile
    __shared__ unsigned s_word[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.y * 4 + wave, c0 = tile * 16;
    const int cb = CG == 32 ? (c0 & ~31) : c0;    // first channel of the reduced range
    const int q = lane >> 4;                      // this lane's 8-wide k chunk
    gh16x8 a[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) a[m] = *(const gh16x8 *)(p.wpk + (((long)tile * NM + m) * 64 + lane) * 8);
    const float alpha = 1.0f / ((p.x_scale ? *p.x_scale : 1.f) * (p.w_scale ? *p.w_scale : 1.f));
    const float yqs = p.yq_scale ? *p.yq_scale : 1.f;
    const int cq = c0 + 4 * q;                    // the four channels this lane holds of every output pixel
    float sc[4] = {alpha, alpha, alpha, alpha}, sf[4] = {0.f, 0.f, 0.f, 0.f};
    float ps[4] = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.scale) sc[r] = p.scale[cq + r] * alpha;
            if (p.shift) sf[r] = p.shift[cq + r];
        }
    } else if (p.post_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ps[r] = p.post_scale[cq + r];
    }
    const int PH = MODE == 0 ? p.OH : p.H, PW = MODE == 0 ? p.OW : p.W;      // produced grid
    const int QH = MODE == 0 ? p.H : p.OH, QW = MODE == 0 ? p.W : p.OW;      // grid that is read
    float amx = 0.f;
    bool sat = false;
#pragma unroll 1
    for (int pt = 0; pt < 4; ++pt) {
        const long pix = (long)blockIdx.x * 64 + pt * 16 + (lane & 15);
        const bool pok = pix < p.M;
        const long pp = pok ? pix : 0;
        const int pw = (int)(pp % PW), ph = (int)((pp / PW) % PH);
        const long n = pp / ((long)PW * PH);
        gh16x8 b[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int tap = CG == 32 ? m : 2 * m + (q >> 1);
            const int choff = CG == 32 ? cb + 8 * q : cb + 8 * (q & 1);
            const int kh = tap / 3, kw = tap - 3 * kh;
            int qh, qw;
            bool ok = pok && tap < 9;
            if (MODE == 0) {
                qh = ph * p.stride - 1 + kh; qw = pw * p.stride - 1 + kw;
            } else {
                const int th = ph + 1 - kh, tw = pw + 1 - kw;
                ok = ok && th >= 0 && tw >= 0 && (th % p.stride) == 0 && (tw % p.stride) == 0;
                qh = th / p.stride; qw = tw / p.stride;
            }
            ok = ok && (unsigned)qh < (unsigned)QH && (unsigned)qw < (unsigned)QW;
            const gh16x8 z = {};
            b[m] = ok ? *(const gh16x8 *)(p.x + ((n * QH + qh) * (long)QW + qw) * p.C + choff) : z;
        }
        gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[m], acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = acc[r] * sc[r] + sf[r];
            if (MODE == 0 && p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        if (MODE == 1 && pok && p.mask16) {
            const gh16x4 mk = *(const gh16x4 *)(p.mask16 + pix * p.C + cq);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!(mk[r] > (_Float16)0)) v[r] = 0.f;
        }
        if (pok) {
            if (p.y) *(float4 *)(p.y + pix * p.C + cq) = make_float4(v[0], v[1], v[2], v[3]);
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * ps[r];
            }
            if (p.y16) {
                gh16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    amx = fmaxf(amx, fabsf(v[r]));
                    float s = v[r] * yqs;
                    if (fabsf(s) > SLN_GF16_MAX) { s = copysignf(SLN_GF16_MAX, s); sat = true; }
                    h[r] = (_Float16)s;
                }
                *(gh16x4 *)(p.y16 + pix * p.C + cq) = h;
            }
        }
    }
    if (p.y16) g_amax_commit(amx, sat, p.yq_amax, p.yq_sat, s_word);
}

// Weights fp32 [C][CG][3][3] (the parameter) -> fragment-ordered scaled fp16 [C/16][NM][64][8] (+ the tensor's amax).
// flip = 0: forward (rows = output channels, k = (tap, input channel)); 1: data gradient (rows = INPUT channels,
// k = (tap, output channel)).  out == NULL: amax only (first use of the weight's scale slot).
__global__ __launch_bounds__(256) void grouped_pack_weights_kernel(const float *__restrict__ w, int C, int CG, int flip,
                                                                   _Float16 *__restrict__ out,
                                                                   const float *__restrict__ q_scale, float *q_amax,
                                                                   int32_t *q_sat) {
    __shared__ unsigned s_word[2];
    const int NM = CG == 32 ? 9 : 5;
    const long total = (long)(C / 16) * NM * 512;
    const float qs = q_scale ? *q_scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int m = (int)((idx >> 9) % NM), tile = (int)(idx / (512L * NM));
        const int row = lane & 15, k = 8 * (lane >> 4) + j;
        const int c_row = tile * 16 + row;
        int tap, kc;
        if (CG == 32) { tap = m; kc = ((tile * 16) & ~31) + k; }
        else { tap = 2 * m + (k >> 4); kc = tile * 16 + (k & 15); }
        float v = 0.f;
        if (tap < 9 && kc / CG == c_row / CG)
            v = flip ? w[((long)kc * CG + (c_row % CG)) * 9 + tap] : w[((long)c_row * CG + (kc % CG)) * 9 + tap];
        amx = fmaxf(amx, fabsf(v));
        if (!out) continue;
        float s = v * qs;
        if (fabsf(s) > SLN_GF16_MAX) { s = copysignf(SLN_GF16_MAX, s); sat = true; }
        out[idx] = (_Float16)s;
    }
    g_amax_commit(amx, sat, q_amax, q_sat, s_word);
}

// Weight gradient on the matrix cores: per tap D[co][ci] = sum over pixels of gz[pix][co] * x[pix @ tap][ci], K = the
// pixels (32 per instruction).  Both operands are pixel-major in memory, the MFMA wants them channel-major: a wave
// stages 32 pixels of its 16 gz channels and of the nine shifted inputs in LDS as [pixel][channel] rows and reads the
// fragments with ds_read_b64_tr_b16 (the transposing read: a 16-lane group gets a 4 x 16 block column-major).
// A block = 64 channels (a 16-channel tile per wave, wave-private LDS) x one pixel range; partial sums per range
// in the parameter's own order, then grouped_wgrad_reduce_kernel (ordered: the same bits on every run).
template <int LD>
__device__ __forceinline__ gh16x8 g_tr_frag(const _Float16 *tile, int k0, int m0, int lane) {
    // lane 4q+p of a 16-lane group addresses row k0+q, columns m0+4p..+3; lane i receives column m0+i, rows k0..k0+3
    const int li = lane & 15, qq = li >> 2, pq = li & 3;
    // (the transposing read moves 16-bit containers: the bf16 form of the builtin, as in conv.hip's tr_frag)
    typedef __attribute__((ext_vector_type(4))) __bf16 g_bf16x4;
    typedef __attribute__((address_space(3))) g_bf16x4 lds_b16x4;
    const _Float16 *p0 = tile + (k0 + qq) * LD + m0 + 4 * pq;
    const gh16x4 lo = __builtin_bit_cast(gh16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b16x4 *)p0));
    const gh16x4 hi = __builtin_bit_cast(gh16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b16x4 *)(p0 + 4 * LD)));
    gh16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int CG>
__global__ __launch_bounds__(256) void grouped_wgrad_mfma_kernel(const _Float16 *__restrict__ x,
                                                                 const _Float16 *__restrict__ gz, int N, int H, int W,
                                                                 int C, int stride, int OH, int OW, long pix_per_range,
                                                                 const float *gz_scale, const float *x_scale,
                                                                 float *__restrict__ partial) {
    constexpr int KC = CG == 32 ? 32 : 16, NB = KC / 16;
    constexpr int LDG = 16 + 8, LDX = KC + 8;          // row strides in halves: 48 / 48 or 80 B (multiples of 16 B)
    __shared__ __attribute__((aligned(16))) _Float16 s_g[4][32 * LDG];
    __shared__ __attribute__((aligned(16))) _Float16 s_x[4][9 * 32 * LDX];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.y * 4 + wave, c0 = tile * 16;
    const int cb = CG == 32 ? (c0 & ~31) : c0;
    const long total = (long)N * OH * OW;
    const long p0 = (long)blockIdx.x * pix_per_range, p1 = min(total, p0 + pix_per_range);
    gf32x4 acc[9][NB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < NB; ++h) acc[tap][h] = gf32x4{0.f, 0.f, 0.f, 0.f};
    _Float16 *sg = s_g[wave], *sx = s_x[wave];
    const int row = lane >> 1, half8 = (lane & 1) * 8;      // staging: lane = (pixel row of the step, 8-channel chunk)
    const gh16x8 z = {};
    for (long base = p0; base < p1; base += 32) {            // (the same trip count for every wave of the block)
        const long pq = base + row;
        const bool pok = pq < p1;
        const long pp = pok ? pq : 0;
        const int ow = (int)(pp % OW), oh = (int)((pp / OW) % OH);
        const long n = pp / ((long)OW * OH);
        __syncthreads();
        *(gh16x8 *)(sg + row * LDG + half8) = pok ? *(const gh16x8 *)(gz + pp * C + c0 + half8) : z;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = oh * stride - 1 + tap / 3, iw = ow * stride - 1 + tap % 3;
            const bool ok = pok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            const _Float16 *src = x + ((n * H + ih) * (long)W + iw) * C + cb + half8;
#pragma unroll
            for (int h = 0; h < NB; ++h)
                *(gh16x8 *)(sx + (tap * 32 + row) * LDX + 16 * h + half8) = ok ? *(const gh16x8 *)(src + 16 * h) : z;
        }
        __syncthreads();
        const gh16x8 a = g_tr_frag<LDG>(sg, 8 * (lane >> 4), 0, lane);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int h = 0; h < NB; ++h) {
                const gh16x8 b = g_tr_frag<LDX>(sx + tap * 32 * LDX, 8 * (lane >> 4), 16 * h, lane);
                acc[tap][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[tap][h], 0, 0, 0);
            }
    }
    const float alpha = 1.0f / ((gz_scale ? *gz_scale : 1.f) * (x_scale ? *x_scale : 1.f));
    float *out = partial + (size_t)blockIdx.x * C * CG * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c0 + 4 * (lane >> 4) + r, ci = cb + 16 * h + (lane & 15);
                if (co / CG == ci / CG) out[((size_t)co * CG + (ci % CG)) * 9 + tap] = acc[tap][h][r] * alpha;
            }
}

static inline int g_nm(int cg) { return cg == 32 ? 9 : 5; }

extern "C" int64_t sln_grouped_conv3x3_packed_weight_elems(int C, int groups) {
    if (C < 16 || groups < 1 || C % groups || C % 64) return 0;
    return (int64_t)(C / 16) * g_nm(C / groups) * 512;
}

extern "C" int sln_grouped_conv3x3_pack_weights_f16(const float *w, int C, int groups, int flip, uint16_t *out,
                                                    const float *q_scale, float *q_amax, int32_t *q_saturated,
                                                    sln_stream_t stream) {
    sln_enter();
    if (!w || C < 64 || groups < 1 || C % groups || C % 64 || (flip != 0 && flip != 1)) return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    if (!out && !q_amax) return SLN_ERR_INVALID_ARG;
    const long total = sln_grouped_conv3x3_packed_weight_elems(C, groups);
    long grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(grouped_pack_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, C, cg, flip,
                       (_Float16 *)out, q_scale, q_amax, q_saturated);
    return sln_launch_status();
}

// mode 0: forward, x16 [N,H,W,C] -> y / y16 [N,OH,OW,C] = relu?(conv * scale + shift); mode 1: data gradient, x16 =
// the prepared gradient [N,OH,OW,C] (ReLU mask and BN scale already applied: sln_conv_grad_prep_f32, parts = 1),
// w_packed packed with flip = 1 -> y = gx [N,H,W,C] fp32 and / or y16 = the prepared gradient of the layer below
// ((mask16 > 0 ? gx : 0) * post_scale[c] as a scaled fp16 part with its amax: chained gradient preparation).
extern "C" int sln_grouped_conv3x3_f16(const uint16_t *x16, int N, int H, int W, int C, int groups,
                                       const uint16_t *w_packed, int stride, int mode, const float *scale,
                                       const float *shift, int relu, float *y, uint16_t *y16, const float *x_scale,
                                       const float *w_scale, const float *y_q_scale, float *y_q_amax,
                                       int32_t *y_q_saturated, const uint16_t *mask16, const float *post_scale,
                                       sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 64 || groups < 1 || C % groups || C % 64 || stride < 1 || stride > 2 ||
        (mode != 0 && mode != 1))
        return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    if (N == 0) return SLN_OK;
    if (!x16 || !w_packed || (!y && !y16) || (mode == 0 && (mask16 || post_scale))) return SLN_ERR_INVALID_ARG;
    if ((((size_t)x16) | ((size_t)y) | ((size_t)y16) | ((size_t)w_packed) | ((size_t)mask16)) & 15) return SLN_ERR_INVALID_ARG;
    GroupedParams p;
    p.x = (const _Float16 *)x16; p.wpk = (const _Float16 *)w_packed; p.scale = scale; p.shift = shift;
    p.y = y; p.y16 = (_Float16 *)y16; p.x_scale = x_scale; p.w_scale = w_scale; p.yq_scale = y_q_scale;
    p.yq_amax = y_q_amax; p.yq_sat = y_q_saturated;
    p.mask16 = (const _Float16 *)mask16; p.post_scale = post_scale;
    p.N = N; p.H = H; p.W = W; p.C = C; p.stride = stride; p.relu = relu;
    p.OH = (H + 2 - 3) / stride + 1; p.OW = (W + 2 - 3) / stride + 1;
    p.M = mode == 0 ? (long)N * p.OH * p.OW : (long)N * H * W;
    const long gx = (p.M + 63) / 64;
    if (gx > 2147483647L) return SLN_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)gx, (unsigned)(C / 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SLN_GM(CGV) do { if (mode == 0) hipLaunchKernelGGL((grouped_mfma_kernel<CGV, 0>), grid, block, 0, st, p); \
                         else hipLaunchKernelGGL((grouped_mfma_kernel<CGV, 1>), grid, block, 0, st, p); } while (0)
    if (cg == 4) SLN_GM(4); else if (cg == 8) SLN_GM(8); else if (cg == 16) SLN_GM(16); else SLN_GM(32);
#undef SLN_GM
    return sln_launch_status();
}

// gw [C][C/groups][3][3] fp32 from the prepared gradient gz16 [N,OH,OW,C] and the layer's input x16 [N,H,W,C] (both
// scaled fp16 parts).  workspace: sln_grouped_conv3x3_wgrad_workspace_bytes() bytes (the same plan as the fp32 path).
extern "C" int sln_grouped_conv3x3_wgrad_f16(const uint16_t *x16, const uint16_t *gz16, int N, int H, int W, int C,
                                             int groups, int stride, const float *gz_scale, const float *x_scale,
                                             float *gw, void *workspace, size_t workspace_bytes, sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 64 || groups < 1 || C % groups || C % 64 || stride < 1 || stride > 2 || !gw)
        return SLN_ERR_INVALID_ARG;
    const int cg = C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SLN_ERR_UNSUPPORTED;
    const long n = (long)C * cg * 9;
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return hipMemsetAsync(gw, 0, sizeof(float) * n, st) == hipSuccess ? SLN_OK : SLN_ERR_LAUNCH;
    if (!x16 || !gz16 || ((((size_t)x16) | ((size_t)gz16)) & 15)) return SLN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < sln_grouped_conv3x3_wgrad_workspace_bytes(N, H, W, C, groups, stride))
        return SLN_ERR_WORKSPACE;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const long npix = (long)N * OH * OW;
    const int ranges = gw_ranges(npix);
    const long per = ((npix + ranges - 1) / ranges + 31) / 32 * 32;
    const dim3 grid((unsigned)ranges, (unsigned)(C / 64)), block(256);
#define SLN_GWM(CGV) hipLaunchKernelGGL(grouped_wgrad_mfma_kernel<CGV>, grid, block, 0, st, (const _Float16 *)x16, (const _Float16 *)gz16, N, H, W, C, stride, OH, OW, per, gz_scale, x_scale, (float *)workspace)
    if (cg == 4) SLN_GWM(4); else if (cg == 8) SLN_GWM(8); else if (cg == 16) SLN_GWM(16); else SLN_GWM(32);
#undef SLN_GWM
    hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const float *)workspace, ranges, n, gw);
    return sln_launch_status();
}
