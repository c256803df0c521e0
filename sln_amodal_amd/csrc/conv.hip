// Implicit-GEMM convolution stack for gfx950 on the 16-bit matrix cores with fp32-class accuracy
// ("split operands").
//
// Every fp32 operand v is carried as a sum of P 16-bit parts and a product of two operands is
// accumulated in fp32 (inside the MFMA) from the dominant part pairs:
//   P = 2 (default): two fp16 parts of v*s, s a per-tensor power of two (delayed scaling):
//           h0 = fp16_rne(v*s), h1 = fp16_rne(v*s - h0)  -- 22 significant bits --
//           products a1*b0 + a0*b1 + a0*b0 on v_mfma_f32_32x32x16_f16, accumulator * 1/(sa*sb):
//           1-3e-7 relative per layer against fp64, the accuracy of an fp32 GEMM
//   P = 3: three bf16 parts, vi = bf16_rne(v - v0 - ... - v(i-1)), six products
//           a2*b0 + a0*b2 + a1*b1 + a1*b0 + a0*b1 + a0*b0 on v_mfma_f32_32x32x16_bf16 (~1e-8, no scales)
//   P = 1 (round 5, BASELINE.json configs[4] "fp16 MFMA"): ONE scaled fp16 part h0 = fp16_rne(v*s) -- plain fp16
//           storage of activations, weights and gradients with the same per-tensor delayed scaling, one product
//           a0*b0 per multiply-add, fp32 accumulate: fp16-class results (~5e-4 relative per element), selected
//           per model (conv_hip.PARTS = 1), never the default; runs on the generic 128-wide kernels below.
// (two UNSCALED bf16 parts, ~4e-6 per layer, were measured in round 1 and rejected.)
// The 16-bit MFMA issues at 16x the rate of the fp32-input MFMA (MI355X_MICROARCH.md), so 3 / 6
// part products per fp32-equivalent product leave 5.3x / 2.7x the fp32-MFMA roofline.  Details of the
// two formats: "operand formats" below.
//
// Data flow of one conv layer (all activations NHWC = [pixels][channels]):
//   act_split      x fp32 -> xparts [P][M][Cp] bf16 (one HBM-bound pass; the
//                  consumer GEMMs then need no VALU work at all on their operands)
//   conv_fwd       y = relu?( conv(xparts, wparts)*scale + shift + residual )
//   grad_prep      gz = gy * (y>0) * scale  -> gzparts (+ fp32 gu, + bias grad) -- or, where the
//                  layer's output has one known reader (two for the RPN), done by that reader's
//                  data-gradient epilogue (mask / post_scale / colsum / parts-only modes)
//   conv_fwd       gx = conv(gzparts, wTparts)        (data gradient, stride 1)
//   conv_wgrad     gw[co][tap][ci] = sum_pix gz[pix][co] * x[pix@tap][ci]
//                  ("TN" GEMM: both operands are pixel-major, fragments are
//                  fetched with the ds_read_b64_tr_b16 transposing LDS read)
//
// (conv_fwd256_kernel / conv_wgrad256_kernel further down are the same GEMMs on 256x256 tiles with
// an LDS-DMA pipeline; sln_conv_fwd_tile / sln_conv_wgrad_tile pick per launch.)
// conv_fwd GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin walked
// tap by tap in 32-channel chunks; taps outside the image read zero (the
// reference's SamePad2d / conv padding).  Tile 128x128x32, 256 threads = 4 waves
// (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32.  Forward LDS rows are 32 bf16
// (64 B) with the 16-B chunk index XOR-swizzled by row bits 2..3 (48 KB per block, 3
// blocks per CU): the ds_read_b128 fragment reads of a 16-lane group hit 16 distinct
// 4-bank groups (conflict-free).  Register prefetch of the next k-step.
#include <atomic>
#include <cstdlib>
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BM 128
#define BN 128
#define BK 32
#ifndef SLN_FWD128_BLOCKS
#define SLN_FWD128_BLOCKS 3     // resident blocks per CU the 128^2 forward kernel is compiled for (48 KB LDS each)
#endif

// ---------------------------------------------------------------- operand formats
// P = 3: three bf16 parts, six part products per fp32 product (see the top of the file).
// P = 2: two fp16 parts of v*s, s a per-tensor power of two ("scaled split-fp16"):
//          h0 = fp16_rne(v*s),  h1 = fp16_rne(v*s - h0)          (22 significant bits)
//        and three part products a0*b0 + a0*b1 + a1*b0 on v_mfma_f32_32x32x16_f16; the dropped
//        a1*b1 term is 2^-22 relative.  fp16's narrow exponent needs the scale: s is chosen so that
//        max|v|*s sits near 2^11 (sln_scale_update_f32, from the maximum over a window of recent steps), which leaves 2^5 of head room before
//        +-65504 and keeps every element down to max|v| * 2^-14 at full 22-bit precision (smaller
//        ones degrade gracefully: absolute error <= max|v| * 2^-36).  The scale of a tensor that a
//        conv epilogue writes is not known before that conv has run, so s comes from the amax the
//        SAME tensor had the last time it was produced (delayed scaling); producers track the
//        running amax (atomic max) and clamp + count anything beyond +-65504 instead of emitting inf.
//        Scales are powers of two, so v*s and acc/(sa*sb) are exact: the only rounding is in the
//        parts themselves.  Parts are stored in the same 16-bit containers as bf16 parts.
#define SLN_F16_MAX 65504.0f
// XOR swizzle of the 16-B chunk index of LDS row r in the 256-wide kernels' stage images (64-B rows; activations by DMA,
// weights pre-tiled by split_weights_tiledh_kernel).  ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and the 16x16x32 fragment read has lane l
// on row l % 16, chunk l / 16.  With bits 2..3 of the row (the swizzle written in round 2 for the 32x32x16 fragments,
// conflict-free there) every group put two lanes on each 16-B bank slot: 8 LDS cycles per read instead of 4 in the
// production body (round 4; SQ_LDS_BANK_CONFLICT in profiles/).  Bits 1..2 are conflict-free for the 16x16x32 read at
// any row offset and 2-way for the diagnostic 32x32x16 body.
#define SLN_SWZH(r) (((r) >> 1) & 3)
// The WEIGHT stage images take row bits 1 and 3 instead (round 5): conflict-free for the plain 16 consecutive rows of a
// fragment read (and for the permuted row order of round 5's register epilogue, measured not faster and removed in
// round 6: profiles/HISTORY_r5.md; simulated with the guide's four 16-lane groups, tools/lds_swizzle_sim.py).  Written
// by split_weights_tiledh_kernel / split_weights_batch_kernel, read by every B-fragment address of conv_fwd256h_kernel
// and conv_fwd128x256h_kernel.
#define SLN_SWZW(r) ((((r) >> 1) & 1) | ((((r) >> 3) & 1) << 1))
#ifndef SLN_W8_DEPTH
#define SLN_W8_DEPTH 2      // half slabs of residual / mask rows in flight ahead, single-epilogue instances of the 256^2 kernel
#endif
#ifndef SLN_W8_DOUBLE_STAGE
#define SLN_W8_DOUBLE_STAGE 1   // two staging slabs in the epilogues of the kernels that own the CU's LDS (0: one, A/B builds)
#endif

struct SplitScale {
    const float *scale;   // device scalar s (NULL = 1)
    float *amax;          // device scalar: running max |v| (NULL = not tracked)
    int *saturated;       // device counter of clamped elements (NULL = not counted)
};

// ---------------------------------------------------------------- elementwise
// Returns true when an element had to be clamped (P = 2 only).
template <int P>
__device__ __forceinline__ bool split4(const float4 v, bf16x4 *parts, float s = 1.f) {
    float r[4] = {v.x, v.y, v.z, v.w};
    if (P == 3) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            bf16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = (__bf16)r[j];
                r[j] -= (float)h[j];
            }
            parts[p] = h;
        }
        return false;
    }
    bool sat = false;
    h16x4 h0, h1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float q = r[j] * s;                      // exact: s is a power of two
        if (fabsf(q) > SLN_F16_MAX) {            // (NaN compares false and stays NaN)
            q = copysignf(SLN_F16_MAX, q);
            sat = true;
        }
        h0[j] = (_Float16)q;
        if (P == 2) h1[j] = (_Float16)(q - (float)h0[j]);
    }
    parts[0] = __builtin_bit_cast(bf16x4, h0);
    if (P == 2) parts[1] = __builtin_bit_cast(bf16x4, h1);      // P = 1: the scaled fp16 value alone (fp16 storage)
    return sat;
}

// Two fp16 parts of four values that are KNOWN to be inside the fp16 range after scaling (|v| * s <= 65504: the
// caller has compared the row's maximum, wave-uniformly): the split without the three clamp instructions per
// element.  The epilogues are bound by their VALU work (~11 instructions per output element, 64 K elements per
// 256 x 256 tile = 18 k cycles of the four SIMDs), not by memory; bit-identical to split4<2> on such values.
__device__ __forceinline__ void split4_inrange(const float4 v, bf16x4 *parts, float s) {
    const float r[4] = {v.x, v.y, v.z, v.w};
    h16x4 h0, h1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float q = r[j] * s;                // exact: s is a power of two
        h0[j] = (_Float16)q;
        h1[j] = (_Float16)(q - (float)h0[j]);
    }
    parts[0] = __builtin_bit_cast(bf16x4, h0);
    parts[1] = __builtin_bit_cast(bf16x4, h1);
}

// max(|a|, |b|, |c|) in ONE instruction (source modifiers).  Written out because fmaxf(fabsf(a), ...) costs a
// canonicalising v_max_f32 |a|, |a| per operand in IEEE mode: 14 instructions for a row of eight instead of 4.
__device__ __forceinline__ float max3abs(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float amax8(const float v[8]) {
    return max3abs(max3abs(max3abs(v[0], v[1], v[2]), v[3], v[4]), max3abs(v[5], v[6], v[7]), 0.f);
}

__device__ __forceinline__ float amax4(float m, const float v[4]) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// Block-level commit of a per-thread running max / saturation flag: wave shuffle, one LDS word,
// then at most one global atomic per block (skipped when the recorded amax already covers it).
__device__ __forceinline__ void amax_commit(float m, bool sat, const SplitScale &q, unsigned *s_word) {
    if (!q.amax && !q.saturated) return;
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
        if (q.amax && bits > __float_as_uint(*(volatile const float *)q.amax))
            atomicMax((unsigned *)q.amax, bits);    // non-negative floats order like their bits
        if (q.saturated && s_word[1]) atomicAdd(q.saturated, 1);
    }
}

// x [M][C] fp32 -> parts [P][M][Cp] (channels C..Cp-1 zero).  C % 4 == 0 fast path.
// parts == NULL: only the amax is taken (first use of a tensor slot, before it has a scale).
template <int P>
__global__ __launch_bounds__(256) void act_split_kernel(const float *__restrict__ x, long M, int C,
                                                        int Cp, __bf16 *__restrict__ parts, SplitScale q) {
    __shared__ unsigned s_word[2];
    const int q4 = Cp / 4;
    const long total = M * q4;
    const long pstride = M * Cp;
    const float qs = (P <= 2 && q.scale) ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long m = e / q4;
        const int c = (int)(e - m * q4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < C && (C & 3) == 0) {
            v = *(const float4 *)(x + m * C + c);
        } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < 4; ++j)
                if (c + j < C) t[j] = x[m * C + c + j];
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        if (P <= 2) {
            const float t4[4] = {v.x, v.y, v.z, v.w};
            amx = amax4(amx, t4);
        }
        if (!parts) continue;
        bf16x4 ps[P];
        sat |= split4<P>(v, ps, qs);
#pragma unroll
        for (int p = 0; p < P; ++p) *(bf16x4 *)(parts + p * pstride + m * Cp + c) = ps[p];
    }
    if (P <= 2) amax_commit(amx, sat, q, s_word);
}

// Patch matrix of a small-Cin convolution (the 3-channel 7x7/2 stems, modals.py:311 and
// resnet_deeplab.py's conv1), emitted directly as operand parts: row m = output pixel (n, oh, ow),
// column k = (kh, kw, c), K = KH*KW*C padded with zeros to K_pad; taps outside the image are zero.
// The stem then IS a 1x1 convolution over K_pad channels and runs on the ordinary forward / weight-
// gradient kernels.  Lanes run along k, so a wave writes 512 contiguous bytes per part; the tap ->
// (input offset, kh, kw) table is built once per block in LDS.  parts == NULL: amax only.
template <int P>
__global__ __launch_bounds__(256) void im2col_split_kernel(const float *__restrict__ x, int N, int H, int W,
                                                           int C, int KH, int KW, int sh, int sw, int pt, int pl,
                                                           int OH, int OW, int Kp, long pstride, long row0,
                                                           __bf16 *__restrict__ parts, SplitScale q) {
    extern __shared__ __attribute__((aligned(16))) int s_tab[];   // [Kp] input offset, [Kp] kh | kw << 8 | valid << 16
    __shared__ unsigned s_word[2];
    int *s_off = s_tab, *s_pos = s_tab + Kp;
    const int K = KH * KW * C;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        const int kh = k / (KW * C), r = k - kh * KW * C;
        const int kw = r / C;
        s_off[k] = (kh * W + kw) * C + (r - kw * C);
        s_pos[k] = k < K ? (kh | (kw << 8) | (1 << 16)) : 0;
    }
    __syncthreads();
    const int q4 = Kp / 4;
    const long M = (long)N * OH * OW, total = M * q4;
    const float qs = (P <= 2 && q.scale) ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long m = e / q4;
        const int kq = (int)(e - m * q4) * 4;
        const int ow = (int)(m % OW);
        const long t = m / OW;
        const int oh = (int)(t % OH), n = (int)(t / OH);
        const int ih0 = oh * sh - pt, iw0 = ow * sw - pl;
        const float *px = x + (((long)n * H + ih0) * W + iw0) * C;
        float v[4];
        // one 16-B read per table (Kp % 8 == 0): four dword reads at a stride of four dwords per lane were 4-way bank
        // conflicts (SQ_LDS_BANK_CONFLICT 74 % of this kernel's LDS cycles, profiles/r4_ah_head_pmc_lds.json)
        const int4 pos4 = *(const int4 *)(s_pos + kq), off4 = *(const int4 *)(s_off + kq);
        const int posv[4] = {pos4.x, pos4.y, pos4.z, pos4.w}, offv[4] = {off4.x, off4.y, off4.z, off4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = posv[j];
            const int ih = ih0 + (pos & 255), iw = iw0 + ((pos >> 8) & 255);
            const bool ok = (pos >> 16) && ih >= 0 && ih < H && iw >= 0 && iw < W;
            v[j] = ok ? px[offv[j]] : 0.f;
        }
        if (P <= 2) amx = amax4(amx, v);
        if (!parts) continue;
        bf16x4 ps[P];
        sat |= split4<P>(make_float4(v[0], v[1], v[2], v[3]), ps, qs);
#pragma unroll
        for (int p = 0; p < P; ++p) *(bf16x4 *)(parts + p * pstride + (row0 + m) * Kp + kq) = ps[p];
    }
    if (P <= 2) amax_commit(amx, sat, q, s_word);
}

// Adjoint of the patch matrix (data gradient of a stem, needed only when the image itself carries a
// gradient): gx[n, ih, iw, c] = sum over the taps (kh, kw) whose output pixel exists of
// cols[(n, oh, ow), (kh, kw, c)].  Gather form: one thread per input element, no atomics.
__global__ __launch_bounds__(256) void col2im_kernel(const float *__restrict__ cols, int N, int H, int W, int C,
                                                     int KH, int KW, int sh, int sw, int pt, int pl, int OH,
                                                     int OW, int Kp, float *__restrict__ gx) {
    const long total = (long)N * H * W * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long t = e / C;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H), n = (int)(t / H);
        float acc = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            const int ty = ih + pt - kh;
            if (ty < 0 || ty % sh) continue;
            const int oh = ty / sh;
            if (oh >= OH) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int tx = iw + pl - kw;
                if (tx < 0 || tx % sw) continue;
                const int ow = tx / sw;
                if (ow >= OW) continue;
                acc += cols[(((long)n * OH + oh) * OW + ow) * Kp + (kh * KW + kw) * C + c];
            }
        }
        gx[e] = acc;
    }
}

// gz = gy * (y > 0 ? 1 : 0) * scale[c]; writes gu = gy*(y>0) (fp32, optional), the
// bf16 parts of gz, and accumulates the per-channel sum of gz (bias gradient).
// A thread keeps one channel quad and walks rows (no index division, bias sums stay in
// registers until the end); two rows are in flight per iteration.
// Source of a gradient that is still "behind" a max-pool (the backbone stem: conv -> BN -> ReLU -> 3x3/2 pool): the
// pooled gradient [N,OH,OW,C] and the pool's winning taps; the row the preparation wants is then gathered here --
// the sum over the windows whose winner is this pixel (csrc/maxpool.hip: the same walk) -- instead of being read
// from a 1-GB fp32 map that a separate pool-backward kernel wrote.
struct PoolSrc {
    const float *g;            // NULL: the gradient is read from gy as usual
    const unsigned char *arg;
    int H, W, K, S, pt, pl, OH, OW;
};

__device__ __forceinline__ void pooled_row(const PoolSrc &ps, long m, int c, int C, float g[4]) {
    const int iw = (int)(m % ps.W);
    const long t = m / ps.W;
    const int ih = (int)(t % ps.H);
    const long n = t / ps.H;
    g[0] = g[1] = g[2] = g[3] = 0.f;
    const int oh_lo = max(0, (ih + ps.pt - ps.K + ps.S) / ps.S), oh_hi = min(ps.OH - 1, (ih + ps.pt) / ps.S);
    const int ow_lo = max(0, (iw + ps.pl - ps.K + ps.S) / ps.S), ow_hi = min(ps.OW - 1, (iw + ps.pl) / ps.S);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        const int kh = ih - (oh * ps.S - ps.pt);
        if (kh < 0 || kh >= ps.K) continue;
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            const int kw = iw - (ow * ps.S - ps.pl);
            if (kw < 0 || kw >= ps.K) continue;
            const long o = ((n * ps.OH + oh) * ps.OW + ow) * C + c;
            const uchar4 a = *(const uchar4 *)(ps.arg + o);
            const float4 gv = *(const float4 *)(ps.g + o);
            const unsigned tap = (unsigned)(kh * ps.K + kw);
            if (a.x == tap) g[0] += gv.x;
            if (a.y == tap) g[1] += gv.y;
            if (a.z == tap) g[2] += gv.z;
            if (a.w == tap) g[3] += gv.w;
        }
    }
}

__device__ __forceinline__ void grad_prep_row(const float *__restrict__ gy, const float *__restrict__ y,
                                              const __bf16 *__restrict__ y16, int Cp, long m, int c, int C,
                                              bool vec, float g[4], const PoolSrc &ps) {
    // ReLU pattern: the layer's fp32 output y, or part 0 of an output that exists as parts only (y16, rows of
    // Cp 16-bit words: h0 > 0)
    if (ps.g) {            // (launcher: C % 4 == 0, fp32 pattern or none)
        pooled_row(ps, m, c, C, g);
        if (y) {
            const float4 yy = *(const float4 *)(y + m * C + c);
            if (!(yy.x > 0.f)) g[0] = 0.f;
            if (!(yy.y > 0.f)) g[1] = 0.f;
            if (!(yy.z > 0.f)) g[2] = 0.f;
            if (!(yy.w > 0.f)) g[3] = 0.f;
        }
        return;
    }
    g[0] = g[1] = g[2] = g[3] = 0.f;
    if (y16) {
        const h16x4 k4 = *(const h16x4 *)(y16 + m * Cp + c);      // (c + 3 < Cp: Cp % 8 == 0, c % 4 == 0)
        if (vec) {
            const float4 v = *(const float4 *)(gy + m * C + c);
            g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
        } else {
            for (int j = 0; j < 4; ++j)
                if (c + j < C) g[j] = gy[m * C + c + j];
        }
        if (!(k4.x > (_Float16)0)) g[0] = 0.f;
        if (!(k4.y > (_Float16)0)) g[1] = 0.f;
        if (!(k4.z > (_Float16)0)) g[2] = 0.f;
        if (!(k4.w > (_Float16)0)) g[3] = 0.f;
        return;
    }
    if (vec) {
        const float4 v = *(const float4 *)(gy + m * C + c);
        g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
        if (y) {
            const float4 yy = *(const float4 *)(y + m * C + c);
            if (!(yy.x > 0.f)) g[0] = 0.f;
            if (!(yy.y > 0.f)) g[1] = 0.f;
            if (!(yy.z > 0.f)) g[2] = 0.f;
            if (!(yy.w > 0.f)) g[3] = 0.f;
        }
    } else {
        for (int j = 0; j < 4; ++j)
            if (c + j < C) {
                float v = gy[m * C + c + j];
                if (y && !(y[m * C + c + j] > 0.f)) v = 0.f;
                g[j] = v;
            }
    }
}

template <int P>
__device__ __forceinline__ void grad_prep_emit(float g[4], long m, int c, int C, int Cp, bool vec,
                                               const float sc[4], float acc[4], long pstride,
                                               float *__restrict__ gu, __bf16 *__restrict__ parts,
                                               float qs, float &amx, bool &sat) {
    if (gu && parts) {
        if (vec) *(float4 *)(gu + m * C + c) = make_float4(g[0], g[1], g[2], g[3]);
        else
            for (int j = 0; j < 4; ++j)
                if (c + j < C) gu[m * C + c + j] = g[j];
    }
    // The product must be rounded to fp32 BEFORE it is split.  With contraction allowed the
    // compiler fuses it into split4's first residual subtraction (v_pk_fma_f32 g, sc, -h0 in
    // the ISA; __fmul_rn is a plain multiply on this toolchain and does not prevent it), and the
    // parts then encode the unrounded product -- 1 ulp away from what the conv epilogue's fused
    // preparation (mask/colsum mode) produces for the same gradient.
    {
#pragma clang fp contract(off)
        for (int j = 0; j < 4; ++j) g[j] = g[j] * sc[j];
    }
    if (P <= 2) amx = amax4(amx, g);
    if (!parts) return;                      // amax-only pass (first use of the tensor's scale slot)
    for (int j = 0; j < 4; ++j) acc[j] += g[j];
    bf16x4 ps[P];
    sat |= split4<P>(make_float4(g[0], g[1], g[2], g[3]), ps, qs);
#pragma unroll
    for (int p = 0; p < P; ++p) *(bf16x4 *)(parts + p * pstride + m * Cp + c) = ps[p];
}

template <int P>
__global__ __launch_bounds__(256) void grad_prep_kernel(const float *__restrict__ gy,
                                                        const float *__restrict__ y,
                                                        const __bf16 *__restrict__ y16,
                                                        const float *__restrict__ scale, long M, int C,
                                                        int Cp, float *__restrict__ gu,
                                                        __bf16 *__restrict__ parts,
                                                        float *__restrict__ gbias, SplitScale q, PoolSrc ps) {
    __shared__ float s_bias[1024];
    __shared__ unsigned s_word[2];
    const int q4 = Cp / 4;
    int tw = 1, sh = 0;                       // tw = channel quads per block row (power of two)
    while (tw < q4 && tw < 256) { tw <<= 1; ++sh; }
    const int R = 256 >> sh;                  // rows per block pass
    const int cq_l = threadIdx.x & (tw - 1), r0 = threadIdx.x >> sh;
    const long pstride = M * Cp;
    const long rstep = (long)gridDim.x * R;
    const float qs = (P <= 2 && q.scale) ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (int ct = 0; ct * tw < q4; ++ct) {
        const int c = (ct * tw + cq_l) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < Cp) {
            const bool vec = (c + 3 < C) && (C & 3) == 0;
            float sc[4];
            for (int j = 0; j < 4; ++j) sc[j] = (c + j < C) ? (scale ? scale[c + j] : 1.f) : 0.f;
            long m = (long)blockIdx.x * R + r0;
            for (; m + rstep < M; m += 2 * rstep) {
                float g0[4], g1[4];
                grad_prep_row(gy, y, y16, Cp, m, c, C, vec, g0, ps);
                grad_prep_row(gy, y, y16, Cp, m + rstep, c, C, vec, g1, ps);
                grad_prep_emit<P>(g0, m, c, C, Cp, vec, sc, acc, pstride, gu, parts, qs, amx, sat);
                grad_prep_emit<P>(g1, m + rstep, c, C, Cp, vec, sc, acc, pstride, gu, parts, qs, amx, sat);
            }
            if (m < M) {
                float g0[4];
                grad_prep_row(gy, y, y16, Cp, m, c, C, vec, g0, ps);
                grad_prep_emit<P>(g0, m, c, C, Cp, vec, sc, acc, pstride, gu, parts, qs, amx, sat);
            }
        }
        if (gbias && parts) {
            for (int i = threadIdx.x; i < tw * 4; i += 256) s_bias[i] = 0.f;
            __syncthreads();
            for (int j = 0; j < 4; ++j)
                if (acc[j] != 0.f) atomicAdd(&s_bias[cq_l * 4 + j], acc[j]);
            __syncthreads();
            for (int i = threadIdx.x; i < tw * 4; i += 256) {
                const int cc = ct * tw * 4 + i;
                if (cc < C && s_bias[i] != 0.f) atomicAdd(gbias + cc, s_bias[i]);
            }
            __syncthreads();
        }
    }
    if (P <= 2) amax_commit(amx, sat, q, s_word);
}

// K order of conv_fwd256_kernel: stage s <-> (tap, 16-channel chunk cc).  64-channel group major,
// tap, then the group's chunks -- the four stages of a (group, tap) walk one 128-B line of every
// pixel, and the KH*KW shifted windows of a group are read within 4*KH*KW consecutive stages, while
// their lines are still in L2 (tap-major order re-fetched them from beyond L2: 4.7x the algorithmic
// reads, PMC).  ncc = number of 16-channel chunks.
#define T2 256
#define T2K 16
__host__ __device__ __forceinline__ void fwd256_stage(int s, int ntap, int ncc, int &tap, int &cc) {
    const int gfull = ncc / 4, nsub_tail = ncc - 4 * gfull;
    if (s < gfull * ntap * 4) {
        const int g = s / (ntap * 4), r = s - g * (ntap * 4);
        tap = r >> 2; cc = 4 * g + (r & 3);
    } else {
        const int r = s - gfull * ntap * 4;
        tap = r / nsub_tail; cc = 4 * gfull + (r - tap * nsub_tail);
    }
}

// Weights in the LDS-image order of conv_fwd256_kernel ("tiled" layout): for every 256-row Cout tile
// nt, stage s and part pp one contiguous 8-KB block = the 256 x 32 B region the kernel's DMA drops into
// LDS verbatim (row r at 32 r, its two 16-B halves XOR-swizzled with bit 3 of r; rows >= Cout and
// channels >= Cin are zero), so that a 1-KiB DMA piece reads 8 consecutive 128-B lines instead of 32
// B from each of 32 lines (tools/micro/dma_patterns.hip: 19 vs 70 cycles per piece and CU, L2-hot).
__global__ __launch_bounds__(256) void split_weights_tiled_kernel(const float *__restrict__ w, int O, int I,
                                                                  int KH, int KW, long s_o, long s_i,
                                                                  long s_kh, long s_kw, int flip, int P,
                                                                  __bf16 *__restrict__ out, SplitScale q) {
    __shared__ unsigned s_word[2];
    const int ntap = KH * KW, ncc = (I + T2K - 1) / T2K, nk = ntap * ncc, gn = (O + T2 - 1) / T2;
    const long per_part = (long)T2 * T2K;                       // 4096 elements = 8 KB
    const long total = (long)gn * nk * per_part;                // logical elements (one part)
    const float qs = (P <= 2 && q.scale) ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int pos = (int)(idx % per_part);
        const long blk = idx / per_part;                        // nt * nk + s
        const int st = (int)(blk % nk), nt = (int)(blk / nk);
        const int r = pos >> 4, j = pos & 15;
        const int jl = (((j >> 3) ^ ((r >> 3) & 1)) << 3) | (j & 7);   // logical channel within the chunk
        int tap, cc;
        fwd256_stage(st, ntap, ncc, tap, cc);
        const int kh = tap / KW, kw = tap - kh * KW;
        const int skh = flip ? KH - 1 - kh : kh, skw = flip ? KW - 1 - kw : kw;
        const int o = nt * T2 + r, i = cc * T2K + jl;
        float v = (o < O && i < I) ? w[o * s_o + i * s_i + skh * s_kh + skw * s_kw] : 0.f;
        const long dst = blk * P * per_part + pos;
        if (P == 2) {
            amx = fmaxf(amx, fabsf(v));
            if (!out) continue;
            float qv = v * qs;
            if (fabsf(qv) > SLN_F16_MAX) { qv = copysignf(SLN_F16_MAX, qv); sat = true; }
            const _Float16 h0 = (_Float16)qv;
            const _Float16 h1 = (_Float16)(qv - (float)h0);
            out[dst] = __builtin_bit_cast(__bf16, h0);
            out[dst + per_part] = __builtin_bit_cast(__bf16, h1);
            continue;
        }
        for (int pp = 0; pp < P; ++pp) {
            const __bf16 h = (__bf16)v;
            out[dst + pp * per_part] = h;
            v -= (float)h;
        }
    }
    if (P <= 2) amax_commit(amx, sat, q, s_word);
}

// ---- conv_fwd256h_kernel (two fp16 parts, 32-channel stages) ----
#define T2H 32
// stage s <-> (tap, 32-channel chunk cc): 64-channel group major, tap, the group's two chunks (the same
// L2 argument as fwd256_stage).  ncc = number of 32-channel chunks.
__host__ __device__ __forceinline__ void fwd256h_stage(int s, int ntap, int ncc, int &tap, int &cc) {
    const int gfull = ncc / 2;
    if (s < gfull * ntap * 2) {
        const int g = s / (ntap * 2), r = s - g * (ntap * 2);
        tap = r >> 1; cc = 2 * g + (r & 1);
    } else {
        tap = s - gfull * ntap * 2; cc = 2 * gfull;
    }
}

// Weights in conv_fwd256h_kernel's LDS-image order: per (Cout tile, stage, part) one contiguous 16-KB
// block of 256 rows x 64 B, the 16-B chunk index of row r XOR-swizzled with SLN_SWZH(r).
__global__ __launch_bounds__(256) void split_weights_tiledh_kernel(const float *__restrict__ w, int O, int I,
                                                                   int KH, int KW, long s_o, long s_i,
                                                                   long s_kh, long s_kw, int flip,
                                                                   __bf16 *__restrict__ out, SplitScale q) {
    __shared__ unsigned s_word[2];
    const int ntap = KH * KW, ncc = (I + T2H - 1) / T2H, nk = ntap * ncc, gn = (O + T2 - 1) / T2;
    const long per_part = (long)T2 * T2H;                       // 8192 elements = 16 KB
    const long total = (long)gn * nk * per_part;
    const float qs = q.scale ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int pos = (int)(idx % per_part);
        const long blk = idx / per_part;                        // nt * nk + s
        const int st = (int)(blk % nk), nt = (int)(blk / nk);
        const int r = pos >> 5, j = pos & 31;
        const int jl = ((((j >> 3) ^ SLN_SWZW(r))) << 3) | (j & 7);   // logical channel within the chunk
        int tap, cc;
        fwd256h_stage(st, ntap, ncc, tap, cc);
        const int kh = tap / KW, kw = tap - kh * KW;
        const int skh = flip ? KH - 1 - kh : kh, skw = flip ? KW - 1 - kw : kw;
        const int o = nt * T2 + r, i = cc * T2H + jl;
        const float v = (o < O && i < I) ? w[o * s_o + i * s_i + skh * s_kh + skw * s_kw] : 0.f;
        amx = fmaxf(amx, fabsf(v));
        if (!out) continue;
        float qv = v * qs;
        if (fabsf(qv) > SLN_F16_MAX) { qv = copysignf(SLN_F16_MAX, qv); sat = true; }
        const _Float16 h0 = (_Float16)qv;
        const _Float16 h1 = (_Float16)(qv - (float)h0);
        const long dst = blk * 2 * per_part + pos;
        out[dst] = __builtin_bit_cast(__bf16, h0);
        out[dst + per_part] = __builtin_bit_cast(__bf16, h1);
    }
    amax_commit(amx, sat, q, s_word);
}

// fp32 weights (any strides) -> [P][O][KH][KW][Ip] parts (i >= I zero).
// flip=1 mirrors the taps (with O/I swapped through the strides this expresses the
// data-gradient convolution as a forward convolution).  out == NULL: amax only.
__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ w, int O, int I,
                                                            int Ip, int KH, int KW, long s_o, long s_i,
                                                            long s_kh, long s_kw, int flip, int P,
                                                            __bf16 *__restrict__ out, SplitScale q) {
    __shared__ unsigned s_word[2];
    const long total = (long)O * KH * KW * Ip;
    const float qs = (P <= 2 && q.scale) ? *q.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % Ip);
        long r = idx / Ip;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH);
        const int o = (int)(r / KH);
        const int skh = flip ? KH - 1 - kh : kh, skw = flip ? KW - 1 - kw : kw;
        float v = (i < I) ? w[o * s_o + i * s_i + skh * s_kh + skw * s_kw] : 0.f;
        if (P <= 2) {
            amx = fmaxf(amx, fabsf(v));
            if (!out) continue;
            float qv = v * qs;
            if (fabsf(qv) > SLN_F16_MAX) { qv = copysignf(SLN_F16_MAX, qv); sat = true; }
            const _Float16 h0 = (_Float16)qv;
            out[idx] = __builtin_bit_cast(__bf16, h0);
            if (P == 2) out[total + idx] = __builtin_bit_cast(__bf16, (_Float16)(qv - (float)h0));
            continue;
        }
        for (int pp = 0; pp < P; ++pp) {
            const __bf16 h = (__bf16)v;
            out[(long)pp * total + idx] = h;
            v -= (float)h;
        }
    }
    if (P <= 2) amax_commit(amx, sat, q, s_word);
}

// All stale weight tensors of a step in ONE launch (parts = 2).  A training step re-splits every trainable
// weight twice (forward order and mirrored / transposed for the data gradient): ~250 launches of ~10 us for
// ~1 us of work each.  Here every 64 k-element chunk of every (weight, orientation, layout) entry is a
// block; the entries are described by a device-resident table (sln_split_desc_t).  Same element mapping and
// arithmetic as split_weights_kernel / split_weights_tiled_kernel / split_weights_tiledh_kernel.
__global__ __launch_bounds__(256) void split_weights_batch_kernel(const sln_split_desc_t *__restrict__ descs,
                                                                  const int32_t *__restrict__ chunk_entry,
                                                                  const int64_t *__restrict__ chunk_first,
                                                                  int chunk_elems) {
    __shared__ unsigned s_word[2];
    const sln_split_desc_t d = descs[chunk_entry[blockIdx.x]];
    const long first = chunk_first[blockIdx.x];
    const long last = min(first + (long)chunk_elems, (long)d.total);
    const float *w = d.w;
    __bf16 *out = (__bf16 *)d.out;
    const SplitScale q = {d.q_scale, d.q_amax, d.q_saturated};
    const float qs = q.scale ? *q.scale : 1.f;
    const int KH = d.KH, KW = d.KW, O = d.O, I = d.I, flip = d.flip;
    const int ntap = KH * KW;
    float amx = 0.f;
    bool sat = false;
    for (long idx = first + threadIdx.x; idx < last; idx += 256) {
        int o, i, kh, kw;
        long dst0, dst1;
        if (d.layout == SLN_WEIGHTS_ROWS) {
            i = (int)(idx % d.Ip);
            long r = idx / d.Ip;
            kw = (int)(r % KW); r /= KW;
            kh = (int)(r % KH);
            o = (int)(r / KH);
            dst0 = idx;
            dst1 = idx + d.total;
        } else {
            const bool h = d.layout == SLN_WEIGHTS_TILED256H;
            const int KS = h ? T2H : T2K;
            const long per_part = (long)T2 * KS;
            const int ncc = (I + KS - 1) / KS, nk = ntap * ncc;
            const int pos = (int)(idx % per_part);
            const long blk = idx / per_part;                        // nt * nk + s
            const int st = (int)(blk % nk), nt = (int)(blk / nk);
            int r, jl, tap, cc;
            if (h) {
                r = pos >> 5;
                const int j = pos & 31;
                jl = ((((j >> 3) ^ SLN_SWZW(r))) << 3) | (j & 7);
                fwd256h_stage(st, ntap, ncc, tap, cc);
            } else {
                r = pos >> 4;
                const int j = pos & 15;
                jl = (((j >> 3) ^ ((r >> 3) & 1)) << 3) | (j & 7);
                fwd256_stage(st, ntap, ncc, tap, cc);
            }
            kh = tap / KW; kw = tap - kh * KW;
            o = nt * T2 + r; i = cc * KS + jl;
            dst0 = blk * 2 * per_part + pos;
            dst1 = dst0 + per_part;
        }
        const int skh = flip ? KH - 1 - kh : kh, skw = flip ? KW - 1 - kw : kw;
        const float v = (o < O && i < I) ? w[o * d.s_o + i * d.s_i + skh * d.s_kh + skw * d.s_kw] : 0.f;
        amx = fmaxf(amx, fabsf(v));
        float qv = v * qs;
        if (fabsf(qv) > SLN_F16_MAX) { qv = copysignf(SLN_F16_MAX, qv); sat = true; }
        const _Float16 h0 = (_Float16)qv;
        out[dst0] = __builtin_bit_cast(__bf16, h0);
        if (d.reserved != 1) out[dst1] = __builtin_bit_cast(__bf16, (_Float16)(qv - (float)h0));   // (reserved = 1: ONE part, ABI 11)
    }
    amax_commit(amx, sat, q, s_word);
}

// scale[i] <- the power of two that puts amax[i] near 2^target_log2 (amax[i] == 0: unchanged);
// amax[i] <- 0.  One launch per step over every tensor slot (delayed scaling).
__global__ __launch_bounds__(256) void scale_update_kernel(float *__restrict__ amax, float *__restrict__ scale,
                                                           float *__restrict__ hist, int32_t *__restrict__ cursor,
                                                           int n, long stride, int window, int target_log2,
                                                           const signed char *__restrict__ headroom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = amax[i];
    if (hist && a > 0.f && a < INFINITY) {
        // The scale follows the maximum over the last `window` steps in which the tensor was produced: the
        // range grows at once, a maximum that comes and goes with the batch is kept (the RPN class-logit
        // gradient of a pyramid level whose anchors were not drawn this time is ~1e-13, x230 ... x1e10 below
        // the next step's), and a tensor that really shrank is followed exactly, `window` steps later.
        int c = cursor[i];
        c = (c < 0 || c >= window) ? 0 : c;
        hist[(long)c * stride + i] = a;
        cursor[i] = c + 1 == window ? 0 : c + 1;
        for (int k = 0; k < window; ++k) a = fmaxf(a, hist[(long)k * stride + i]);
    }
    if (a > 0.f && a < INFINITY) {
        int e;
        (void)frexpf(a, &e);                 // a = f * 2^e, f in [0.5, 1)
        int k = target_log2 - e;             // a * 2^k in [2^(target-1), 2^target)
        if (headroom) k -= headroom[i];      // (per-slot extra head room, in bits: gradient roles)
        k = k < -120 ? -120 : (k > 120 ? 120 : k);
        scale[i] = ldexpf(1.f, k);
    }
    amax[i] = 0.f;
}

// ------------------------------------------------------------------- forward
#define SLN_MAX_SEG 4
struct ConvParams {
    const __bf16 *x;      // [P][Min][Cin] parts (Min = N*H*W)
    const __bf16 *w;      // [P][Cout][Ktot]
    const float *scale;   // [Cout] or null (=1)
    const float *shift;   // [Cout] or null (=0)
    const float *residual;  // [M][Cout] or null
    float *y;             // [M][Cout], or null when only the parts are wanted
    __bf16 *yparts;       // [P][M][Cop] or null: the output's own bf16 parts (fused act_split)
    const float *post_scale;  // [Cout] or null: the PARTS (and colsum) hold output * post_scale[c]
    const float *mask;    // [M][Cout] or null: output elements whose mask value is not > 0 become 0
    float *colsum;        // [Cout] or null: += per-channel sums of the output (bias gradient)
    // P = 2 (scaled split-fp16): the operands' scales (device scalars, NULL = 1): the accumulator
    // holds (sx*x) (*) (sw*w), the epilogue multiplies by 1/(sx*sw); yq: scale / amax / saturation
    // record of the output's own parts
    const float *x_scale, *w_scale;
    SplitScale yq;
    // P = 2, fixed-feature epilogue only: operands that exist as PARTS ONLY (a bottleneck's conv outputs have no
    // reader but convolutions, the next shortcut and ReLU masks: their fp32 copy is never written)
    const __bf16 *res_parts;    // [2][M][Cop] parts of the residual (value (h0 + h1) / *res_scale) or null
    const float *res_scale;     // device scalar (NULL = 1)
    const __bf16 *mask_part0;   // [M][Cop] part 0 of the tensor whose ReLU pattern masks the output (h0 > 0) or null
    long x_part_stride, w_part_stride, y_part_stride;
    int Cop;
    int Cin, Cout, KH, KW, sh, sw, dh, dw, pt, pl, relu;
    int M, Ktot, cin_chunks, gm, gn;
    int w_tiled;   // weights in conv_fwd256_kernel's LDS-image order (split_weights_tiled_kernel)
    int tapmode;   // conv_fwd256h_kernel's ROW instances: 1 = a stage per kernel row (taps kw by row shifts), 2 = per kernel column (taps kh)
    int dbg;   // ablation bits, debug sessions only (SLN_CONV_DBG): 1 no DMA in the k-loop, 2 no MFMA, 4 no fragment reads, 8 no wave-group stagger, 16 general epilogue, 4096 activation stages one ahead instead of two (conv_fwd256h_kernel), 8192 eight-channel epilogue without its part stores, 16384 ... without its split arithmetic, 32768 no epilogue at all (conv_fwd256h_kernel), bits 28 / 29 no weight / no activation pieces in the k-loop (conv_fwd256h_kernel), bit 30 activation pieces issued in phase 0 next to the weight pieces (the placement before round 4's last change), bit 20 activation pieces only for the stages of the first tap column (what a per-kernel-row activation stage would save), 131072 eight-channel epilogue without its column constants' loads
    // Up to SLN_MAX_SEG image groups of different sizes share one launch (the GLM's three
    // scales): group s holds segN[s] images of segH x segW, its output rows start at
    // seg_m0[s] and its input pixels at seg_x0[s] of the flat [pixels][C] buffers.
    int nseg;
    int segH[SLN_MAX_SEG], segW[SLN_MAX_SEG], segOH[SLN_MAX_SEG], segOW[SLN_MAX_SEG];
    int seg_m0[SLN_MAX_SEG], seg_x0[SLN_MAX_SEG];
};

template <int P>
__device__ __forceinline__ void mfma_products(const bf16x8 (&a)[P], const bf16x8 (&b)[P], f32x16 &c) {
    // smallest terms first
    if constexpr (P == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    } else if constexpr (P == 1) {   // one scaled fp16 part per operand: one product
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a[0]), __builtin_bit_cast(h16x8, b[0]), c,
                                                   0, 0, 0);
    } else {   // two fp16 parts: the same 16-bit containers, the f16 matrix instruction
        const h16x8 a0 = __builtin_bit_cast(h16x8, a[0]), a1 = __builtin_bit_cast(h16x8, a[P - 1]);
        const h16x8 b0 = __builtin_bit_cast(h16x8, b[0]), b1 = __builtin_bit_cast(h16x8, b[P - 1]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
    }
}

// 1 / (sx * sw): exact for power-of-two scales
__device__ __forceinline__ float operand_unscale(const float *sx, const float *sw) {
    return 1.0f / ((sx ? *sx : 1.f) * (sw ? *sw : 1.f));
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run
    // of tiles so that the tiles sharing an operand panel share an L2 (bijective).
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// One 64-row slab of an output tile, already staged as fp32 in LDS (row stride LD floats):
// thread t owns columns 4*(t % NCOLQ)..+3 of rows t/NCOLQ + RG*q (RG = NTHREADS/NCOLQ row groups).  Per element:
// v = acc*scale[c] + shift[c] (+ residual) (ReLU) (mask) -> y; parts / colsum of v (* post_scale).
template <int P, int NCOLQ, int LD, int NTHREADS>
__device__ __forceinline__ void epilogue_slab(const ConvParams &p, const float *stage, int m_base, int n0,
                                              int t, float *s_colsum, float alpha, float yqs, float &amx,
                                              bool &sat) {
    constexpr int RG = NTHREADS / NCOLQ;      // row groups: thread t owns rows t/NCOLQ + RG*q
    constexpr int NQ = 64 / RG;
    const bool vec_ok = (p.Cout & 3) == 0;
    const int c = n0 + 4 * (t & (NCOLQ - 1));
    if (c < p.Cout) {
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sf[4] = {0.f, 0.f, 0.f, 0.f}, ps_[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < p.Cout) {
                if (p.scale) sc[e] = p.scale[c + e];
                if (p.shift) sf[e] = p.shift[c + e];
                if (p.post_scale) ps_[e] = p.post_scale[c + e];
                if (P <= 2) sc[e] *= alpha;      // exact (power of two): acc * alpha * scale == acc * (alpha*scale)
            }
        // all residual rows of this half are requested before the first store: the
        // loads cannot be moved across the y stores by the compiler (may alias)
        float4 res4[NQ], msk4[NQ];
        if (vec_ok && p.residual) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int m = m_base + (t / NCOLQ) + RG * q;
                res4[q] = m < p.M ? *(const float4 *)(p.residual + (long)m * p.Cout + c)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (vec_ok && p.mask) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int m = m_base + (t / NCOLQ) + RG * q;
                msk4[q] = m < p.M ? *(const float4 *)(p.mask + (long)m * p.Cout + c)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = (t / NCOLQ) + RG * q;
            const int m = m_base + row;
            if (m >= p.M) continue;
            const float4 a4 = *(const float4 *)&stage[row * LD + 4 * (t & (NCOLQ - 1))];
            float v[4] = {a4.x * sc[0] + sf[0], a4.y * sc[1] + sf[1], a4.z * sc[2] + sf[2],
                          a4.w * sc[3] + sf[3]};
            const long o = (long)m * p.Cout + c;
            if (vec_ok) {
                if (p.residual) {
                    const float4 r4 = res4[q];
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.mask) {
                    const float4 k4 = msk4[q];
                    if (!(k4.x > 0.f)) v[0] = 0.f;
                    if (!(k4.y > 0.f)) v[1] = 0.f;
                    if (!(k4.z > 0.f)) v[2] = 0.f;
                    if (!(k4.w > 0.f)) v[3] = 0.f;
                }
                if (p.y) *(float4 *)(p.y + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < p.Cout) {
                        if (p.residual) v[e] += p.residual[o + e];
                        if (p.relu) v[e] = fmaxf(v[e], 0.f);
                        if (p.mask && !(p.mask[o + e] > 0.f)) v[e] = 0.f;
                        if (p.y) p.y[o + e] = v[e];
                    } else {
                        v[e] = 0.f;
                    }
            }
            if (p.post_scale) {
                // rounded to fp32 before the split (no contraction into split4's subtraction):
                // the same value sln_conv_grad_prep_f32 would split
#pragma clang fp contract(off)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * ps_[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += v[e];
            if (p.yparts) {   // fused act_split of the output (Cop % 8 == 0, c % 4 == 0)
                bf16x4 ps[3];
                if (P <= 2) amx = amax4(amx, v);
                sat |= split4<P>(make_float4(v[0], v[1], v[2], v[3]), ps, yqs);
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    *(bf16x4 *)(p.yparts + pp * p.y_part_stride + (long)m * p.Cop + c) = ps[pp];
            }
        }
        if (p.colsum) {   // 8 row groups x 2 halves share a column: combine in LDS first
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (csum[e] != 0.f) atomicAdd(&s_colsum[4 * (t & (NCOLQ - 1)) + e], csum[e]);
        }
    }
}

// The same slab for the fp16 x 2 format whenever rows are whole 16-B groups (Cout % 4 == 0): residual and
// output parts fixed at compile time, the backward extras (mask, post-scale, column sums, no fp32 output)
// as uniform branches, running row pointers instead of 64-bit index math per row, saturation derived once
// from the running amax.  The general epilogue_slab spends ~150 VALU instructions per four outputs on
// per-element feature selects and index arithmetic: the 256x256 tile's epilogue cost 50-70 k cycles with it
// (stamps, tools/conv_stamps.py: the slab calls themselves, not their barriers or the store drain) -- more
// than the whole k-loop of a K = 256 layer; 17-40 k with this one.  Same arithmetic per element as
// epilogue_slab + split4<2> (bit-identical y, parts and column sums up to their summation order).
template <int NCOLQ, int LD, int NTHREADS, int RES, bool PARTS, int NP = 2>
__device__ __forceinline__ void epilogue_slab_f16(const ConvParams &p, const float *stage, int m_base, int n0,
                                                  int t, float *s_colsum, float alpha, float yqs, float &amx) {
    // RES: 0 no residual, 1 fp32 residual, 2 residual from its two fp16 parts ((h0 + h1) / s_res: what the
    // next convolution reads of the same tensor -- a block output that exists as parts only)
    constexpr int RG = NTHREADS / NCOLQ;
    constexpr int NQ = 64 / RG;
    const int c = n0 + 4 * (t & (NCOLQ - 1));
    if (c >= p.Cout) return;
    const int row0 = t / NCOLQ;
    float4 sc = make_float4(alpha, alpha, alpha, alpha), sf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) {
        const float4 s4 = *(const float4 *)(p.scale + c);
        sc = make_float4(s4.x * alpha, s4.y * alpha, s4.z * alpha, s4.w * alpha);
    }
    if (p.shift) sf = *(const float4 *)(p.shift + c);
    const long o0 = (long)(m_base + row0) * p.Cout + c;
    const long ostep = (long)RG * p.Cout;
    const long q0 = (long)(m_base + row0) * p.Cop + c;       // the same element in a [M][Cop] parts row
    const long pstep = (long)RG * p.Cop;
    float4 res4[RES == 1 ? NQ : 1];
    h16x4 rp0[RES == 2 ? NQ : 1], rp1[RES == 2 ? NQ : 1];
    float rinv = 1.f;
    if (RES == 1) {
        const float *r = p.residual + o0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            res4[q] = (m_base + row0 + RG * q) < p.M ? *(const float4 *)r : make_float4(0.f, 0.f, 0.f, 0.f);
            r += ostep;
        }
    }
    if (RES == 2) {
        rinv = 1.0f / (p.res_scale ? *p.res_scale : 1.f);
        const __bf16 *r0 = p.res_parts + q0, *r1 = r0 + p.y_part_stride;
        const h16x4 z4 = {};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const bool ok = (m_base + row0 + RG * q) < p.M;
            rp0[q] = ok ? *(const h16x4 *)r0 : z4;
            rp1[q] = (ok && NP == 2) ? *(const h16x4 *)r1 : z4;      // (NP = 1: the tensor is its one fp16 part)
            r0 += pstep; r1 += pstep;
        }
    }
    // backward extras (uniform per launch): the producer's ReLU mask (its fp32 output, or part 0 of an output
    // that exists as parts only), a scale applied to the parts and column sums only, per-channel sums of what
    // the parts hold (the previous layer's bias gradient)
    // the ReLU pattern's rows as a block too, before the first store (a load may not be moved above a store
    // that might alias: inside the row loop every row waited a full memory round trip for its mask)
    float4 mk4[NQ];
    h16x4 mk16v[NQ];
    const bool has_mk = p.mask != nullptr, has_mk16 = p.mask_part0 != nullptr;
    if (has_mk) {
        const float *mk = p.mask + o0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mk4[q] = (m_base + row0 + RG * q) < p.M ? *(const float4 *)mk : make_float4(0.f, 0.f, 0.f, 0.f);
            mk += ostep;
        }
    }
    if (has_mk16) {
        const __bf16 *mk16 = p.mask_part0 + q0;
        const h16x4 z4 = {};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mk16v[q] = (m_base + row0 + RG * q) < p.M ? *(const h16x4 *)mk16 : z4;
            mk16 += pstep;
        }
    }
    float4 ps4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.post_scale) ps4 = *(const float4 *)(p.post_scale + c);
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    float *yp = p.y ? p.y + o0 : nullptr;
    __bf16 *p0 = nullptr, *p1 = nullptr;
    if (PARTS) {
        p0 = p.yparts + q0;
        p1 = p0 + p.y_part_stride;
    }
    const float *sg = stage + row0 * LD + 4 * (t & (NCOLQ - 1));
    const bool relu = p.relu != 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (m_base + row0 + RG * q < p.M) {
            const float4 a4 = *(const float4 *)(sg + q * RG * LD);
            float v[4] = {a4.x * sc.x + sf.x, a4.y * sc.y + sf.y, a4.z * sc.z + sf.z, a4.w * sc.w + sf.w};
            if (RES == 1) {
                v[0] += res4[q].x; v[1] += res4[q].y; v[2] += res4[q].z; v[3] += res4[q].w;
            }
            if (RES == 2 && NP == 2) {
                const h16x4 a0 = rp0[q], a1 = rp1[q];
                v[0] += ((float)a0.x + (float)a1.x) * rinv;
                v[1] += ((float)a0.y + (float)a1.y) * rinv;
                v[2] += ((float)a0.z + (float)a1.z) * rinv;
                v[3] += ((float)a0.w + (float)a1.w) * rinv;
            }
            if (RES == 2 && NP == 1) {
                const h16x4 a0 = rp0[q];
                v[0] += (float)a0.x * rinv; v[1] += (float)a0.y * rinv;
                v[2] += (float)a0.z * rinv; v[3] += (float)a0.w * rinv;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_mk) {
                const float4 k4 = mk4[q];
                if (!(k4.x > 0.f)) v[0] = 0.f;
                if (!(k4.y > 0.f)) v[1] = 0.f;
                if (!(k4.z > 0.f)) v[2] = 0.f;
                if (!(k4.w > 0.f)) v[3] = 0.f;
            }
            if (has_mk16) {
                const h16x4 k4 = mk16v[q];
                if (!(k4.x > (_Float16)0)) v[0] = 0.f;
                if (!(k4.y > (_Float16)0)) v[1] = 0.f;
                if (!(k4.z > (_Float16)0)) v[2] = 0.f;
                if (!(k4.w > (_Float16)0)) v[3] = 0.f;
            }
            if (yp) *(float4 *)yp = make_float4(v[0], v[1], v[2], v[3]);
            if (p.post_scale) {
                // rounded to fp32 before the split (no contraction into split4's subtraction): the same
                // value sln_conv_grad_prep_f32 would split
#pragma clang fp contract(off)
                v[0] = v[0] * ps4.x; v[1] = v[1] * ps4.y; v[2] = v[2] * ps4.z; v[3] = v[3] * ps4.w;
            }
            if (p.colsum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) csum[e] += v[e];
            }
            if (PARTS) {
                const float rmax = amax4(0.f, v);
                amx = fmaxf(amx, rmax);
                bf16x4 ps[2];
                if (NP == 2 && !(p.dbg & 128) && __builtin_amdgcn_ballot_w64(rmax * yqs > SLN_F16_MAX) == 0)
                    split4_inrange(make_float4(v[0], v[1], v[2], v[3]), ps, yqs);
                else
                    (void)split4<NP>(make_float4(v[0], v[1], v[2], v[3]), ps, yqs);
                *(bf16x4 *)p0 = ps[0];
                if (NP == 2) *(bf16x4 *)p1 = ps[1];
            }
        }
        if (yp) yp += ostep;
        if (PARTS) { p0 += pstep; p1 += pstep; }
    }
    if (p.colsum) {   // the row groups share a column: combine in LDS first
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (csum[e] != 0.f) atomicAdd(&s_colsum[4 * (t & (NCOLQ - 1)) + e], csum[e]);
    }
}

// The slab with EIGHT channels per thread, for outputs that carry parts and whose rows are whole 32-B groups
// (Cout % 8 == 0: every backbone / head layer): a part's store is then 16 B per lane -- half the store
// instructions of the four-channel slab for the same bytes.  These epilogues are bound by the number of memory
// instructions they issue, not by bytes or arithmetic (tools/hbm_layers.py: on the 128^2 kernel a parts-only
// output, 2 x 8-B stores per four elements, took 18 % longer than an fp32-only one, 1 x 16-B store, for the
// same 4 B per element; MI355X_MICROARCH.md "store-ISSUE-bound" tails).  Thread t owns columns
// 8*(t % NCOL8)..+7 of rows t/NCOL8 + RG*q.  Same arithmetic per element as epilogue_slab_f16.
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8_t;

// What a slab reads from global memory besides its accumulators: NQ rows of the residual (RES 1: fp32, 2: its
// two fp16 parts) and of the ReLU pattern (MASK 1: fp32, 2: part 0).  Loaded as a block BEFORE the previous
// slab's stores are issued (the compiler may not move a load above a store that might alias), so that their
// HBM round trip -- 2-4 us under load, once per slab and, for the mask, once per ROW in the first version --
// overlaps the previous slab's arithmetic and stores instead of standing in front of every slab.
template <int NQ, int RES, int MASK>
struct W8Pre {
    float4 ra[RES == 1 ? NQ : 1], rb[RES == 1 ? NQ : 1];
    h16x8_t rp0[RES == 2 ? NQ : 1], rp1[RES == 2 ? NQ : 1];
    float4 ma[MASK == 1 ? NQ : 1], mb[MASK == 1 ? NQ : 1];
    h16x8_t m16[MASK == 2 ? NQ : 1];
};

// Per-tile constants of a thread's eight columns.
struct W8Cols {
    float sc[8], sf[8], ps8[8];
    float rinv;
    float csum[8];      // running column sums of the tile (colsum mode): one LDS atomic per column at the end
};

template <int NCOL8>
__device__ __forceinline__ void w8_cols(const ConvParams &p, int n0, int t, float alpha, W8Cols &k) {
    const int c = n0 + 8 * (t & (NCOL8 - 1));
#pragma unroll
    for (int e = 0; e < 8; ++e) { k.sc[e] = alpha; k.sf[e] = 0.f; k.ps8[e] = 1.f; k.csum[e] = 0.f; }
    k.rinv = 1.f;
    if (c >= p.Cout) return;
    // (round 4: these loads stand at the head of every tile's epilogue, and skipping them -- dbg 131072, wrong
    // results -- made a pointwise launch 6-16 % shorter; but DMA-ing the constants into LDS during the last k-loop
    // stage and reading them from there changed nothing, 0.142 vs 0.139 ms and 87.7 vs 87.6 img/s: what the
    // ablation removed was not their latency.  profiles/r4_i_column_constants.txt)
    if (p.dbg & 131072) return;
    if (p.scale) {
        const float4 s0 = *(const float4 *)(p.scale + c), s1 = *(const float4 *)(p.scale + c + 4);
        k.sc[0] = s0.x * alpha; k.sc[1] = s0.y * alpha; k.sc[2] = s0.z * alpha; k.sc[3] = s0.w * alpha;
        k.sc[4] = s1.x * alpha; k.sc[5] = s1.y * alpha; k.sc[6] = s1.z * alpha; k.sc[7] = s1.w * alpha;
    }
    if (p.shift) {
        const float4 s0 = *(const float4 *)(p.shift + c), s1 = *(const float4 *)(p.shift + c + 4);
        k.sf[0] = s0.x; k.sf[1] = s0.y; k.sf[2] = s0.z; k.sf[3] = s0.w;
        k.sf[4] = s1.x; k.sf[5] = s1.y; k.sf[6] = s1.z; k.sf[7] = s1.w;
    }
    if (p.post_scale) {
        const float4 s0 = *(const float4 *)(p.post_scale + c), s1 = *(const float4 *)(p.post_scale + c + 4);
        k.ps8[0] = s0.x; k.ps8[1] = s0.y; k.ps8[2] = s0.z; k.ps8[3] = s0.w;
        k.ps8[4] = s1.x; k.ps8[5] = s1.y; k.ps8[6] = s1.z; k.ps8[7] = s1.w;
    }
    if (p.res_parts) k.rinv = 1.0f / (p.res_scale ? *p.res_scale : 1.f);
}

// FULL: the tile lies completely inside the output (all but the last row / column tiles of a launch): no per-row or
// per-column predicate, so every load and store is issued unconditionally and the compiler can count them -- with
// the predicates it cannot, and waits for the look-ahead loads with vmcnt(0), i.e. also for every store and for
// the loads of the half AFTER the one it is about to use.
template <int NCOL8, int NTHREADS, int NQ, int RES, int MASK, bool FULL>
__device__ __forceinline__ void w8_load(const ConvParams &p, int m_base, int n0, int t, int q0,
                                        W8Pre<NQ, RES, MASK> &pre) {
    // rows t/NCOL8 + RG*(q0 + q), q = 0..NQ-1, of the 64-row slab at m_base
    constexpr int RG = NTHREADS / NCOL8;
    const int c = n0 + 8 * (t & (NCOL8 - 1));
    if (!FULL && c >= p.Cout) return;
    const int row0 = t / NCOL8 + RG * q0;
    const long o0 = (long)(m_base + row0) * p.Cout + c;      // (Cop == Cout here)
    const long ostep = (long)RG * p.Cout;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const h16x8_t z8 = {};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool ok = FULL || (m_base + row0 + RG * q) < p.M;
        const long o = o0 + q * ostep;
        if (RES == 1) {
            pre.ra[q] = ok ? *(const float4 *)(p.residual + o) : z4;
            pre.rb[q] = ok ? *(const float4 *)(p.residual + o + 4) : z4;
        }
        if (RES == 2) {
            pre.rp0[q] = ok ? *(const h16x8_t *)(p.res_parts + o) : z8;
            pre.rp1[q] = ok ? *(const h16x8_t *)(p.res_parts + p.y_part_stride + o) : z8;
        }
        if (MASK == 1) {
            pre.ma[q] = ok ? *(const float4 *)(p.mask + o) : z4;
            pre.mb[q] = ok ? *(const float4 *)(p.mask + o + 4) : z4;
        }
        if (MASK == 2) pre.m16[q] = ok ? *(const h16x8_t *)(p.mask_part0 + o) : z8;
    }
}

template <int NCOL8, int LD, int NTHREADS, int NQ, int RES, int MASK, bool FULL>
__device__ __forceinline__ void w8_compute(const ConvParams &p, const float *stage, int m_base, int n0, int t, int q0,
                                           float yqs, float &amx, W8Cols &k, const W8Pre<NQ, RES, MASK> &pre) {
    constexpr int RG = NTHREADS / NCOL8;
    const int cg = t & (NCOL8 - 1);
    const int c = n0 + 8 * cg;
    if (!FULL && c >= p.Cout) return;
    const int row0 = t / NCOL8 + RG * q0;
    const long o0 = (long)(m_base + row0) * p.Cout + c;
    const long ostep = (long)RG * p.Cout;
    float *yp = p.y ? p.y + o0 : nullptr;
    __bf16 *p0 = p.yparts + o0, *p1 = p0 + p.y_part_stride;
    const float *sg = stage + row0 * LD + 8 * cg;
    const bool relu = p.relu != 0;
    const bool nt = (p.dbg & 64) != 0;       // experiment: non-temporal output stores (no effect measured)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (FULL || m_base + row0 + RG * q < p.M) {
            const float4 a0 = *(const float4 *)(sg + q * RG * LD), a1 = *(const float4 *)(sg + q * RG * LD + 4);
            float v[8] = {a0.x * k.sc[0] + k.sf[0], a0.y * k.sc[1] + k.sf[1], a0.z * k.sc[2] + k.sf[2],
                          a0.w * k.sc[3] + k.sf[3], a1.x * k.sc[4] + k.sf[4], a1.y * k.sc[5] + k.sf[5],
                          a1.z * k.sc[6] + k.sf[6], a1.w * k.sc[7] + k.sf[7]};
            if (RES == 1) {
                v[0] += pre.ra[q].x; v[1] += pre.ra[q].y; v[2] += pre.ra[q].z; v[3] += pre.ra[q].w;
                v[4] += pre.rb[q].x; v[5] += pre.rb[q].y; v[6] += pre.rb[q].z; v[7] += pre.rb[q].w;
            }
            if (RES == 2) {
                const h16x8_t b0 = pre.rp0[q], b1 = pre.rp1[q];
                v[0] += ((float)b0.s0 + (float)b1.s0) * k.rinv; v[1] += ((float)b0.s1 + (float)b1.s1) * k.rinv;
                v[2] += ((float)b0.s2 + (float)b1.s2) * k.rinv; v[3] += ((float)b0.s3 + (float)b1.s3) * k.rinv;
                v[4] += ((float)b0.s4 + (float)b1.s4) * k.rinv; v[5] += ((float)b0.s5 + (float)b1.s5) * k.rinv;
                v[6] += ((float)b0.s6 + (float)b1.s6) * k.rinv; v[7] += ((float)b0.s7 + (float)b1.s7) * k.rinv;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (MASK == 1) {
                const float4 k0 = pre.ma[q], k1 = pre.mb[q];
                if (!(k0.x > 0.f)) v[0] = 0.f;
                if (!(k0.y > 0.f)) v[1] = 0.f;
                if (!(k0.z > 0.f)) v[2] = 0.f;
                if (!(k0.w > 0.f)) v[3] = 0.f;
                if (!(k1.x > 0.f)) v[4] = 0.f;
                if (!(k1.y > 0.f)) v[5] = 0.f;
                if (!(k1.z > 0.f)) v[6] = 0.f;
                if (!(k1.w > 0.f)) v[7] = 0.f;
            }
            if (MASK == 2) {
                const h16x8_t k8 = pre.m16[q];
                if (!(k8.s0 > (_Float16)0)) v[0] = 0.f;
                if (!(k8.s1 > (_Float16)0)) v[1] = 0.f;
                if (!(k8.s2 > (_Float16)0)) v[2] = 0.f;
                if (!(k8.s3 > (_Float16)0)) v[3] = 0.f;
                if (!(k8.s4 > (_Float16)0)) v[4] = 0.f;
                if (!(k8.s5 > (_Float16)0)) v[5] = 0.f;
                if (!(k8.s6 > (_Float16)0)) v[6] = 0.f;
                if (!(k8.s7 > (_Float16)0)) v[7] = 0.f;
            }
            if (yp) {
                typedef __attribute__((ext_vector_type(4))) float f32x4_t;
                const f32x4_t y0 = {v[0], v[1], v[2], v[3]}, y1 = {v[4], v[5], v[6], v[7]};
                if (nt) {
                    __builtin_nontemporal_store(y0, (f32x4_t *)yp);
                    __builtin_nontemporal_store(y1, (f32x4_t *)(yp + 4));
                } else {
                    *(f32x4_t *)yp = y0;
                    *(f32x4_t *)(yp + 4) = y1;
                }
            }
            if (p.post_scale) {
                // rounded to fp32 before the split (no contraction into split4's subtraction): the same
                // value sln_conv_grad_prep_f32 would split
#pragma clang fp contract(off)
                v[0] = v[0] * k.ps8[0]; v[1] = v[1] * k.ps8[1]; v[2] = v[2] * k.ps8[2]; v[3] = v[3] * k.ps8[3];
                v[4] = v[4] * k.ps8[4]; v[5] = v[5] * k.ps8[5]; v[6] = v[6] * k.ps8[6]; v[7] = v[7] * k.ps8[7];
            }
            if (p.colsum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) k.csum[e] += v[e];
            }
            if (p.yparts && (p.dbg & 16384)) {       // ablation: the part stores without the split arithmetic
                bf16x8 w0 = __builtin_bit_cast(bf16x8, a0), w1 = __builtin_bit_cast(bf16x8, a1);
                *(bf16x8 *)p0 = w0;
                *(bf16x8 *)p1 = w1;
            } else
            if (p.yparts) {                          // (fp32-only outputs: uniform per launch)
            const float rmax = (p.dbg & 512) ? amax4(amax4(0.f, v), v + 4) : amax8(v);      // (dbg 512: A/B)
            amx = fmaxf(amx, rmax);
            bf16x4 lo[2], hi[2];
            if (!(p.dbg & 128) && __builtin_amdgcn_ballot_w64(rmax * yqs > SLN_F16_MAX) == 0) {   // the wave's rows in range (dbg 128: A/B)
                split4_inrange(make_float4(v[0], v[1], v[2], v[3]), lo, yqs);
                split4_inrange(make_float4(v[4], v[5], v[6], v[7]), hi, yqs);
            } else {
                (void)split4<2>(make_float4(v[0], v[1], v[2], v[3]), lo, yqs);
                (void)split4<2>(make_float4(v[4], v[5], v[6], v[7]), hi, yqs);
            }
            bf16x8 w0, w1;
            w0.s0 = lo[0].x; w0.s1 = lo[0].y; w0.s2 = lo[0].z; w0.s3 = lo[0].w;
            w0.s4 = hi[0].x; w0.s5 = hi[0].y; w0.s6 = hi[0].z; w0.s7 = hi[0].w;
            w1.s0 = lo[1].x; w1.s1 = lo[1].y; w1.s2 = lo[1].z; w1.s3 = lo[1].w;
            w1.s4 = hi[1].x; w1.s5 = hi[1].y; w1.s6 = hi[1].z; w1.s7 = hi[1].w;
            if (p.dbg & 8192) {            // ablation: the epilogue without its part stores
                asm volatile("" ::"v"(w0), "v"(w1));
            } else if (nt) {
                __builtin_nontemporal_store(w0, (bf16x8 *)p0);
                __builtin_nontemporal_store(w1, (bf16x8 *)p1);
            } else {
                *(bf16x8 *)p0 = w0;
                *(bf16x8 *)p1 = w1;
            }
            }
        }
        if (yp) yp += ostep;
        p0 += ostep; p1 += ostep;
    }
}

// A whole tile through the eight-channel slabs, NSLAB slabs of 64 rows: slab h's accumulators go to LDS
// (stage_slab(h): the kernel's own C-layout write), and the slab is worked off in two halves of NQ / 2 rows per
// thread; BEFORE a half's arithmetic and stores the global loads of the NEXT half (of this slab or the next)
// are issued (w8_load).  Halves, not whole slabs: two sets of a whole slab's residual + mask rows next to the
// 128 accumulator registers of the wave group whose slabs come last do not fit the 256-register budget.
// (SR: rows per slab, 64 everywhere but in conv_fwd128x256h_kernel, whose 256 threads take 32-row slabs so that a
// half slab's look-ahead set stays two rows per thread)
// (DBOFF > 0: a SECOND staging slab DBOFF floats behind the first -- the kernels that own their CU's LDS in the
// epilogue.  Slab h + 1 is staged (stage_slab(h + 1, buffer)) at the head of slab h's work, into the buffer slab
// h - 1 was read from: one barrier per slab instead of two, and the staging -- which only the waves that hold the
// slab's accumulators do -- runs beside the other waves' arithmetic.)
template <int NCOL8, int LD, int NTHREADS, int NSLAB, int RES, int MASK, bool FULL, typename StageFn, int DEPTH = 1,
          int SR = 64, int DBOFF = 0>
__device__ __forceinline__ void epilogue_tile_w8(const ConvParams &p, const float *stage, int m0, int n0, int t,
                                                 float *s_colsum, float alpha, float yqs, float &amx,
                                                 StageFn stage_slab) {
    constexpr int NQ = SR / (NTHREADS / NCOL8);
    constexpr int NH = NQ >= 2 ? 2 : 1, NQH = NQ / NH;
    constexpr int NSTEP = NSLAB * NH;               // half slabs of the tile, in order
    // (fp32 residual AND fp32 mask -- a block whose input is an ordinary tensor: rare -- would need 64 more
    // registers for the look-ahead set: that instance loads each half right before it is used)
    // D half slabs of residual / mask rows are in flight ahead of the one being worked on
    constexpr int D = (RES == 1 && MASK == 1) ? 0 : ((RES == 0 && MASK == 0) ? 0 : DEPTH);
    W8Cols k;
    w8_cols<NCOL8>(p, n0, t, alpha, k);
    W8Pre<NQH, RES, MASK> pre[D + 1];
#pragma unroll
    for (int i = 0; i < D && i < NSTEP; ++i)
        w8_load<NCOL8, NTHREADS, NQH, RES, MASK, FULL>(p, m0 + (i / NH) * SR, n0, t, (i % NH) * NQH, pre[i % (D + 1)]);
    if constexpr (DBOFF > 0) {
        stage_slab(0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const int h = i / NH, u = i % NH;
        if constexpr (DBOFF > 0) {
            if (u == 0 && h + 1 < NSLAB) stage_slab(h + 1, (h + 1) & 1);
        } else {
            if (u == 0) {
                stage_slab(h);
                __syncthreads();
            }
        }
        if (i + D < NSTEP || D == 0)
            w8_load<NCOL8, NTHREADS, NQH, RES, MASK, FULL>(p, m0 + ((i + D) / NH) * SR, n0, t, ((i + D) % NH) * NQH,
                                                           pre[(i + D) % (D + 1)]);
        w8_compute<NCOL8, LD, NTHREADS, NQH, RES, MASK, FULL>(p, stage + (DBOFF > 0 ? (h & 1) * DBOFF : 0), m0 + h * SR,
                                                              n0, t, u * NQH, yqs, amx, k, pre[i % (D + 1)]);
        if (u == NH - 1) __syncthreads();
    }
    if (p.colsum && n0 + 8 * (t & (NCOL8 - 1)) < p.Cout) {   // the row groups share a column: combine in LDS
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k.csum[e] != 0.f) atomicAdd(&s_colsum[8 * (t & (NCOL8 - 1)) + e], k.csum[e]);
        // (the kernel's own barrier-free read of s_colsum follows a __syncthreads below)
    }
    __syncthreads();
}

// Which (RES, MASK) instance a launch needs: the shortcut from its parts (forward block outputs), the ReLU
// pattern from part 0 (chained data gradients), or neither.
template <int NCOL8, int LD, int NTHREADS, int NSLAB, typename StageFn>
__device__ __forceinline__ void epilogue_tile_w8_any(const ConvParams &p, const float *stage, int m0, int n0, int t,
                                                     float *s_colsum, float alpha, float yqs, float &amx,
                                                     StageFn stage_slab) {
#define SLN_W8(R, K, F) epilogue_tile_w8<NCOL8, LD, NTHREADS, NSLAB, R, K, F>(p, stage, m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab)
    const bool full = m0 + 64 * NSLAB <= p.M && n0 + 8 * NCOL8 <= p.Cout && !(p.dbg & 256);      // block-uniform (dbg 256: A/B)
    if (full) {
        if (p.res_parts) SLN_W8(2, 0, true);           // (launcher: never together with a mask)
        else if (p.residual && p.mask_part0) SLN_W8(1, 2, true);
        else if (p.residual) SLN_W8(1, 0, true);
        else if (p.mask_part0) SLN_W8(0, 2, true);
        else SLN_W8(0, 0, true);
    } else {
        if (p.res_parts) SLN_W8(2, 0, false);
        else if (p.residual && p.mask_part0) SLN_W8(1, 2, false);
        else if (p.residual) SLN_W8(1, 0, false);
        else if (p.mask_part0) SLN_W8(0, 2, false);
        else SLN_W8(0, 0, false);
    }
#undef SLN_W8
}

__host__ __device__ __forceinline__ bool epilogue_is_w8(const ConvParams &p) {
    // (launches that READ an fp32 residual or mask stay on the four-channel slab: eight channels per thread make
    // an fp32 row 32-B pieces at a 32-B stride per instruction -- measured 20-35 % slower for the loads, while
    // an fp32 OUTPUT next to the parts still gains; dbg 32: A/B against the four-channel slab)
    // (round 3, later: with predicate-free full tiles the fp32 residual rows gain too -- dbg 1024 keeps them on
    // the four-channel slab for A/B; an fp32 MASK stays there)
    // (fp32-only outputs -- the plain data gradients of the FPN / RPN convolutions, ~20 large launches per step --
    // run here as well: they were the last users of the all-in-one instance; dbg 2048: A/B)
    return (p.yparts || (p.y && !(p.dbg & 2048))) && (p.Cout & 7) == 0 && (!p.residual || !(p.dbg & 1024)) && !p.mask &&
           !(p.dbg & 32);
}

// Dispatch of a slab: the fixed-feature version whenever the rows are whole 16-B groups.
template <int P, int NCOLQ, int LD, int NTHREADS>
__device__ __forceinline__ void epilogue_any(const ConvParams &p, const float *stage, int m_base, int n0, int t,
                                             float *s_colsum, float alpha, float yqs, float &amx, bool &sat,
                                             bool fast) {
    if (P <= 2 && fast) {
#define SLN_EPI(R, Q) epilogue_slab_f16<NCOLQ, LD, NTHREADS, R, Q, (P <= 2 ? P : 2)>(p, stage, m_base, n0, t, s_colsum, alpha, yqs, amx)
        if (p.res_parts) {
            if (p.yparts) SLN_EPI(2, true); else SLN_EPI(2, false);
        } else if (p.residual) {
            if (p.yparts) SLN_EPI(1, true); else SLN_EPI(1, false);
        } else {
            if (p.yparts) SLN_EPI(0, true); else SLN_EPI(0, false);
        }
#undef SLN_EPI
    } else {
        epilogue_slab<P, NCOLQ, LD, NTHREADS>(p, stage, m_base, n0, t, s_colsum, alpha, yqs, amx, sat);
    }
}

__host__ __device__ __forceinline__ bool epilogue_is_plain(const ConvParams &p) {
    return (p.Cout & 3) == 0 && (!p.yparts || (p.Cop & 3) == 0);
}

// BNT = 128: waves 2x2, each 64x64 (2x2 MFMA tiles).  BNT = 64 (Cout <= 64: the C2 stage): waves 4x1,
// each 32x64 (1x2 tiles) -- half the B tile and half the MFMAs of a 128-wide tile that would be half empty.
// EPI as in conv_fwd256h_kernel: 0 = every epilogue, chosen at run time; 1..5 = only that eight-channel tile
// epilogue (the all-in-one 128-wide instance: 168 VGPRs at three blocks per CU with 144 B of scratch per lane).
// ROW3 (round 5, the small-K 3-wide layers of C2 / C3: 64 -> 64 at 256^2, 128 -> 128 at 128^2): a k-step is (32-channel
// chunk, kernel ROW) instead of (chunk, tap).  The tile's 128 pixels are consecutive pixels of ONE image row (the
// launcher asks for W % 128 == 0, stride 1, pad_left == dil_w <= 8), so the three taps of a kernel row read the same
// 128 + 2 dil_w input pixels shifted by 0 / d / 2d: they are staged ONCE (144 LDS rows, pixels outside the image row
// as zeros) next to the three taps' weight tiles, and one pair of barriers covers three taps of MFMA work -- a third
// of the activation loads, a third of the barriers per product.  The chunk swizzle (row bits 2..3) is conflict-free
// for any 32 consecutive rows, so a shifted fragment read costs the same as an aligned one.  Same products, another
// fp32 summation order (chunk, kh, kw).
template <int P, int BNT, int EPI = 0, bool ROW3 = false>
__global__ __launch_bounds__(256, ROW3 ? (BNT == 128 ? 2 : 3) : (BNT == 128 ? SLN_FWD128_BLOCKS : 1)) void conv_fwd_kernel(const ConvParams p) {
    constexpr int NI = BNT == 128 ? 2 : 1;        // 32-row MFMA tiles per wave along M
    constexpr int NBI = BNT / 64;                 // 64-row staging passes of the B tile
    constexpr int SLD = BNT + 4;                  // staging slab row stride (floats)
    constexpr int AROWS = ROW3 ? BM + 16 : BM;    // activation rows of a stage (ROW3: the halo of 2 x 8 pixels at most)
    constexpr int NT = ROW3 ? 3 : 1;              // weight tiles (taps) of a stage
    constexpr int NAI = ROW3 ? 3 : 2;             // 64-row staging passes of the A tile
    static_assert(!ROW3 || P == 2, "tap-row k-steps: the two-part format");
    // one LDS region: operand tiles during the k-loop, fp32 staging tile in the epilogue
    // Operand rows are 32 bf16 = 64 B, unpadded; the 16-B chunk index is XOR-swizzled
    // with bits 2..3 of the row, so the 16 rows a ds_read_b128 group touches cover
    // all 16 four-bank groups (conflict-free) and a block needs 48 KB -> 3 blocks/CU.
    constexpr int TILE_B = P * (AROWS + NT * BNT) * BK * 2, STAGE_B = 64 * SLD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_B > STAGE_B ? TILE_B : STAGE_B];
    typedef __bf16 (*tile_t)[AROWS][BK];
    typedef __bf16 (*tileb_t)[NT * BNT][BK];
    tile_t sA = (tile_t)smem;
    tileb_t sB = (tileb_t)(smem + P * AROWS * BK * 2);
    __shared__ float s_colsum[BNT];   // per-block column sums of the output (colsum mode)
    __shared__ unsigned s_word[2];    // block amax / saturation flag of the output parts (P = 2)
    if (threadIdx.x < BNT) s_colsum[threadIdx.x] = 0.f;   // ordered by the k-loop's barriers
    const float alpha = P <= 2 ? operand_unscale(p.x_scale, p.w_scale) : 1.f;
    const float yqs = (P <= 2 && p.yq.scale) ? *p.yq.scale : 1.f;
    float amx = 0.f;
    bool sat = false;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = BNT == 128 ? wave >> 1 : wave, wc = BNT == 128 ? wave & 1 : 0;   // wave tile (32*NI) x 64
    const int bid = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (bid / p.gn) * BM;
    const int n0 = (bid % p.gn) * BNT;

    // each thread stages two 16-B chunks per part for A and for B:
    // row = (t>>2) + 64*i (i = 0,1), chunk = t&3 (8 bf16 each)
    const int chunk = (t & 3) * 8;
    int a_ih0[2], a_iw0[2], a_H[2], a_W[2];
    long a_nbase[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (t >> 2) + 64 * i;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int sg = 0;
        for (int q = 1; q < p.nseg; ++q)
            if (mm >= p.seg_m0[q]) sg = q;
        mm -= p.seg_m0[sg];
        const int OHs = p.segOH[sg], OWs = p.segOW[sg];
        a_H[i] = p.segH[sg]; a_W[i] = p.segW[sg];
        const int n = mm / (OHs * OWs);
        const int rem = mm - n * (OHs * OWs);
        const int oh = rem / OWs, ow = rem - oh * OWs;
        a_ih0[i] = oh * p.sh - p.pt;
        a_iw0[i] = ow * p.sw - p.pl;
        a_nbase[i] = (long)p.seg_x0[sg] + (long)n * a_H[i] * a_W[i];
    }
    bool b_ok[NBI];
    const __bf16 *bptr[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int row = (t >> 2) + 64 * i;
        b_ok[i] = (n0 + row) < p.Cout;
        bptr[i] = p.w + (long)(n0 + (b_ok[i] ? row : 0)) * p.Ktot;
    }

    bf16x8 ra[P][NAI], rb[P][NT * NBI];
    const int nk = ROW3 ? p.KH * p.cin_chunks : p.KH * p.KW * p.cin_chunks;
    const bf16x8 zero8 = {};
    // ROW3: the tile's image row (block-uniform)
    const int r3_n = m0 / (p.segOH[0] * p.segOW[0]);
    const int r3_rem = m0 - r3_n * (p.segOH[0] * p.segOW[0]);
    const int r3_oh = r3_rem / p.segOW[0], r3_ow0 = r3_rem - r3_oh * p.segOW[0];

    auto load_tile = [&](int ks) {
        if constexpr (ROW3) {
            const int cc = ks / p.KH, kh = ks - cc * p.KH;
            const int ci = cc * BK + chunk;
            const bool cok = ci < p.Cin;
            const int H = p.segH[0], W = p.segW[0];
            const int ih = r3_oh - p.pt + kh * p.dh;
            const bool rok = cok && ih >= 0 && ih < H;
            const long rbase = ((long)r3_n * H + ih) * W;
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int r = (t >> 2) + 64 * i;
                const int iw = r3_ow0 - p.dw + r;
                const bool ok = rok && r < BM + 2 * p.dw && iw >= 0 && iw < W;
                const long off = (rbase + iw) * p.Cin + ci;
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    ra[pp][i] = ok ? *(const bf16x8 *)(p.x + pp * p.x_part_stride + off) : zero8;
            }
#pragma unroll
            for (int kw = 0; kw < NT; ++kw) {
                const long koff = (long)(kh * 3 + kw) * p.Cin + ci;
#pragma unroll
                for (int i = 0; i < NBI; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        rb[pp][kw * NBI + i] = (b_ok[i] && cok) ? *(const bf16x8 *)(bptr[i] + pp * p.w_part_stride + koff)
                                                                : zero8;
            }
            return;
        }
        // channel-chunk major, tap minor: the KH*KW shifted windows of one 32-channel
        // slab are read in consecutive k-steps, so the re-reads hit L1/L2
        // Order: channel-chunk PAIR major, tap, then the two 32-channel chunks of the
        // pair (= one 128-B line of a pixel's bf16 row): the second chunk's loads hit
        // the L1 lines the first one fetched, and the KH*KW shifted windows of a pair
        // are read in consecutive steps (L2 hits).
        const int ntap = p.KH * p.KW;
        const int pair = ks / (2 * ntap);
        const int rem2 = ks - pair * 2 * ntap;
        const int npair = min(2, p.cin_chunks - 2 * pair);   // last pair may hold one chunk
        const int tap = rem2 / npair;
        const int cc = 2 * pair + (rem2 - tap * npair);
        const int ci0 = cc * BK;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int ci = ci0 + chunk;
        const bool cok = ci < p.Cin;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ih = a_ih0[i] + kh * p.dh, iw = a_iw0[i] + kw * p.dw;
            const bool ok = a_ok[i] && cok && ih >= 0 && ih < a_H[i] && iw >= 0 && iw < a_W[i];
            const long off = (a_nbase[i] + (long)ih * a_W[i] + iw) * p.Cin + ci;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
                ra[pp][i] = ok ? *(const bf16x8 *)(p.x + pp * p.x_part_stride + off) : zero8;
        }
        const long koff = (long)tap * p.Cin + ci;
#pragma unroll
        for (int i = 0; i < NBI; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
                rb[pp][i] = (b_ok[i] && cok) ? *(const bf16x8 *)(bptr[i] + pp * p.w_part_stride + koff)
                                             : zero8;
    };
    const int schunk = ((t & 3) ^ ((t >> 4) & 3)) * 8;   // swizzled chunk of this thread's rows
    auto store_tile = [&]() {
        if constexpr (ROW3) {
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int row = (t >> 2) + 64 * i;
                if (row < AROWS) {
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) *(bf16x8 *)&sA[pp][row][schunk] = ra[pp][i];
                }
            }
#pragma unroll
            for (int kw = 0; kw < NT; ++kw)
#pragma unroll
                for (int i = 0; i < NBI; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        *(bf16x8 *)&sB[pp][kw * BNT + (t >> 2) + 64 * i][schunk] = rb[pp][kw * NBI + i];
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (t >> 2) + 64 * i;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                *(bf16x8 *)&sA[pp][row][schunk] = ra[pp][i];
                if (i < NBI) *(bf16x8 *)&sB[pp][row][schunk] = rb[pp][i < NBI ? i : 0];
            }
        }
    };

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile();
    __syncthreads();

    const int frow = lane & 31, fsw = (frow >> 2) & 3, fhi = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_tile(ks + 1);
        if constexpr (ROW3) {
#pragma unroll
            for (int kw = 0; kw < NT; ++kw) {
                const int ashift = kw * p.dw;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 a[NI][P], b[2][P];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp) {
                            if (i < NI) {
                                const int arow = wr * (32 * NI) + i * 32 + frow + ashift;
                                a[i < NI ? i : 0][pp] = *(const bf16x8 *)&sA[pp][arow][((kk * 2 + fhi) ^ ((arow >> 2) & 3)) * 8];
                            }
                            b[i][pp] = *(const bf16x8 *)&sB[pp][kw * BNT + wc * 64 + i * 32 + frow][((kk * 2 + fhi) ^ fsw) * 8];
                        }
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[i][j]);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[NI][P], b[2][P];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    if (i < NI)
                        a[i < NI ? i : 0][pp] =
                            *(const bf16x8 *)&sA[pp][wr * (32 * NI) + i * 32 + frow][((kk * 2 + fhi) ^ fsw) * 8];
                    b[i][pp] = *(const bf16x8 *)&sB[pp][wc * 64 + i * 32 + frow][((kk * 2 + fhi) ^ fsw) * 8];
                }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[i][j]);
        }
        }
        __syncthreads();
        if (ks + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue through LDS: coalesced 16-B stores of y (and of its bf16 parts) ----
    // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The 128x128
    // fp32 tile is staged in two 64-row halves ([64][132] floats, reusing the operand
    // tiles), read back row-major: thread t owns columns 4*(t&31)..+3 of rows
    // (t>>5) + 8q.  Per element: v = acc*scale[c] + shift[c] (+ residual) (ReLU).
    float(*stage)[SLD] = (float(*)[SLD])smem;
    const bool plain = P <= 2 && epilogue_is_plain(p) && !(p.dbg & 16);
    auto stage_slab = [&](int h) {
        // rows 64h .. 64h+63 of the tile: waves wr == h (BNT 128, 64 rows each) or wr>>1 == h (BNT 64, 32 each)
        if ((BNT == 128 ? wr : wr >> 1) == h) {
            const int rbase = BNT == 128 ? 0 : (wr & 1) * 32;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wc * 64 + j * 32 + (lane & 31)] =
                            acc[i][j][r];
        }
    };
    if (EPI > 0) {
        const bool full = m0 + BM <= p.M && n0 + BNT <= p.Cout;          // block-uniform
#define SLN_W8E(R, K, F) epilogue_tile_w8<BNT / 8, SLD, 256, 2, R, K, F>(p, &stage[0][0], m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab)
        if (full) {
            if (EPI == 2) SLN_W8E(2, 0, true); else if (EPI == 3) SLN_W8E(0, 2, true);
            else if (EPI == 4) SLN_W8E(1, 2, true); else if (EPI == 5) SLN_W8E(1, 0, true); else SLN_W8E(0, 0, true);
        } else {
            if (EPI == 2) SLN_W8E(2, 0, false); else if (EPI == 3) SLN_W8E(0, 2, false);
            else if (EPI == 4) SLN_W8E(1, 2, false); else if (EPI == 5) SLN_W8E(1, 0, false); else SLN_W8E(0, 0, false);
        }
#undef SLN_W8E
    } else if (P == 2 && plain && epilogue_is_w8(p)) {
        epilogue_tile_w8_any<BNT / 8, SLD, 256, 2>(p, &stage[0][0], m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            stage_slab(h);
            __syncthreads();
            epilogue_any<P, BNT / 4, SLD, 256>(p, &stage[0][0], m0 + h * 64, n0, t, s_colsum, alpha, yqs, amx, sat, plain);
            __syncthreads();
        }
    }
    if (p.colsum && t < BNT && n0 + t < p.Cout && s_colsum[t] != 0.f)   // one global atomic per column
        atomicAdd(p.colsum + n0 + t, s_colsum[t]);
    if (P <= 2 && plain) sat = amx * yqs > SLN_F16_MAX;
    if (P <= 2 && p.yparts) amax_commit(amx, sat, p.yq, s_word);
}

// ---------------------------------------------------------------- 256x256 forward tile
// The same GEMM on a 256x256 output tile: 8 waves (2 x 4), each 128x64 = 4x2 MFMA tiles, so a
// 16-deep k-step needs 18 fragment reads for 48 MFMAs (the 128^2 kernel: 12 for 24).  Operand
// tiles go global -> LDS by DMA (global_load_lds_dwordx4: lane l's 16 B land at base + 16 l, the
// SOURCE address is per lane), three 48-KB stages deep, one barrier per stage, the DMA of stage
// s+2 in flight across it (counted s_waitcnt vmcnt, raw s_barrier).  LDS rows are 16 bf16 = 32 B;
// the 16-B half is XOR-swizzled with bit 3 of the row, applied on the source side, so that the
// 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank groups.  Taps outside the image and
// channel / row tails read a 16-B page of zeros.  One block per CU (144 KB LDS).
__device__ __attribute__((aligned(16))) const unsigned char sln_zero_page[16] = {0};

template <int P>
__global__ __launch_bounds__(512) void conv_fwd256_kernel(const ConvParams p) {
    constexpr int REGION = T2 * T2K * 2;      // one part of one operand: 256 rows x 32 B = 8 KB
    constexpr int STAGE = 2 * P * REGION;     // A parts then B parts
    // ONE LDS object: with a second __shared__ array beside the DMA staging buffer hipcc (ROCm 7.2)
    // emits s_waitcnt vmcnt(0) before the first ds_read of every k-step, draining the DMA pipeline
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE + T2 * 4 + 16];
    float *s_colsum = (float *)(smem + 3 * STAGE);
    unsigned *s_word = (unsigned *)(smem + 3 * STAGE + T2 * 4);
    const float alpha = P == 2 ? operand_unscale(p.x_scale, p.w_scale) : 1.f;   // (scalar loads: not on vmcnt)
    const float yqs = (P == 2 && p.yq.scale) ? *p.yq.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    if (t < T2) s_colsum[t] = 0.f;
    const int bid = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (bid / p.gn) * T2;
    const int n0 = (bid % p.gn) * T2;

    // ---- DMA slot of this lane: row 32*wave + lane/2 of every region, 16-B half lane&1 ----
    const int drow = 32 * wave + (lane >> 1);
    const int dlog = ((lane & 1) ^ ((drow >> 3) & 1)) * 8;   // logical channel offset it fetches
    int a_ih0, a_iw0, a_H, a_W;
    long a_nbase;
    bool a_ok;
    {
        const int m = m0 + drow;
        a_ok = m < p.M;
        int mm = a_ok ? m : 0;
        int sg = 0;
        for (int q = 1; q < p.nseg; ++q)
            if (mm >= p.seg_m0[q]) sg = q;
        mm -= p.seg_m0[sg];
        const int OHs = p.segOH[sg], OWs = p.segOW[sg];
        a_H = p.segH[sg]; a_W = p.segW[sg];
        const int n = mm / (OHs * OWs);
        const int rem = mm - n * (OHs * OWs);
        const int oh = rem / OWs, ow = rem - oh * OWs;
        a_ih0 = oh * p.sh - p.pt;
        a_iw0 = ow * p.sw - p.pl;
        a_nbase = (long)p.seg_x0[sg] + (long)n * a_H * a_W;
    }
    const int ncc = (p.Cin + T2K - 1) / T2K;
    const int nk = p.KH * p.KW * ncc;
    // weights: tiled layout, this lane's 16 B of every (stage, part) block: contiguous 1-KiB pieces
    const __bf16 *bptr = p.w + (long)(bid % p.gn) * nk * P * (T2 * T2K) + wave * 512 + lane * 8;
    const int ntap = p.KH * p.KW;
    auto issue = [&](int s) {
        int tap, cc;
        fwd256_stage(s, ntap, ncc, tap, cc);
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int ci = cc * T2K + dlog;
        const bool cok = ci < p.Cin;
        const int ih = a_ih0 + kh * p.dh, iw = a_iw0 + kw * p.dw;
        const bool aok = a_ok && cok && ih >= 0 && ih < a_H && iw >= 0 && iw < a_W;
        const long aoff = (a_nbase + (long)ih * a_W + iw) * p.Cin + ci;
        const __bf16 *gbs = bptr + (long)s * P * (T2 * T2K);
        unsigned char *base = smem + (s % 3) * STAGE + wave * 1024;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const void *ga = aok ? (const void *)(p.x + pp * p.x_part_stride + aoff) : (const void *)sln_zero_page;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(base + pp * REGION), 16, 0, 0);
        }
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
            __builtin_amdgcn_global_load_lds((glb_void *)(gbs + pp * (T2 * T2K)),
                                             (lds_void *)(base + (P + pp) * REGION), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses (bytes inside a region): row*32 + swizzled half*16
    const int frow = lane & 31, fhalf = lane >> 5;
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 128 * wr + 32 * i + frow;
        a_off[i] = row * 32 + ((fhalf ^ ((row >> 3) & 1)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 64 * wc + 32 * j + frow;
        b_off[j] = row * 32 + ((fhalf ^ ((row >> 3) & 1)) * 16);
    }

    issue(0);
    if (nk > 1) issue(1);
    for (int s = 0; s < nk; ++s) {
        // this wave's DMAs of stage s have landed (those of stage s+1 may still be in flight) ...
        if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everyone's; everyone has also finished reading stage s-1's buffer,
        // which the DMA of stage s+2 overwrites
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char *st = smem + (s % 3) * STAGE;
        bf16x8 a[4][P], b[2][P];
        if (!(p.dbg & 4) || s == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) b[j][pp] = *(const bf16x8 *)(st + (P + pp) * REGION + b_off[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) a[i][pp] = *(const bf16x8 *)(st + pp * REGION + a_off[i]);
        }
        // fragment reads first, then the first quarter of the MFMAs, and only then the DMA of
        // stage s+2 (its buffer is free since the barrier): the DMA issue no longer delays the
        // reads this stage's MFMAs wait for
        if (!(p.dbg & 2)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma_products<P>(a[0], b[j], acc[0][j]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) asm volatile("" ::"v"(a[i][pp]));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) asm volatile("" ::"v"(b[j][pp]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nk && !((p.dbg & 1) && s > 0)) issue(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2)) {
#pragma unroll
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[i][j]);
        }
    }
    __syncthreads();

    // ---- epilogue: four 64-row slabs through LDS ([64][260] floats), as in the 128^2 kernel ----
    float *stage = (float *)smem;
    static_assert(64 * 260 * 4 <= 3 * STAGE, "staging slab must fit");
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (wr == (h >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + wc * 64 + j * 32 +
                              (lane & 31)] = acc[2 * (h & 1) + ii][j][r];
        }
        __syncthreads();
        epilogue_slab<P, 64, 260, 512>(p, stage, m0 + h * 64, n0, t, s_colsum, alpha, yqs, amx, sat);
        __syncthreads();
    }
    if (p.colsum && t < T2 && n0 + t < p.Cout && s_colsum[t] != 0.f) atomicAdd(p.colsum + n0 + t, s_colsum[t]);
    if (P == 2 && p.yparts) amax_commit(amx, sat, p.yq, s_word);
}

// ---------------------------------------------------------------- 256x256 forward tile, fp16 x 2
// conv_fwd256_kernel rebuilt around what limits it with three part products per fp32 product
// (tools/conv_dbg.py ablations, tools/micro/dma_patterns.hip):
//  * the LDS-DMA path costs ~2.2 cycles per distinct 128-B line a 1-KiB piece touches (L2-hot): a
//    16-channel stage row is 32 B, i.e. 32 lines per piece = 70 cycles, and the k-loop was DMA-bound
//    (DMA alone 1.9 ms, MFMA alone 2.0 ms, together 4.0 ms on the 256^2 FPN layer).  Here a stage is
//    32 channels (64-B rows, 16 rows per piece: 35 cycles) and the weights come pre-arranged in the
//    LDS image order (contiguous pieces: 19 cycles);
//  * all eight waves used to read fragments, issue their DMA pieces and run their MFMAs at the same
//    time, so that DMA issue (which blocks a wave while the queue is full) and MFMA never overlapped.
//    The two wave groups (waves 0-3 / 4-7: one of each per SIMD) now run half a phase apart, the
//    guide's 8-phase arrangement: while one group's 12-MFMA cluster owns the matrix pipes the other
//    reads its next fragments and issues DMA.
// A stage = 4 phases of (fragment reads [+ DMA of the next stage] | barrier | 12 MFMAs | barrier); two
// 64-KB stage buffers; the DMA of stage s+1 is issued during phases 0..2 of stage s (its buffer was
// last read in stage s-1: every wave has waited for those reads before the barrier that precedes the
// first issue), and every wave waits for its own pieces (vmcnt(0)) before the middle barrier of
// phase 3, which the first reads of stage s+1 lie behind for both groups.
// Diagnostic build (STAMP = true, SLN_CONV_STAMP knob): s_memtime stamps around the four segments of
// every phase, summed per wave in scalar registers and stored after the loop -- read the SHARES, not
// the run time (the stamps' fences forbid overlaps the real kernel has).
__device__ unsigned long long sln_stamp_sums[8 * 16];     // [wave][phase * 4 + segment], block 0 only
#define SLN_STAMP(var)                                                                        \
    do {                                                                                      \
        if (STAMP) {                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");       \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)

// MS = 16: the same tile on v_mfma_f32_16x16x32_f16 (8 x 4 tiles of 16 x 16 per wave, a whole 32-channel stage
// per instruction) instead of 32x32x16 (4 x 2 tiles, two k16 sub-steps): equal MFMA cycles per FLOP and the same
// 24 fragment reads per stage, but under an MFMA-dense loop the chip holds a higher clock on the 16x16 shape
// (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x the FLOP/s on bf16 loops) -- kept only if faster by
// wall time on the train step (SLN_CONV_MFMA16 knob, default decided in sln_conv2d_fwd_ms_f32).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ __forceinline__ void mfma16_products(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x4v &c) {
    const h16x8 a0 = __builtin_bit_cast(h16x8, a[0]), a1 = __builtin_bit_cast(h16x8, a[1]);
    const h16x8 b0 = __builtin_bit_cast(h16x8, b[0]), b1 = __builtin_bit_cast(h16x8, b[1]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);      // smallest terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
}

// EPI: which epilogue the instance carries.  0 = all of them, chosen at run time (diagnostic builds, odd shapes);
// 1 / 2 / 3 / 4 / 5 = ONLY the eight-channel tile epilogue without operands / with the shortcut from its parts / with
// the ReLU pattern from part 0 / fp32 residual + that pattern / fp32 residual -- the launcher knows which a launch needs.  One variant per instance instead of
// seven: the all-in-one instance spills 1600 scalar registers into vector lanes (26 VGPRs, a v_readlane per use).
// ROW (round 4, "tap-row stages"): 3-wide kernels with stride 1 on maps of 32 / 64 / 128 / 256 columns.  The plain k-loop
// DMAs a tile's 256 activation rows once per TAP; the three taps of one kernel ROW read the same input rows shifted by
// 0 / d / 2d pixels.  Here an activation stage is (32-channel chunk, kernel row kh) and serves THREE weight stages.
// Its LDS image is the tile's 256 / W whole image rows with EIGHT ZERO ROWS in front of each and behind the last
// (the DMA's range check writes them): output pixel (k, ow) of the tile sits at LDS row k (W + 8) + 8 + ow, tap kw
// reads row + (kw - 1) d, and a tap that leaves its image row lands in a gap -- no mask, no address select (the
// three earlier builds of this idea lost to exactly that arithmetic).  Gaps of 8 rows keep every
// row congruent to its pixel column modulo 8, so the chunk swizzle of a fragment read depends on (lane, kw) only:
// address = [lane and kw part, one register per stage] + [tile part, a scalar], one add per read like the plain loop.
// Two activation buffers of 336 rows (the next stage has three weight stages to land), the weight ring unchanged.
// Stage order (64-channel group, kh, half, kw) instead of (group, tap, half): same products, another fp32 summation
// order.  Launcher: one image group, KW == 3, stride 1, pad_left == dil_w <= 8, OW == W in {32, 64, 128, 256},
// M % 256 == 0.
template <bool STAMP, int NPH, int MS = 32, int EPI = 0, bool ROW = false>
__global__ __launch_bounds__(512) void conv_fwd256h_kernel(const ConvParams p) {
    constexpr int P = 2;
    static_assert(MS == 32 || (MS == 16 && NPH == 2), "the 16x16x32 body is written for two phases per stage");
    static_assert(!ROW || (MS == 16 && !STAMP), "tap-row stages: the 16x16x32 body, no stamps");
    unsigned long long sums[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    constexpr int REGION = T2 * T2H * 2;      // one part of one operand: 256 rows x 64 B = 16 KB
    constexpr int OBUF = P * REGION;          // one stage of ONE operand (its two parts): 32 KB
    // Round 4: the ACTIVATION stages are prefetched TWO stages ahead (three 32-KB buffers), the weight stages one
    // (two buffers): a stage of a pointwise layer is 1.5 k cycles of MFMA work against a 4-5 k-cycle HBM round
    // trip, so with one stage in flight the k-loop of the HBM-bound launches ran at the pace of the round trips
    // (K = 1024 -> 256: 172-187 k cycles for 48 k of MFMA work); the weights of a Cout tile are served by the L2
    // and stay one stage ahead.  3 + 2 buffers = 160 KB, the whole LDS of a CU: the column sums and the amax word
    // live in the (by then idle) weight ring during the epilogue.  dbg 4096: one stage ahead, as before (A/B).
    constexpr int BBASE = 3 * OBUF;
    __shared__ __attribute__((aligned(16))) unsigned char smem[5 * OBUF];   // ONE LDS object: 163 840 B
    // epilogue: two staging slabs of [64][260] floats (133 120 B), then the column sums and the amax word
    constexpr int SLAB_FLOATS = 64 * 260;
    static_assert(2 * SLAB_FLOATS * 4 + T2 * 4 + 16 <= 5 * OBUF, "two staging slabs + column sums must fit");
    float *s_colsum = (float *)(smem + 2 * SLAB_FLOATS * 4);
    unsigned *s_word = (unsigned *)(smem + 2 * SLAB_FLOATS * 4 + T2 * 4);
    const float alpha = operand_unscale(p.x_scale, p.w_scale);
    const float yqs = p.yq.scale ? *p.yq.scale : 1.f;
    float amx = 0.f;
    bool sat = false;
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned long long ts_start = 0, ts_loop0 = 0, ts_loop1 = 0, ts_end = 0;   // STAMP: whole-tile segments
    SLN_STAMP(ts_start);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (bid / p.gn) * T2;
    const int n0 = (bid % p.gn) * T2;

    // ---- DMA slots of this lane: rows 32*wave + 16*q + lane/4 (q = 0, 1), 16-B chunk lane&3 of the
    //      64-B row; the chunk it FETCHES is the swizzled one.  Pieces are buffer loads to LDS: a lane
    //      whose tap falls outside the image (or whose row / channel is a tail) gets the offset
    //      0xFFFFFFFF, which the buffer range check turns into zeros in LDS -- no per-lane pointer
    //      select, no branches; per stage a lane computes two 32-bit offsets (q = 0, 1), the parts
    //      differ by a scalar offset. ----
    int a_ih0[2], a_iw0[2], a_H[2], a_W[2];
    unsigned a_base[2], a_c0[2];      // byte offset of (image base pixel, channel 0) / of the lane's chunk
    bool a_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int drow = 32 * wave + 16 * q + (lane >> 2);
        a_c0[q] = (unsigned)(((lane & 3) ^ SLN_SWZH(drow)) * 8);       // logical channel offset in the chunk
        const int m = m0 + drow;
        a_ok[q] = m < p.M;
        int mm = a_ok[q] ? m : 0;
        int sg = 0;
        for (int k = 1; k < p.nseg; ++k)
            if (mm >= p.seg_m0[k]) sg = k;
        mm -= p.seg_m0[sg];
        const int OHs = p.segOH[sg], OWs = p.segOW[sg];
        a_H[q] = p.segH[sg]; a_W[q] = p.segW[sg];
        const int n = mm / (OHs * OWs);
        const int rem = mm - n * (OHs * OWs);
        const int oh = rem / OWs, ow = rem - oh * OWs;
        a_ih0[q] = oh * p.sh - p.pt;
        a_iw0[q] = ow * p.sw - p.pl;
        a_base[q] = (unsigned)(p.seg_x0[sg] + n * a_H[q] * a_W[q]);      // pixels (< 2^31 / Cin: launcher)
    }
    const int ncc = (p.Cin + T2H - 1) / T2H;
    const int ntap = p.KH * p.KW;
    const int nk = ntap * ncc;
    const int gfull = ncc / 2;
    // one buffer resource per part (each part < 4 GiB: the launcher checks)
    const unsigned x_part_bytes = (unsigned)(p.x_part_stride * 2);
    const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)x_part_bytes,
                                                                             0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + p.x_part_stride), 0,
                                                                             (int)x_part_bytes, 0x00020000);
    // weights: the tiled image of this Cout tile; this lane's 16 B of the wave's 2-KiB slice of a block
    const unsigned w_tile_bytes = (unsigned)nk * P * (T2 * T2H * 2);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.w + (long)(bid % p.gn) * nk * P * (T2 * T2H)), 0, (int)w_tile_bytes, 0x00020000);
    const unsigned b_voff = (unsigned)(wave * 2048 + lane * 16);

    // the stage whose DMA is being issued: (group of 64 channels, tap, half) walked incrementally
    int n_g = 0, n_tap = 0, n_half = 0, n_kh = 0, n_kw = 0, n_s = 0;
    unsigned a_voff[2];
    auto stage_offsets = [&]() {          // per-lane byte offsets of the two A rows for stage n_s
        const int cc = 2 * n_g + n_half;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ih = a_ih0[q] + n_kh * p.dh, iw = a_iw0[q] + n_kw * p.dw;
            const unsigned ci = (unsigned)(cc * T2H) + a_c0[q];
            const bool ok = a_ok[q] && (unsigned)ih < (unsigned)a_H[q] && (unsigned)iw < (unsigned)a_W[q] &&
                            ci < (unsigned)p.Cin;
            const unsigned off = ((a_base[q] + (unsigned)(ih * a_W[q] + iw)) * (unsigned)p.Cin + ci) * 2u;
            a_voff[q] = ok ? off : 0xFFFFFFFFu;
        }
    };
    auto stage_advance = [&]() {
        ++n_s;
        if (++n_half == (n_g < gfull ? 2 : 1)) {
            n_half = 0;
            ++n_tap;
            if (++n_kw == p.KW) { n_kw = 0; ++n_kh; }
            if (n_tap == ntap) { n_tap = 0; n_kh = 0; n_kw = 0; ++n_g; }
        }
    };
    // piece g (0..7): g&3 -> {part 0 rows q=0, part 0 q=1, part 1 q=0, part 1 q=1}; g < 4: of the ACTIVATION stage
    // n_s (whose offsets a_voff hold) into activation buffer a_wr; g >= 4: of the WEIGHT stage bs into weight
    // buffer bs & 1.
    const int AD = (p.dbg & 4096) ? 1 : 2;    // stages the activations run ahead of the one being multiplied
    int a_wr = 0, a_rd = 0;                   // activation ring: buffer being filled / being read (0..2)
    auto issue_piece = [&](int g, int bs) {
        const int pp = (g >> 1) & 1, q = g & 1;
        if (g < 4) {
            unsigned char *base = smem + a_wr * OBUF + wave * 2048;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? rsrc_a1 : rsrc_a0,
                                                     (lds_void *)(base + pp * REGION + q * 1024), 16, a_voff[q], 0,
                                                     0, 0);
        } else {
            unsigned char *base = smem + BBASE + (bs & 1) * OBUF + wave * 2048;
            const unsigned soff = (unsigned)(bs * P + pp) * (T2 * T2H * 2) + q * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(base + pp * REGION + q * 1024), 16,
                                                     b_voff, soff, 0, 0);
        }
    };
    auto ring_next = [](int &r) { r = r == 2 ? 0 : r + 1; };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses (bytes inside a region) of k16 sub-step 0: row*64 + ((half) ^ swz(row))*16;
    // sub-step 1 flips bit 1 of the chunk index: address ^ 32
    const int frow = lane & 31, fhalf = lane >> 5;
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 128 * wr + 32 * i + frow;
        a_off[i] = row * 64 + ((fhalf ^ SLN_SWZH(row)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 64 * wc + 32 * j + frow;
        b_off[j] = row * 64 + ((fhalf ^ SLN_SWZW(row)) * 16);
    }
    // MS = 16: lane l holds row l % 16 of a 16-row tile and the 16-B chunk l / 16 of its 64-B stage row (the
    // whole 32-channel stage is one instruction's K); 8 row tiles of A, 4 of B per wave
    f32x4v acc16[8][4];
    int a_off16[8], b_off16[4];
    if (MS == 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const int r16 = lane & 15, c16 = lane >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 128 * wr + 16 * i + r16;
            a_off16[i] = row * 64 + ((c16 ^ SLN_SWZH(row)) * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 64 * wc + 16 * j + r16;
            b_off16[j] = row * 64 + ((c16 ^ SLN_SWZW(row)) * 16);
        }
    }

    // (Round 4, tried and removed: pulling this tile's shortcut / ReLU-mask rows towards the L2 / Infinity Cache
    // here -- one 4-byte LDS-DMA load per 128-B line -- so that the epilogue would read them from cache between
    // its stores.  Same box A/B: the train step 88.0 -> 86.0 img/s, the K = 256 expand layer with a parts shortcut
    // 0.188 -> 0.221 ms: the extra reads compete with the k-loop's stages and much of what they fetch is gone
    // again before the epilogue asks for it.  profiles/r4_g_prefetch_epilogue_inputs.txt)
    if constexpr (ROW) {
        constexpr int RROWS = 336;                  // LDS rows of an activation stage: 256 + 8 (256 / W + 1) <= 328, in whole pieces
        constexpr int RREG = RROWS * 64;            // one part of it: 21 504 B
        constexpr int RBUF = P * RREG;              // 43 008 B
        constexpr int RB = 2 * RBUF;                // the weight ring behind the two activation buffers
        static_assert(RB + 2 * OBUF <= 5 * OBUF && RB % 16 == 0, "tap-row stages: LDS budget");
        // (tapmode 2, maps of exactly one tile -- the mask head's 16 x 16 rois: the stage is a kernel COLUMN, its three
        // taps kh read rows shifted by dh * W = one image row; the image has dh * W zero rows above and below it in
        // LDS and no gaps inside, the horizontal tap is the DMA's bounds check as in the plain loop)
        const bool col = p.tapmode == 2;
        const int Wm = p.segW[0], Hm = p.segH[0], OHW = p.segOH[0] * p.segOW[0];
        const int sft = col ? p.dh * Wm : p.dw;                     // LDS rows between consecutive taps of a stage
        const int lead = col ? sft : 8;                             // zero rows in front of the first pixel row
        const int W8 = Wm + 8, R = T2 / Wm;
        const int NPC = ((col ? T2 + 2 * sft : T2 + 8 * R + 8) + 15) >> 4;      // 16-row pieces per part: 17 ... 21
        const bool has3 = wave + 16 < NPC;                         // this wave DMAs a third piece (wave-uniform)
        // ---- DMA slots: piece wave + 8 q (q = 0, 1, 2), LDS row j = 16 piece + lane / 4 ----
        int r_ih0[3], r_iw[3];
        unsigned r_base[3], r_c0[3], r_voff[3];
        bool r_ok[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = 16 * (wave + 8 * q) + (lane >> 2);
            r_c0[q] = (unsigned)(((lane & 3) ^ SLN_SWZH(j)) * 8);
            const int k = col ? 0 : j / W8, pos = col ? j - lead : j - k * W8 - 8;
            const int m = m0 + k * Wm + pos;
            r_ok[q] = pos >= 0 && (col ? pos < T2 : k < R) && (q < 2 || has3);
            const int mm = r_ok[q] ? m : 0;
            const int n = mm / OHW;
            const int rem = mm - n * OHW;
            const int oh = rem / Wm;
            r_ih0[q] = oh - p.pt;
            r_iw[q] = rem - oh * Wm - p.pl;
            r_base[q] = (unsigned)(n * Hm * Wm);
            r_voff[q] = 0xFFFFFFFFu;
        }
        // ---- fragment addresses: [lane, kw] part c_kw (bytes inside a part's region) + the tile's scalar s_off[i] ----
        const int r16 = lane & 15, c16 = lane >> 4;
        const int wru = __builtin_amdgcn_readfirstlane(wr);
        int c_kw[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int jl = r16 + lead + (kw - 1) * sft;            // the row modulo 8 is that of jl
            c_kw[kw] = jl * 64 + ((c16 ^ SLN_SWZH(jl)) * 16);
        }
        int s_off[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = 128 * wru + 16 * i;
            s_off[i] = (col ? r0 : r0 + 8 * (r0 / Wm)) * 64;
        }
        // ---- walkers: the activation stage to issue next (group, outer tap index, half) -- the outer index is kh in row
        //      mode, kw in column mode; the stage holds the CENTRE tap of the other direction -- and the weight stage ----
        const int nouter = col ? p.KW : p.KH;
        int ag = 0, akh = 0, ahalf = 0;
        auto a_offsets = [&]() {
            const int cc = 2 * ag + ahalf;
            const int dih = (col ? 1 : akh) * p.dh, diw = (col ? akh : 1) * p.dw;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ih = r_ih0[q] + dih, iw = r_iw[q] + diw;
                const unsigned ci = (unsigned)(cc * T2H) + r_c0[q];
                const bool ok = r_ok[q] && (unsigned)ih < (unsigned)Hm && (unsigned)iw < (unsigned)Wm && ci < (unsigned)p.Cin;
                const unsigned off = ((r_base[q] + (unsigned)(ih * Wm + iw)) * (unsigned)p.Cin + ci) * 2u;
                r_voff[q] = ok ? off : 0xFFFFFFFFu;
            }
        };
        auto a_advance = [&]() {
            if (++ahalf == (ag < gfull ? 2 : 1)) {
                ahalf = 0;
                if (++akh == nouter) { akh = 0; ++ag; }
            }
        };
        auto issue_a = [&](int buf) {
            unsigned char *base = smem + buf * RBUF + wave * 1024;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? rsrc_a1 : rsrc_a0,
                                                             (lds_void *)(base + pp * RREG + q * 8192), 16, r_voff[q],
                                                             0, 0, 0);
            if (has3) {
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? rsrc_a1 : rsrc_a0,
                                                             (lds_void *)(base + pp * RREG + 16384), 16, r_voff[2], 0,
                                                             0, 0);
            }
        };
        int wg = 0, wkh = 0, whalf = 0, wkw = 0;
        auto issue_w = [&](int wbuf) {        // the weight stage the walker points at, in the image's (group, tap, half) order
            const int nh = wg < gfull ? 2 : 1;
            const int bs = wg * ntap * 2 + (col ? wkw * p.KW + wkh : wkh * 3 + wkw) * nh + whalf;
            unsigned char *base = smem + RB + wbuf * OBUF + wave * 2048;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(base + pp * REGION + q * 1024), 16,
                                                             b_voff, (unsigned)(bs * P + pp) * (T2 * T2H * 2) + q * 1024,
                                                             0, 0);
            if (++wkw == 3) {
                wkw = 0;
                if (++whalf == nh) {
                    whalf = 0;
                    if (++wkh == nouter) { wkh = 0; ++wg; }
                }
            }
        };
        const int nA = ncc * nouter;          // activation stages of the tile
        int a_issued = 1;
        a_offsets();
        issue_a(0);
        issue_w(0);
        a_advance();
        if (nA > 1) {
            a_offsets();
            issue_a(1);                       // stage 1 stays in flight behind stage 0
            a_advance();
            a_issued = 2;
            if (nA > 2) a_offsets();
            if (has3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const bool stagger = !(p.dbg & 8);
        if (wr == 1 && stagger) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        bf16x8 b16[4][P];
        int kw = 0, a_rd = 0;
        for (int s = 0; s < nk; ++s) {
            const unsigned char *stB = smem + RB + (s & 1) * OBUF;
            const bool moreB = s + 1 < nk;
            // the first weight stage of an activation stage refills the OTHER buffer (read until a stage ago)
            const bool fillA = kw == 0 && s > 0 && a_issued < nA;
            const int vb = (kw == 0 ? c_kw[0] : kw == 1 ? c_kw[1] : c_kw[2]) + a_rd * RBUF;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                bf16x8 a[4][P];
                // (dbg bit 16: ablation -- the weight fragments of stage 0 for every stage, no weight pieces: what the
                // k-loop would cost if the weights did not pass through LDS; results wrong)
                if (ph == 0 && (s == 0 || !(p.dbg & 65536))) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp)
                            b16[j][pp] = *(const bf16x8 *)(stB + pp * REGION + b_off16[j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        a[i][pp] = *(const bf16x8 *)(smem + pp * RREG + (vb + s_off[4 * ph + i]));
                __builtin_amdgcn_sched_barrier(0);
                if (ph == 0 && moreB && !(p.dbg & 65536)) issue_w((s + 1) & 1);
                if (ph == 1 && fillA) issue_a(a_rd ^ 1);
                if (ph == 1) {
                    // in order: the weight pieces of stage s + 1 are older than this stage's activation pieces
                    if (fillA) {
                        if (has3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                        a_advance();
                        if (++a_issued < nA) a_offsets();
                    } else {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                    if (++kw == 3) { kw = 0; a_rd ^= 1; }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) 
                        mfma16_products(a[i], b16[j], acc16[4 * ph + i][j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        if (wr == 0 && stagger) __builtin_amdgcn_s_barrier();      // re-align the two groups
    } else {
    stage_offsets();
#pragma unroll
    for (int g = 0; g < 8; ++g) issue_piece(g, 0);          // activation and weight stage 0
    stage_advance();
    ring_next(a_wr);
    if (nk > 1) stage_offsets();
    if (AD == 2 && nk > 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) issue_piece(g, 0);      // activation stage 1 stays in flight behind stage 0
        stage_advance();
        ring_next(a_wr);
        if (nk > 2) stage_offsets();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // (in order: everything but the four youngest pieces)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool stagger = !(p.dbg & 8);
    if (wr == 1 && stagger) __builtin_amdgcn_s_barrier();      // the second wave group runs one barrier behind
    asm volatile("" ::: "memory");

    bf16x8 b[2][P];
    bf16x8 b16[4][P];
    SLN_STAMP(ts_loop0);
    for (int s = 0; s < nk; ++s) {
        const unsigned char *stA = smem + a_rd * OBUF;
        const unsigned char *stB = smem + BBASE + (s & 1) * OBUF;
        const bool moreB = s + 1 < nk;        // weight stage s + 1 is issued in this stage ...
        const bool moreA = s + AD < nk;       // ... and activation stage s + AD (== n_s: the walker points at it)
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            // NPH = 4: phase = (k16 sub-step, half of the wave's 128 rows), 12 MFMAs; NPH = 2: phase =
            // k16 sub-step, 24 MFMAs
            constexpr int NA = NPH == 4 ? 2 : 4;
            const int sub = NPH == 4 ? ph >> 1 : ph, hi = NPH == 4 ? ph & 1 : 0;
            bf16x8 a[NA][P];
            SLN_STAMP(t0);
            if (MS == 16) {
                // phase 0: the four B tiles (kept for phase 1) and A tiles 0..3; phase 1: A tiles 4..7
                if (ph == 0 && (s == 0 || !(p.dbg & 65536))) {       // (dbg bit 16: as in the tap-row loop)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp)
                            b16[j][pp] = *(const bf16x8 *)(stB + pp * REGION + b_off16[j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        a[i][pp] = *(const bf16x8 *)(stA + pp * REGION + a_off16[4 * ph + i]);
            } else {
            if (!hi) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        b[j][pp] = *(const bf16x8 *)(stB + pp * REGION + (b_off[j] ^ (sub * 32)));
            }
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    a[i][pp] = *(const bf16x8 *)(stA + pp * REGION + (a_off[NA * hi + i] ^ (sub * 32)));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.dbg & 1)) {
                // the WEIGHT pieces first, then the activation pieces: the wait below lets the four youngest pieces
                // (the activation stage two ahead) stay in flight, and memory operations complete in order.
                // n_s == s + AD here; its offsets were computed in the last phase of the previous stage
                if (NPH == 4) {
                    if (ph == 0 && moreB) { issue_piece(4, s + 1); issue_piece(5, s + 1); issue_piece(6, s + 1); }
                    if (ph == 1 && moreB) issue_piece(7, s + 1);
                    if (ph == 1 && moreA) { issue_piece(0, 0); issue_piece(1, 0); }
                    if (ph == 2 && moreA) { issue_piece(2, 0); issue_piece(3, 0); }
                } else if (ph == 0) {
                    if (moreB && !(p.dbg & (268435456 | 65536))) {          // (dbg bit 28: no weight pieces in the loop)
#pragma unroll
                        for (int g = 4; g < 8; ++g) issue_piece(g, s + 1);
                    }
                    if (moreA && (p.dbg & 1073741824)) {          // (dbg bit 30: the activation pieces here too, as before)
#pragma unroll
                        for (int g = 0; g < 4; ++g) issue_piece(g, 0);
                    }
                } else if (ph == 1 && !(p.dbg & (536870912 | 1073741824))) {
                    // The activation pieces of stage s + 2 go out in PHASE 1, half a stage behind the weight pieces
                    // (round 4, same-box per-shape tables of the train step, profiles/r4_aa_shapes_*: 3x3 256 -> 256 at
                    // 64^2 452 -> 466 TFLOP/s, 512 -> 256 at 256^2 595 -> 627, the K = 1024 pointwise layers 298 -> 314
                    // and 321 -> 337; the step +0.7 ... +1.1 % in two A/B sessions, profiles/r4_y_bench_ab_phase1.json.
                    // Isolated launches of the K = 1024 layer are bimodal, 0.100 / 0.120 ms, whatever the placement:
                    // not a measure of this.)  They are still the four youngest pieces at
                    // the wait below, and their buffer -- stage s - 1's -- has been idle for longer.  (dbg bit 29:
                    // no activation pieces in the loop)
                    if (moreA && !((p.dbg & 1048576) && n_kw != 0)) {     // (dbg bit 20: only the stages of the first tap column)
#pragma unroll
                        for (int g = 0; g < 4; ++g) issue_piece(g, 0);
                    }
                }
            }
            if (ph == NPH - 1) {
                // own DMA pieces of the NEXT stage landed before the middle barrier of the last phase, which the
                // first reads of the next stage lie behind for both groups (the activation stage after it may
                // still be on its way); then the offsets of the activation stage to issue next (cheap VALU in the
                // shortest read segment)
                if (AD == 2 && moreA && !(p.dbg & (1 | 268435456 | 536870912 | 1048576))) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (moreA) {
                    stage_advance();
                    ring_next(a_wr);
                    if (n_s < nk) stage_offsets();
                }
                ring_next(a_rd);
            } else {
                // own fragment reads returned BEFORE the barrier: the other group may re-stage this
                // buffer right behind it (phase 0 of the next stage)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            SLN_STAMP(t1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            SLN_STAMP(t2);
            __builtin_amdgcn_s_setprio(1);
            if (MS == 16 && !(p.dbg & 2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) 
                        mfma16_products(a[i], b16[j], acc16[4 * ph + i][j]);
            } else if (!(p.dbg & 2)) {
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[NA * hi + i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) asm volatile("" ::"v"(a[i][pp]), "v"(b[i & 1][pp]));
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (STAMP) {   // the MFMAs' results must exist before the stamp: tie them to a scalar wait
                if (MS == 16) asm volatile("s_nop 0" ::"v"(acc16[4 * ph][0][0]), "v"(acc16[4 * ph + 3][3][3]));
                else asm volatile("s_nop 0" ::"v"(acc[NA * hi][0][0]), "v"(acc[NA * hi + NA - 1][1][15]));
            }
            SLN_STAMP(t3);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SLN_STAMP(t4);
            if (STAMP) {
                sums[ph * 4 + 0] += t1 - t0; sums[ph * 4 + 1] += t2 - t1;
                sums[ph * 4 + 2] += t3 - t2; sums[ph * 4 + 3] += t4 - t3;
            }
        }
    }
    SLN_STAMP(ts_loop1);
    if (wr == 0 && stagger) __builtin_amdgcn_s_barrier();      // re-align the two groups
    }
    __syncthreads();
    if (t < T2) s_colsum[t] = 0.f;      // (behind the staging slabs; ordered by the epilogue's first barrier)
    if (p.dbg & 32768) {                 // ablation: prologue + k-loop only (one word per lane keeps the MFMAs alive)
        if (MS == 16) { if (acc16[0][0][0] == 12345.678f && acc16[7][3][3] == 1.f) p.y[t] = 0.f; }
        else if (acc[0][0][0] == 12345.678f) p.y[t] = 0.f;
        return;
    }

    // ---- epilogue: four 64-row slabs through LDS ([64][260] floats), as in conv_fwd256_kernel ----
    float *stage = (float *)smem;
    const bool plain = epilogue_is_plain(p) && !(p.dbg & 16);      // (dbg 16: A/B against epilogue_slab)
    SLN_STAMP(t0);
    auto stage_slab = [&](int h, int buf = 0) {       // buf: which of the two staging slabs (eight-channel tiles)
        float *dst = stage + buf * SLAB_FLOATS;
        if (wr == (h >> 1)) {
            if (MS == 16) {      // C layout of 16x16: column lane % 16, rows 4 * (lane / 16) + r
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            dst[(ii * 16 + 4 * (lane >> 4) + r) * 260 + wc * 64 + j * 16 + (lane & 15)] =
                                acc16[4 * (h & 1) + ii][j][r];
            } else {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + wc * 64 + j * 32 +
                              (lane & 31)] = acc[2 * (h & 1) + ii][j][r];
            }
        }
    };
    if (EPI > 0) {
        const bool full = m0 + T2 <= p.M && n0 + T2 <= p.Cout;          // block-uniform
#define SLN_W8E(R, K, F) epilogue_tile_w8<32, 260, 512, 4, R, K, F, decltype(stage_slab), SLN_W8_DEPTH, 64, SLN_W8_DOUBLE_STAGE * SLAB_FLOATS>(p, stage, m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab)
        if (full) {
            if (EPI == 2) SLN_W8E(2, 0, true); else if (EPI == 3) SLN_W8E(0, 2, true);
            else if (EPI == 4) SLN_W8E(1, 2, true); else if (EPI == 5) SLN_W8E(1, 0, true); else SLN_W8E(0, 0, true);
        } else {
            if (EPI == 2) SLN_W8E(2, 0, false); else if (EPI == 3) SLN_W8E(0, 2, false);
            else if (EPI == 4) SLN_W8E(1, 2, false); else if (EPI == 5) SLN_W8E(1, 0, false); else SLN_W8E(0, 0, false);
        }
#undef SLN_W8E
    } else if (plain && epilogue_is_w8(p)) {
        epilogue_tile_w8_any<32, 260, 512, 4>(p, stage, m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab);
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            stage_slab(h);
            SLN_STAMP(t1);
            __syncthreads();
            SLN_STAMP(t2);
            epilogue_any<P, 64, 260, 512>(p, stage, m0 + h * 64, n0, t, s_colsum, alpha, yqs, amx, sat, plain);
            SLN_STAMP(t3);
            __syncthreads();
            SLN_STAMP(t4);
            if (STAMP && NPH == 2) {
                sums[11] += t1 - t0; sums[12] += t2 - t1; sums[13] += t3 - t2; sums[14] += t4 - t3;
            }
            t0 = t4;
        }
    }
    if (p.colsum && t < T2 && n0 + t < p.Cout && s_colsum[t] != 0.f) atomicAdd(p.colsum + n0 + t, s_colsum[t]);
    if (plain) sat = amx * yqs > SLN_F16_MAX;      // the fixed-feature slabs clamp without recording it
    if (p.yparts) amax_commit(amx, sat, p.yq, s_word);
    if (STAMP) {
        SLN_STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tile's stores have left the wave
        SLN_STAMP(ts_end);
        if (NPH == 2) sums[15] = ts_end - t1;
        if (NPH == 2) {   // (phases 2, 3 unused) whole-tile segments: prologue, k-loop, epilogue
            sums[8] = ts_loop0 - ts_start; sums[9] = ts_loop1 - ts_loop0; sums[10] = ts_end - ts_loop1;
        }
        if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sln_stamp_sums[wave * 16 + i] = sums[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_fwd128x256h_kernel (round 4): the 128-row x 256-column sibling of conv_fwd256h_kernel for the launches that
// the memory system bounds (pointwise layers, K = 256 .. 2048).  conv_fwd256h_kernel holds ONE 8-wave block per CU
// whose two wave groups share the weight stage and run half a phase apart; all 256 CUs then move through prologue,
// k-loop and epilogue together, the HBM idles during the k-loops and a tile's epilogue (half of a K = 256 tile's
// time, tools/epilogue_ablation.py) is never overlapped with anything.  Here a block is ONE such wave group --
// 4 waves, 128 rows x 256 columns, the same 8 x 4 tiles of v_mfma_f32_16x16x32_f16 per wave, the same weight
// image -- with 80 KB of LDS, so that TWO blocks are resident per CU and drift apart: one tile's epilogue runs
// under the other's k-loop, the way the three resident blocks of the 128^2 kernel cover for each other.
// LDS (81 920 B): three 16-KB activation buffers (128 rows x 64 B x 2 parts; prefetched two stages ahead) and ONE
// 32-KB weight buffer: all four weight tiles of a stage are read into registers in phase 0 (they are kept for
// phase 1 anyway), so the buffer is free for stage s + 1 right behind that read -- a barrier, then the refill,
// which has the rest of the stage to land.  Two barriers per stage.  The price: each block streams its own copy of
// the weight stage (L2 -> LDS traffic of the weights doubles), which is why the matrix-bound 3x3 launches stay on
// the 256^2 kernel.  Epilogue: the eight-channel slabs (EPI 1..5 as in conv_fwd256h_kernel), four 32-row slabs.
#ifndef SLN_128H_SLAB_ROWS
#define SLN_128H_SLAB_ROWS 32      // (64: half the barriers, twice the look-ahead registers -- measured)
#endif
#ifndef SLN_128H_DEPTH
#define SLN_128H_DEPTH 1
#endif
template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_fwd128x256h_kernel(const ConvParams p) {
    constexpr int P = 2;
    constexpr int TM = 128;                    // rows of the tile
    constexpr int REGA = TM * T2H * 2;         // one part of an activation stage: 128 rows x 64 B = 8 KB
    constexpr int ABUF = P * REGA;             // 16 KB
    constexpr int REGB = T2 * T2H * 2;         // one part of a weight stage: 256 rows x 64 B = 16 KB
    constexpr int BBASE = 3 * ABUF;            // 48 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem[BBASE + P * REGB];   // 81 920 B: two blocks per CU
    constexpr int SLD = 260;
    constexpr int SRH = SLN_128H_SLAB_ROWS;    // rows per epilogue slab
    constexpr int SLABF = SRH * SLD;           // floats of one staging slab; two of them, then the column sums
    static_assert(2 * SLABF * 4 + T2 * 4 + 16 <= BBASE + P * REGB, "staging slabs + column sums must fit");
    float *s_colsum = (float *)(smem + 2 * SLABF * 4);
    unsigned *s_word = (unsigned *)(smem + 2 * SLABF * 4 + T2 * 4);
    const float alpha = operand_unscale(p.x_scale, p.w_scale);
    const float yqs = p.yq.scale ? *p.yq.scale : 1.f;
    float amx = 0.f;
    typedef __attribute__((address_space(3))) void lds_void;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;          // wave = its 64-column strip of the tile
    const int bid = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (bid / p.gn) * TM;
    const int n0 = (bid % p.gn) * T2;

    // ---- activation DMA slots of this lane: rows 32*wave + 16*q + lane/4 (q = 0, 1), as in conv_fwd256h_kernel ----
    int a_ih0[2], a_iw0[2], a_H[2], a_W[2];
    unsigned a_base[2], a_c0[2];
    bool a_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int drow = 32 * wave + 16 * q + (lane >> 2);
        a_c0[q] = (unsigned)(((lane & 3) ^ SLN_SWZH(drow)) * 8);
        const int m = m0 + drow;
        a_ok[q] = m < p.M;
        int mm = a_ok[q] ? m : 0;
        int sg = 0;
        for (int k = 1; k < p.nseg; ++k)
            if (mm >= p.seg_m0[k]) sg = k;
        mm -= p.seg_m0[sg];
        const int OHs = p.segOH[sg], OWs = p.segOW[sg];
        a_H[q] = p.segH[sg]; a_W[q] = p.segW[sg];
        const int n = mm / (OHs * OWs);
        const int rem = mm - n * (OHs * OWs);
        const int oh = rem / OWs, ow = rem - oh * OWs;
        a_ih0[q] = oh * p.sh - p.pt;
        a_iw0[q] = ow * p.sw - p.pl;
        a_base[q] = (unsigned)(p.seg_x0[sg] + n * a_H[q] * a_W[q]);
    }
    const int ncc = (p.Cin + T2H - 1) / T2H;
    const int ntap = p.KH * p.KW;
    const int nk = ntap * ncc;
    const int gfull = ncc / 2;
    const unsigned x_part_bytes = (unsigned)(p.x_part_stride * 2);
    const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)x_part_bytes,
                                                                             0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + p.x_part_stride), 0,
                                                                             (int)x_part_bytes, 0x00020000);
    const unsigned w_tile_bytes = (unsigned)nk * P * (T2 * T2H * 2);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.w + (long)(bid % p.gn) * nk * P * (T2 * T2H)), 0, (int)w_tile_bytes, 0x00020000);

    int n_g = 0, n_tap = 0, n_half = 0, n_kh = 0, n_kw = 0, n_s = 0;     // the activation stage to issue next
    unsigned a_voff[2];
    auto stage_offsets = [&]() {
        const int cc = 2 * n_g + n_half;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ih = a_ih0[q] + n_kh * p.dh, iw = a_iw0[q] + n_kw * p.dw;
            const unsigned ci = (unsigned)(cc * T2H) + a_c0[q];
            const bool ok = a_ok[q] && (unsigned)ih < (unsigned)a_H[q] && (unsigned)iw < (unsigned)a_W[q] &&
                            ci < (unsigned)p.Cin;
            const unsigned off = ((a_base[q] + (unsigned)(ih * a_W[q] + iw)) * (unsigned)p.Cin + ci) * 2u;
            a_voff[q] = ok ? off : 0xFFFFFFFFu;
        }
    };
    auto stage_advance = [&]() {
        ++n_s;
        if (++n_half == (n_g < gfull ? 2 : 1)) {
            n_half = 0;
            ++n_tap;
            if (++n_kw == p.KW) { n_kw = 0; ++n_kh; }
            if (n_tap == ntap) { n_tap = 0; n_kh = 0; n_kw = 0; ++n_g; }
        }
    };
    int a_wr = 0, a_rd = 0;
    auto ring_next = [](int &r) { r = r == 2 ? 0 : r + 1; };
    // the four activation pieces of stage n_s (offsets in a_voff) into buffer a_wr
    auto issue_a = [&]() {
        unsigned char *base = smem + a_wr * ABUF + wave * 2048;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int pp = g >> 1, q = g & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? rsrc_a1 : rsrc_a0,
                                                     (lds_void *)(base + pp * REGA + q * 1024), 16, a_voff[q], 0, 0, 0);
        }
    };
    // the eight weight pieces of stage bs: this wave moves the 2-KB slices 2*wave and 2*wave + 1 (of eight) of both
    // parts of the stage's LDS image
    auto issue_b = [&](int bs) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int pp = g >> 2, h = (g >> 1) & 1, q = g & 1;
            const unsigned vw = (unsigned)(2 * wave + h);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_b, (lds_void *)(smem + BBASE + pp * REGB + vw * 2048 + q * 1024), 16, vw * 2048u + (unsigned)lane * 16u,
                (unsigned)(bs * P + pp) * (T2 * T2H * 2) + q * 1024, 0, 0);
        }
    };

    f32x4v acc16[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, c16 = lane >> 4;
    int a_off16[8], b_off16[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 16 * i + r16;
        a_off16[i] = row * 64 + ((c16 ^ SLN_SWZH(row)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 64 * wave + 16 * j + r16;
        b_off16[j] = row * 64 + ((c16 ^ SLN_SWZW(row)) * 16);
    }

    // (experiment, dbg 524288: the blocks that are dealt to the SECOND slot of a CU -- blockIdx / 8 / 32 odd, as far
    // as the dispatcher deals blocks round-robin over XCDs, then CUs -- start half a stage late, so that the two
    // blocks of a CU begin out of phase: one multiplies while the other reads its fragments)
    if ((p.dbg & 524288) && (((blockIdx.x >> 3) >> 5) & 1)) {
        const int n = (p.dbg >> 20) & 15;
        for (int i = 0; i <= n; ++i) __builtin_amdgcn_s_sleep(8);      // 8 x 64 = 512 cycles each
    }
    // (experiment, dbg bits 24..26: the two blocks of a CU get DIFFERENT priorities for their MFMA clusters, so that
    // when both want the matrix pipe one of them takes it whole and the other reads its fragments meanwhile -- the
    // half-phase offset conv_fwd256h_kernel enforces with a barrier.  Which blocks share a CU is the dispatcher's
    // business: the bit of blockIdx used as "parity" is selectable, 0 = off)
    const int psel = (p.dbg >> 24) & 7;
    // (psel 6: by the block's LDS base -- HW_REG_LDS_ALLOC bits 0..11 -- the two blocks of a CU differ in it by
    // construction, whatever the dispatcher did; psel 7: by the wave's hardware slot on its SIMD, HW_REG_HW_ID bits 0..3)
    const bool prio_hi = psel == 0 ? false
                       : psel == 6 ? (__builtin_amdgcn_s_getreg((11 << 11) | 6) != 0)
                       : psel == 7 ? ((__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) != 0)
                       : (((blockIdx.x >> (psel == 1 ? 0 : psel == 2 ? 3 : psel == 3 ? 8 : psel == 4 ? 9 : 4)) & 1) != 0);
    // ---- prologue: weight stage 0, activation stages 0 and 1 ----
    stage_offsets();
    issue_a();
    issue_b(0);
    stage_advance();
    ring_next(a_wr);
    if (nk > 1) {
        stage_offsets();
        issue_a();
        stage_advance();
        ring_next(a_wr);
        if (nk > 2) stage_offsets();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // all but the four youngest pieces (activation stage 1)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 b16[4][P];
    for (int s = 0; s < nk; ++s) {
        const unsigned char *stA = smem + a_rd * ABUF;
        const unsigned char *stB = smem + BBASE;
        const bool moreB = s + 1 < nk, moreA = s + 2 < nk;      // (n_s == s + 2 when moreA)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            bf16x8 a[4][P];
            if (ph == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) b16[j][pp] = *(const bf16x8 *)(stB + pp * REGB + b_off16[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    a[i][pp] = *(const bf16x8 *)(stA + pp * REGA + a_off16[4 * ph + i]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ph == 0) {
                // every wave has the stage's weight tiles in registers: the weight buffer may be refilled.  Weight
                // pieces first, then the activation pieces two stages ahead (the wait at the end of the stage lets
                // the four youngest pieces fly on; memory operations complete in order).
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (moreB) issue_b(s + 1);
                if (moreA && (p.dbg & 1073741824)) issue_a();      // (dbg bit 30: here, as first built)
            } else if (moreA && !(p.dbg & 1073741824)) {
                issue_a();          // the activation pieces two stages ahead: in phase 1, as in conv_fwd256h_kernel
            }
            __builtin_amdgcn_sched_barrier(0);
            if (prio_hi) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) 
                    mfma16_products(a[i], b16[j], acc16[4 * ph + i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (ph == 1) {
                if (moreA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (moreA) {
                    stage_advance();
                    ring_next(a_wr);
                    if (n_s < nk) stage_offsets();
                }
                ring_next(a_rd);
                // everyone's pieces of stage s + 1 have landed, and everyone is done with activation buffer a_rd - 1
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }

    // ---- epilogue: four 32-row slabs through LDS ([32][260] floats), eight channels per thread ----
    float *stage = (float *)smem;
    if (t < T2) s_colsum[t] = 0.f;            // (ordered by the first slab's barrier)
    auto stage_slab = [&](int h, int buf = 0) {     // rows SRH h .. SRH (h + 1) - 1 = SRH / 16 row tiles of every wave
        float *dst = stage + buf * SLABF;
#pragma unroll
        for (int ii = 0; ii < SRH / 16; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(ii * 16 + 4 * (lane >> 4) + r) * SLD + wave * 64 + j * 16 + (lane & 15)] =
                        acc16[(SRH / 16) * h + ii][j][r];
    };
    const bool full = m0 + TM <= p.M && n0 + T2 <= p.Cout;          // block-uniform
#define SLN_W8H(R, K, F) epilogue_tile_w8<32, SLD, 256, TM / SRH, R, K, F, decltype(stage_slab), SLN_128H_DEPTH, SRH, SLN_W8_DOUBLE_STAGE * SLABF>(p, stage, m0, n0, t, s_colsum, alpha, yqs, amx, stage_slab)
    if (full) {
        if (EPI == 2) SLN_W8H(2, 0, true); else if (EPI == 3) SLN_W8H(0, 2, true);
        else if (EPI == 4) SLN_W8H(1, 2, true); else if (EPI == 5) SLN_W8H(1, 0, true); else SLN_W8H(0, 0, true);
    } else {
        if (EPI == 2) SLN_W8H(2, 0, false); else if (EPI == 3) SLN_W8H(0, 2, false);
        else if (EPI == 4) SLN_W8H(1, 2, false); else if (EPI == 5) SLN_W8H(1, 0, false); else SLN_W8H(0, 0, false);
    }
#undef SLN_W8H
    if (p.colsum && t < T2 && n0 + t < p.Cout && s_colsum[t] != 0.f) atomicAdd(p.colsum + n0 + t, s_colsum[t]);
    if (p.yparts) amax_commit(amx, amx * yqs > SLN_F16_MAX, p.yq, s_word);
}

// ------------------------------------------------------------ weight gradient
// gw[co][tap][ci] += sum over a pixel range of gz[pix][co] * x[pix@tap][ci].
// "TN" GEMM: M = Cout, N = one tap's Cin slice, K = output pixels.  Both operand
// tiles are staged pixel-major ([k][m], rows of 128 bf16 + 32 pad = 320 B so that
// the four k-rows of a transposing read sit 16 banks apart) and the MFMA fragments
// (8 consecutive k per lane) are fetched with two ds_read_b64_tr_b16 each.
// Split-K over pixel chunks, fp32 atomics into the (callee-zeroed) gradient.
// (An unpadded XOR-swizzled image, 3 blocks/CU, measured 1-3 % slower here.)

struct WgradParams {
    const __bf16 *gz;   // [P][M][Cop]
    const __bf16 *x;    // [P][Min][Cip]
    float *gw;          // [Cout][KH][KW][Cin]
    long gz_part_stride, x_part_stride;
    int N, H, W, Cin, Cip, Cout, Cop, KH, KW, sh, sw, dh, dw, pt, pl, OH, OW;
    int M, gm, gn_per_tap, ksplit, pix_per_split, xcd_wgrad;
    const float *gz_scale, *x_scale;   // P = 2: operand scales (device scalars, NULL = 1)
    float *partial;   // two-phase split-K: [ksplit][Cout][KH*KW][Cin] slabs, one per pixel range, plain
                      // stores (every element of a slab is written by exactly one block); NULL = atomics
};

template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16 *tile, int k0, int m0, int lane) {
    // lane 4q+p of a 16-lane group addresses row k0+q, columns m0+4p..+3; the group
    // receives the 4x16 block transposed: lane i gets column m0+i, rows k0..k0+3.
    const int li = lane & 15, q = li >> 2, pq = li & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const __bf16 *p0 = tile + (k0 + q) * LD + m0 + 4 * pq;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p0 + 4 * LD));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// TM x TN output tile (Cout x Cin of one tap), 128 or 64 each: 2x2 waves of (TM/2) x (TN/2); the 64-wide
// sides serve the 64-channel C2 stage, where a 128-wide tile would be half (or three quarters) empty.
// ROW3 (round 5, the 3-wide layers of C2 / C3): a block owns the THREE taps of one kernel row.  A k-step's 32 output
// pixels are consecutive pixels of one image row (the launcher asks for stride 1, OW == W, OW % 32 == 0), so the three
// taps read the same 32 + 2 dil_w input pixels shifted by 0 / d / 2d: the gradient rows and the input rows are
// staged once per k-step for three accumulator sets -- a third of the loads and barriers per product, a third of the
// blocks.  The plan (pixel ranges, slabs) is the per-tap kernel's; TN is 64 (three accumulator sets of a 128-wide
// tile would leave one wave per SIMD).
template <int P, int TM, int TN, bool ROW3 = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int LDA = TM + 32, LDB = TN + 32;    // +32 bf16 (64 B): the four k-rows of a tr read 16 banks apart
    constexpr int NA = TM / 64, NB = TN / 64;      // 32-wide MFMA tiles per wave along M / N
    constexpr int XR = ROW3 ? BK + 16 : BK;        // input rows of a stage (ROW3: + the halo of 2 x 8 pixels at most)
    constexpr int NKW = ROW3 ? 3 : 1, NXI = ROW3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) __bf16 sA[P][BK][LDA];  // [k=pix][m=co]
    __shared__ __attribute__((aligned(16))) __bf16 sB[P][XR][LDB];  // [k=pix][n=ci]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntile = p.gm * p.gn_per_tap * (ROW3 ? p.KH : p.KH * p.KW);
    // all (Cout tile, Cin tile, tap) blocks of one pixel range on the SAME XCD, next to each other in dispatch
    // order: the range's gz and x rows come from HBM once and from that L2 for the other ntile - 1 readers
    int bid = (p.xcd_wgrad & 2) ? xcd_remap(blockIdx.x, ntile * p.ksplit) : (int)blockIdx.x;
    const int split = bid / ntile;
    bid -= split * ntile;
    const int mt = bid % p.gm;
    int rest = bid / p.gm;
    const int nt = rest % p.gn_per_tap;
    const int tap = rest / p.gn_per_tap;           // ROW3: the kernel row
    const int kh = ROW3 ? tap : tap / p.KW, kw = ROW3 ? 0 : tap - kh * p.KW;
    const int m0 = mt * TM, n0 = nt * TN;

    // staging: a tile part is 32 rows x 128 bf16 = 32 x 16 chunks of 16 B; thread t
    // handles rows (t>>4) and (t>>4)+16, chunk t&15
    const int ch = (t & 15) * 8;
    const bool a_cok = ch < TM && (m0 + ch) < p.Cop;
    const bool b_cok = ch < TN && (n0 + ch) < p.Cip;
    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.M, pix_begin + p.pix_per_split);
    const int nk = (pix_end - pix_begin + BK - 1) / BK;
    const bf16x8 zero8 = {};
    bf16x8 ra[P][2], rb[P][NXI];

    auto load_tile = [&](int ks) {
        if constexpr (ROW3) {
            // the k-step's image row (block-uniform): its 32 pixels start at (n, oh, ow0)
            const int pix0 = pix_begin + ks * BK;
            const int n = pix0 / (p.OH * p.OW);
            const int rem = pix0 - n * (p.OH * p.OW);
            const int oh = rem / p.OW, ow0 = rem - oh * p.OW;
            const int ih = oh - p.pt + kh * p.dh;
            const bool rok = pix0 < pix_end && b_cok && ih >= 0 && ih < p.H;
            const long rbase = ((long)n * p.H + ih) * p.W;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pix = pix0 + (t >> 4) + 16 * i;
                const bool pok = pix < pix_end && a_cok;
                const long aoff = (long)pix * p.Cop + m0 + ch;
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    ra[pp][i] = pok ? *(const bf16x8 *)(p.gz + pp * p.gz_part_stride + aoff) : zero8;
            }
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const int r = (t >> 4) + 16 * i;
                const int iw = ow0 - p.pl + r;
                const bool xok = rok && r < BK + 2 * p.dw && iw >= 0 && iw < p.W;
                const long boff = (rbase + iw) * p.Cip + n0 + ch;
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    rb[pp][i] = xok ? *(const bf16x8 *)(p.x + pp * p.x_part_stride + boff) : zero8;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = pix_begin + ks * BK + (t >> 4) + 16 * i;
            const bool pok = pix < pix_end;
            const long aoff = (long)pix * p.Cop + m0 + ch;
            // input pixel of this output pixel under the tap
            const int pp_ = pok ? pix : 0;
            const int n = pp_ / (p.OH * p.OW);
            const int rem = pp_ - n * (p.OH * p.OW);
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const int ih = oh * p.sh - p.pt + kh * p.dh, iw = ow * p.sw - p.pl + kw * p.dw;
            const bool xok = pok && b_cok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const long boff = (((long)n * p.H + ih) * p.W + iw) * p.Cip + n0 + ch;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                ra[pp][i] = (pok && a_cok) ? *(const bf16x8 *)(p.gz + pp * p.gz_part_stride + aoff) : zero8;
                rb[pp][i] = xok ? *(const bf16x8 *)(p.x + pp * p.x_part_stride + boff) : zero8;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int row = (t >> 4) + 16 * i;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (i < 2 && ch < TM) *(bf16x8 *)&sA[pp][row < BK ? row : 0][ch] = ra[pp][i < 2 ? i : 0];
                if (ch < TN) *(bf16x8 *)&sB[pp][row][ch] = rb[pp][i];
            }
        }
    };

    f32x16 acc[NKW][NA][NB];
#pragma unroll
    for (int q = 0; q < NKW; ++q)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    if (nk > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();
    const int g = lane >> 4;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_tile(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[NA][P], b[NB][P];
            const int k0 = kk * 16 + 8 * (g >> 1);
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    a[i][pp] = tr_frag<LDA>(&sA[pp][0][0], k0, wr * (TM / 2) + i * 32 + 16 * (g & 1), lane);
#pragma unroll
            for (int q = 0; q < NKW; ++q) {
                const int xs = ROW3 ? q * p.dw : 0;        // tap kw = q reads the input rows shifted by q dil_w pixels
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        b[j][pp] = tr_frag<LDB>(&sB[pp][0][0], k0 + xs, wc * (TN / 2) + j * 32 + 16 * (g & 1), lane);
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) mfma_products<P>(a[i], b[j], acc[q][i][j]);
            }
        }
        __syncthreads();
        if (ks + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }
    // epilogue: row = co, col = ci; atomics (split-K partial sums)
    const float alpha = P <= 2 ? operand_unscale(p.gz_scale, p.x_scale) : 1.f;
#pragma unroll
    for (int q = 0; q < NKW; ++q) {
        const int tapq = ROW3 ? kh * 3 + q : tap;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int ci = n0 + wc * (TN / 2) + j * 32 + (lane & 31);
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + wr * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (co >= p.Cout) continue;
                    const float v = acc[q][i][j][r] * alpha;
                    const long e = ((long)co * p.KH * p.KW + tapq) * p.Cin + ci;
                    if (p.partial) p.partial[(long)split * ((long)p.Cout * p.KH * p.KW * p.Cin) + e] = v;
                    else if (v != 0.f) atomicAdd(p.gw + e, v);
                }
        }
    }
}

// ---------------------------------------------------------------- 256x256 weight-gradient tile
// The weight gradient on the structure of conv_fwd256_kernel: one (tap, 256 Cout x 256 Cin) tile per
// block, 8 waves of 128x64, K = output pixels in 16-pixel stages, both operands pixel-major in LDS
// (rows of 256 bf16 = 512 B) filled by LDS-DMA three stages deep; the fragments (8 consecutive
// pixels per lane) come from ds_read_b64_tr_b16 as in conv_wgrad_kernel.  Rows are unpadded: the
// 64-B quarter of a row is XOR-swizzled with k&3 on the DMA's source side so that the four k-rows
// of a transposing read sit 16 banks apart.  Split-K over pixel ranges, fp32 atomics.
__device__ __forceinline__ bf16x8 tr_frag256(const unsigned char *region, int k0, int m0, int lane) {
    const int li = lane & 15, q = li >> 2, pq = li & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int col = (m0 + 4 * pq) ^ (q << 5);      // k0 is a multiple of 8: rows k0+q, k0+4+q share key q
    const __bf16 *p0 = (const __bf16 *)region + (k0 + q) * T2 + col;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p0 + 4 * T2));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// The same fragment through inline asm: hipcc drains the LDS-DMA queue (s_waitcnt vmcnt(0)) in front
// of the tr-read INTRINSIC even when all LDS is one object; it cannot see into the asm, so the
// reads are ordered by hand -- DMA landed: counted vmcnt + barrier before; data returned:
// tr_wait() (s_waitcnt lgkmcnt(0), tied to the fragment registers) before the first MFMA use.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
struct TrFrag { u32x2 lo, hi; };
__device__ __forceinline__ void tr_issue(TrFrag &f, unsigned region_lds, int k0, int m0, int lane) {
    const int li = lane & 15, q = li >> 2, pq = li & 3;
    const int col = (m0 + 4 * pq) ^ (q << 5);
    const unsigned a = region_lds + 2u * (unsigned)((k0 + q) * T2 + col);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a), "n"(4 * T2 * 2));
}
__device__ __forceinline__ bf16x8 tr_value(const TrFrag &f) {
    union { u32x2 u[2]; bf16x8 v; } c;
    c.u[0] = f.lo; c.u[1] = f.hi;
    return c.v;
}

template <int P>
__global__ __launch_bounds__(512) void conv_wgrad256_kernel(const WgradParams p) {
    constexpr int REGION = T2K * T2 * 2;      // one part of one operand: 16 pixel rows x 512 B = 8 KB
    constexpr int STAGE = 2 * P * REGION;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];   // ONE LDS object (see fwd256)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntile = p.gm * p.gn_per_tap * p.KH * p.KW;
    // the tap tiles of one pixel range are consecutive block ids: keep them on one XCD (one L2), so
    // that the range's gz rows are fetched from HBM once and not once per tap (xcd_wgrad)
    int bid = (p.xcd_wgrad & 1) ? xcd_remap(blockIdx.x, ntile * p.ksplit) : (int)blockIdx.x;
    const int split = bid / ntile;
    bid -= split * ntile;
    const int mt = bid % p.gm;
    int rest = bid / p.gm;
    const int nt = rest % p.gn_per_tap;
    const int tap = rest / p.gn_per_tap;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int m0 = mt * T2, n0 = nt * T2;
    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.M, pix_begin + p.pix_per_split);
    const int nk = (pix_end - pix_begin + T2K - 1) / T2K;

    // DMA slot: pixel row 2*wave + lane/32 of every region, 16-B chunk lane%32 (8 channels)
    const int drow = 2 * wave + (lane >> 5);
    const int dch = (((lane & 31) ^ ((drow & 3) << 2))) * 8;     // logical channel offset fetched
    const bool a_cok = (m0 + dch) < p.Cop;
    const bool b_cok = (n0 + dch) < p.Cip;

    auto issue = [&](int s) {
        const int pix = pix_begin + s * T2K + drow;
        const bool pok = pix < pix_end;
        const int pp_ = pok ? pix : 0;
        const int n = pp_ / (p.OH * p.OW);
        const int rem = pp_ - n * (p.OH * p.OW);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih = oh * p.sh - p.pt + kh * p.dh, iw = ow * p.sw - p.pl + kw * p.dw;
        const bool xok = pok && b_cok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const long aoff = (long)pix * p.Cop + m0 + dch;
        const long boff = (((long)n * p.H + ih) * p.W + iw) * p.Cip + n0 + dch;
        unsigned char *base = smem + (s % 3) * STAGE + wave * 1024;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const void *ga = (pok && a_cok) ? (const void *)(p.gz + pp * p.gz_part_stride + aoff)
                                            : (const void *)sln_zero_page;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(base + pp * REGION), 16, 0, 0);
        }
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const void *gb = xok ? (const void *)(p.x + pp * p.x_part_stride + boff) : (const void *)sln_zero_page;
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(base + (P + pp) * REGION), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int g = lane >> 4;
    const int k0 = 8 * (g >> 1);
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    for (int s = 0; s < nk; ++s) {
        if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        typedef __attribute__((address_space(3))) unsigned char lds_u8;
        const unsigned st = (unsigned)(size_t)(lds_u8 *)(smem + (s % 3) * STAGE);
        TrFrag fa[4][P], fb[2][P];
        const int cb = 64 * wc + 16 * (g & 1), ca = 128 * wr + 16 * (g & 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) tr_issue(fb[j][pp], st + (P + pp) * REGION, k0, cb + 32 * j, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) tr_issue(fa[i][pp], st + pp * REGION, k0, ca + 32 * i, lane);
        // B and A rows 0..63 have returned; A rows 64..127 are requested before the MFMAs start.
        // sched_barrier(0): nothing (in particular no MFMA that consumes the asm's outputs) may be
        // moved across the hand-placed wait
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) tr_issue(fa[i][pp], st + pp * REGION, k0, ca + 32 * i, lane);
        bf16x8 a[4][P], b[2][P];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) b[j][pp] = tr_value(fb[j][pp]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) a[i][pp] = tr_value(fa[i][pp]);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_products<P>(a[0], b[j], acc[0][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nk) issue(s + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_products<P>(a[1], b[j], acc[1][j]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) a[i][pp] = tr_value(fa[i][pp]);
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[i][j]);
    }
    // epilogue: row = co, col = ci; atomics (split-K partial sums)
    const float alpha = P == 2 ? operand_unscale(p.gz_scale, p.x_scale) : 1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = n0 + wc * 64 + j * 32 + (lane & 31);
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                const float v = acc[i][j][r] * alpha;
                const long e = ((long)co * p.KH * p.KW + tap) * p.Cin + ci;
                if (p.partial) p.partial[(long)split * ((long)p.Cout * p.KH * p.KW * p.Cin) + e] = v;
                else if (v != 0.f) atomicAdd(p.gw + e, v);
            }
    }
}

// ---------------------------------------------------------------- 256x256 weight-gradient tile, fp16 x 2
// conv_wgrad256_kernel on the structure of conv_fwd256h_kernel: 32-pixel stages (two 64-KB buffers),
// buffer-load DMA with range-check zero fill and 32-bit offsets computed once per stage (exact integer
// division by float reciprocal instead of two 64-bit divisions per piece), two wave groups half a
// phase apart.  A phase = one 16-pixel sub-block: 24 transposing fragment reads, 24 MFMAs.
__device__ __forceinline__ int div_small(int a, int d, float rd) {   // a < 2^24, exact
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// MS = 16: v_mfma_f32_16x16x32_f16 (see conv_fwd256h_kernel): a stage's 32 pixels are one instruction's K; lane group
// g (16 lanes) reads pixels 8g..8g+7 of the same 16 channels with two transposing reads.
template <int MS>
__global__ __launch_bounds__(512) void conv_wgrad256h_kernel(const WgradParams p) {
    constexpr int P = 2;
    constexpr int KS = 32;                       // pixels per stage
    constexpr int REGION = KS * T2 * 2;          // one part of one operand: 32 pixel rows x 512 B = 16 KB
    constexpr int STAGE = 2 * P * REGION;        // 64 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];   // ONE LDS object
    typedef __attribute__((address_space(3))) void lds_void;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntile = p.gm * p.gn_per_tap * p.KH * p.KW;
    int bid = (p.xcd_wgrad & 1) ? xcd_remap(blockIdx.x, ntile * p.ksplit) : (int)blockIdx.x;
    const int split = bid / ntile;
    bid -= split * ntile;
    const int mt = bid % p.gm;
    int rest = bid / p.gm;
    const int nt = rest % p.gn_per_tap;
    const int tap = rest / p.gn_per_tap;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int m0 = mt * T2, n0 = nt * T2;
    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.M, pix_begin + p.pix_per_split);
    const int nk = (pix_end - pix_begin + KS - 1) / KS;

    // DMA slots: pixel rows 4*wave + 2*j + lane/32 (j = 0, 1) of every region, 16-B chunk lane%32; the
    // chunk FETCHED is swizzled with the row's low two bits (= 2j + lane/32)
    const int hrow = lane >> 5;
    unsigned a_col[2], b_col[2];       // byte offset of the fetched channels inside a gz / x pixel row, or OOB
    int drow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        drow[j] = 4 * wave + 2 * j + hrow;
        const int ch = ((lane & 31) ^ ((2 * j + hrow) << 2)) * 8;
        a_col[j] = (m0 + ch) < p.Cop ? (unsigned)(m0 + ch) * 2u : 0xFFFFFFFFu;
        b_col[j] = (n0 + ch) < p.Cip ? (unsigned)(n0 + ch) * 2u : 0xFFFFFFFFu;
    }
    const unsigned gz_bytes = (unsigned)(p.gz_part_stride * 2), x_bytes = (unsigned)(p.x_part_stride * 2);
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.gz, 0, (int)gz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.gz + p.gz_part_stride), 0,
                                                                         (int)gz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + p.x_part_stride), 0,
                                                                         (int)x_bytes, 0x00020000);
    const int ohw = p.OH * p.OW;
    const float r_ohw = 1.0f / (float)ohw, r_ow = 1.0f / (float)p.OW;

    int n_s = 0;                       // the stage whose DMA is issued next
    unsigned a_voff[2], b_voff[2];
    auto stage_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pix = pix_begin + n_s * KS + drow[j];
            const bool pok = pix < pix_end;
            const int pp_ = pok ? pix : 0;
            const int n = div_small(pp_, ohw, r_ohw);
            const int rem = pp_ - n * ohw;
            const int oh = div_small(rem, p.OW, r_ow), ow = rem - oh * p.OW;
            const int ih = oh * p.sh - p.pt + kh * p.dh, iw = ow * p.sw - p.pl + kw * p.dw;
            const bool xok = pok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            a_voff[j] = (pok && a_col[j] != 0xFFFFFFFFu) ? (unsigned)pix * (unsigned)(p.Cop * 2) + a_col[j] : 0xFFFFFFFFu;
            b_voff[j] = (xok && b_col[j] != 0xFFFFFFFFu)
                            ? (unsigned)((n * p.H + ih) * p.W + iw) * (unsigned)(p.Cip * 2) + b_col[j] : 0xFFFFFFFFu;
        }
    };
    // piece g (0..7): bit 2 = operand (0 gz, 1 x), bit 1 = part, bit 0 = row pair j
    auto issue_piece = [&](int g) {
        unsigned char *dst = smem + (n_s & 1) * STAGE + ((g >> 1) & 3) * REGION + (4 * wave + 2 * (g & 1)) * 512;
        const int pp = (g >> 1) & 1, j = g & 1;
        if (g < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? ra1 : ra0, (lds_void *)dst, 16, a_voff[j], 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pp ? rb1 : rb0, (lds_void *)dst, 16, b_voff[j], 0, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int g4 = lane >> 4;
    const int k0 = MS == 16 ? 8 * g4 : 8 * (g4 >> 1);
    const int cb = 64 * wc + (MS == 16 ? 0 : 16 * (g4 & 1)), ca = 128 * wr + (MS == 16 ? 0 : 16 * (g4 & 1));
    f32x4v acc16[8][4];
    if (MS == 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    if (nk > 0) {
        stage_offsets();
#pragma unroll
        for (int g = 0; g < 8; ++g) issue_piece(g);
        n_s = 1;
        if (nk > 1) stage_offsets();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // the second wave group runs one barrier behind
    asm volatile("" ::: "memory");

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    bf16x8 b16[4][P];
    for (int s = 0; s < nk; ++s) {
        const unsigned st = (unsigned)(size_t)(lds_u8 *)(smem + (s & 1) * STAGE);
        const bool more = s + 1 < nk;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            TrFrag fa[4][P], fb[MS == 16 ? 4 : 2][P];
            const int kk = (MS == 16 ? 0 : 16 * ph) + k0;
            if (MS == 16) {
                // phase 0: the four channel tiles of x (kept for phase 1) and gz tiles 0..3; phase 1: gz tiles 4..7
                if (ph == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp) tr_issue(fb[j][pp], st + (P + pp) * REGION, kk, cb + 16 * j, lane);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        tr_issue(fa[i][pp], st + pp * REGION, kk, ca + 16 * (4 * ph + i), lane);
            } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) tr_issue(fb[j][pp], st + (P + pp) * REGION, kk, cb + 32 * j, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) tr_issue(fa[i][pp], st + pp * REGION, kk, ca + 32 * i, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more && ph == 0) {
#pragma unroll
                for (int g = 0; g < 8; ++g) issue_piece((g >> 1) | ((g & 1) << 2));
            }
            if (ph == 1) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                ++n_s;
                if (s + 2 < nk) stage_offsets();
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 a[4][P], b[2][P];
            if (MS == 16) {
                if (ph == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp) b16[j][pp] = tr_value(fb[j][pp]);
                }
            } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) b[j][pp] = tr_value(fb[j][pp]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) a[i][pp] = tr_value(fa[i][pp]);
            __builtin_amdgcn_s_setprio(1);
            if (MS == 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mfma16_products(a[i], b16[j], acc16[4 * ph + i][j]);
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_products<P>(a[i], b[j], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // re-align the two groups

    // epilogue: row = co, col = ci; partial slab (two-phase split-K) or atomics
    const float alpha = operand_unscale(p.gz_scale, p.x_scale);
    if (MS == 16) {     // C layout of 16x16: column lane % 16, rows 4 * (lane / 16) + r
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = n0 + wc * 64 + j * 16 + (lane & 15);
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m0 + wr * 128 + i * 16 + 4 * (lane >> 4) + r;
                    if (co >= p.Cout) continue;
                    const float v = acc16[i][j][r] * alpha;
                    const long e = ((long)co * p.KH * p.KW + tap) * p.Cin + ci;
                    if (p.partial) p.partial[(long)split * ((long)p.Cout * p.KH * p.KW * p.Cin) + e] = v;
                    else if (v != 0.f) atomicAdd(p.gw + e, v);
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = n0 + wc * 64 + j * 32 + (lane & 31);
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                const float v = acc[i][j][r] * alpha;
                const long e = ((long)co * p.KH * p.KW + tap) * p.Cin + ci;
                if (p.partial) p.partial[(long)split * ((long)p.Cout * p.KH * p.KW * p.Cin) + e] = v;
                else if (v != 0.f) atomicAdd(p.gw + e, v);
            }
    }
}

// Second phase of the split-K weight gradient: gw[e] = sum over the pixel ranges, in range order --
// the same sum on every run (fp32 atomics arrive in any order), and plain stores instead of ~8 GB of
// atomic traffic per step.  One thread per 4 elements when the count allows.
// taps > 0: gw is written in the parameter's own [Cout][Cin][KH][KW] order (element (co, tap, ci) of the
// slabs' [Cout][taps][Cin] order goes to (co*Cin + ci)*taps + tap): autograd then accumulates it as it is,
// without the layout copy a permuted view costs per weight and step.
__device__ __forceinline__ void wgrad_reduce_block(const float *__restrict__ partial, int ksplit, long n,
                                                   float *__restrict__ gw, int taps, int Cin, long block) {
    // 32 element quads x 8 range lanes per block: lane kl adds the ranges kl, kl+8, ... in order, the
    // eight sums are then added in lane order -- a fixed summation tree, whatever the launch timing
    __shared__ float4 s_part[8][32];
    const int q = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long i4 = (block * 32 + q) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n) {
        if (i4 + 3 < n && (n & 3) == 0) {
            int sidx = kl;
            for (; sidx + 24 < ksplit; sidx += 32) {       // four loads in flight
                const float4 b0 = *(const float4 *)(partial + (long)sidx * n + i4);
                const float4 b1 = *(const float4 *)(partial + (long)(sidx + 8) * n + i4);
                const float4 b2 = *(const float4 *)(partial + (long)(sidx + 16) * n + i4);
                const float4 b3 = *(const float4 *)(partial + (long)(sidx + 24) * n + i4);
                a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
                a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
                a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
                a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
            }
            for (; sidx < ksplit; sidx += 8) {
                const float4 b = *(const float4 *)(partial + (long)sidx * n + i4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
        } else {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int sidx = kl; sidx < ksplit; sidx += 8)
                for (int e = 0; e < 4; ++e)
                    if (i4 + e < n) t4[e] += partial[(long)sidx * n + i4 + e];
            a = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
    }
    s_part[kl][q] = a;
    __syncthreads();
    if (kl == 0 && i4 < n) {
        float4 r = s_part[0][q];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float4 b = s_part[k][q];
            r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
        }
        if (taps > 1) {
            const float t4[4] = {r.x, r.y, r.z, r.w};
            for (int e = 0; e < 4 && i4 + e < n; ++e) {
                const long i = i4 + e;
                const long row = i / Cin;                    // co * taps + tap
                const int ci = (int)(i - row * Cin);
                const long co = row / taps;
                const int tap = (int)(row - co * taps);
                gw[(co * Cin + ci) * taps + tap] = t4[e];
            }
        } else if (i4 + 3 < n) *(float4 *)(gw + i4) = r;
        else {
            const float t4[4] = {r.x, r.y, r.z, r.w};
            for (int e = 0; e < 4 && i4 + e < n; ++e) gw[i4 + e] = t4[e];
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int ksplit,
                                                           long n, float *__restrict__ gw, int taps, int Cin) {
    wgrad_reduce_block(partial, ksplit, n, gw, taps, Cin, blockIdx.x);
}

// The reduce passes of up to SLN_WGRAD_REDUCE_BATCH layers in one launch (sln_wgrad_reduce_batch_f32): a train
// step has ~137 weight gradients, each followed by a 13-us reduce launch; deferred and batched they are ~9.
struct WgradReduceBatch {
    sln_wgrad_reduce_desc_t d[SLN_WGRAD_REDUCE_BATCH];
    long first[SLN_WGRAD_REDUCE_BATCH + 1];      // first block of entry i; first[n] = grid size
    int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceBatch b) {
    int e = 0;
    while (e + 1 < b.n && (long)blockIdx.x >= b.first[e + 1]) ++e;      // (block-uniform)
    const sln_wgrad_reduce_desc_t &d = b.d[e];
    wgrad_reduce_block(d.partial, d.ksplit, d.n, d.gw, d.taps, d.Cin, (long)blockIdx.x - b.first[e]);
}

static inline int sln_knob(const char *name, int dflt);
// Split-K plan of sln_conv2d_wgrad_f32 (pure host function of the problem).
struct WgradPlan { int tile, gm, gn, ksplit, pps, tm, tn; };
static WgradPlan wgrad_plan(long M, int Cout, int Cin, int taps, int parts) {
    WgradPlan w;
    w.tile = sln_conv_wgrad_tile(M, Cout, Cin, taps, parts);
    if (w.tile == T2) {
        const long gm2 = sln_div_up(Cout, T2), gn2 = sln_div_up(Cin, T2), nt2 = gm2 * gn2 * taps;
        long ks2 = 256 / nt2;                         // one round of the 256 CUs
        const long cap = (M + 255) / 256;             // >= 16 stages per block
        if (ks2 > cap) ks2 = cap;
        if (ks2 < 1) ks2 = 1;
        long pps2 = (M + ks2 - 1) / ks2;
        pps2 = ((pps2 + 31) / 32) * 32;
        ks2 = (M + pps2 - 1) / pps2;
        w.gm = (int)gm2; w.gn = (int)gn2; w.ksplit = (int)ks2; w.pps = (int)pps2; w.tm = w.tn = T2;
        return w;
    }
    const bool t64 = sln_knob("SLN_WGRAD_T64", 1) != 0;        // 0 disables the 64-wide sides
    w.tm = (t64 && Cout <= 64) ? 64 : BM; w.tn = (t64 && Cin <= 64) ? 64 : BN;
    w.gm = sln_div_up(Cout, w.tm);
    w.gn = sln_div_up(Cin, w.tn);
    const long ntile = (long)w.gm * w.gn * taps;
    // split the pixel range: ~24 blocks per CU (short blocks balance the tail; swept 2..32
    // on the train step), but at least 1024 pixels (32 k-steps) per block
    long ks = (256L * 24 + ntile - 1) / ntile;
    const long max_ks = (M + 1023) / 1024;
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    long pps = (M + ks - 1) / ks;
    pps = ((pps + BK - 1) / BK) * BK;
    ks = (M + pps - 1) / pps;
    w.ksplit = (int)ks; w.pps = (int)pps;
    return w;
}

// ---------------------------------------------------------------- C ABI
// Tuning knobs (A/B runs, the tests' forced tile modes) exist only in debug sessions: unless the
// process was started with SLN_DEBUG_KNOBS set, no entry point reads the environment and every
// choice is a pure function of its arguments.
static inline int sln_knob(const char *name, int dflt) {
    static const bool enabled = getenv("SLN_DEBUG_KNOBS") != nullptr;
    if (!enabled) return dflt;
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    return (int)(g < 1 ? 1 : g);
}

// Debug sessions only: the stamp sums of the last conv_fwd256h_kernel<true> launch (block 0), host copy.
extern "C" int sln_debug_read_stamps(uint64_t *out128) {
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(sln_stamp_sums), sizeof(unsigned long long) * 128) == hipSuccess
               ? SLN_OK : SLN_ERR_LAUNCH;
}

extern "C" int64_t sln_conv_tiled_weight_elems(int O, int I, int KH, int KW, int layout) {
    if (O < 1 || I < 1 || KH < 1 || KW < 1) return 0;
    const int kc = layout == SLN_WEIGHTS_TILED256H ? T2H : T2K;
    return (int64_t)sln_div_up(O, T2) * KH * KW * sln_div_up(I, kc) * (T2 * kc);
}

// Layout of the weight parts sln_conv2d_fwd(_ms)_f32 reads for this problem (host-side rule).
// conv_fwd256h_kernel addresses its operands with 32-bit byte offsets (buffer loads): one part of the
// activations (<= the output rows' worth of input pixels x Cin here: stride 1) and one Cout tile of the
// tiled weights must stay below 4 GiB each.
static inline bool fwd256h_fits(long x_part_elems, long w_tile_elems) {
    return x_part_elems * 2 < 4294967295L && w_tile_elems * 2 < 4294967295L;
}
extern "C" int sln_conv_fwd_weights_layout(int64_t M, int Cout, int Cin, int taps, int parts, int64_t x_pixels) {
    if (Cin < 1 || taps < 1) return SLN_WEIGHTS_ROWS;
    if (sln_conv_fwd_tile(M, Cout, (int64_t)taps * Cin, parts) != T2) return SLN_WEIGHTS_ROWS;
    const bool h = parts == 2 && sln_knob("SLN_CONV_F16_KERNEL", 1) &&
                   fwd256h_fits(x_pixels * Cin, (long)taps * sln_div_up(Cin, T2H) * 2 * (T2 * T2H));
    return h ? SLN_WEIGHTS_TILED256H : SLN_WEIGHTS_TILED256;
}

extern "C" int sln_conv_split_weights_f32(const float *w, int O, int I, int I_pad, int KH, int KW,
                                          long s_o, long s_i, long s_kh, long s_kw, int flip, int parts,
                                          int layout, uint16_t *out, const float *q_scale, float *q_amax,
                                          int32_t *q_saturated, sln_stream_t stream) {
    sln_enter();
    if (!w || O < 1 || I < 1 || I_pad < I || KH < 1 || KW < 1 || parts < 1 || parts > 3 || layout < 0 ||
        layout > 2 || (layout == SLN_WEIGHTS_TILED256H && parts != 2) || (layout != SLN_WEIGHTS_ROWS && parts == 1))
        return SLN_ERR_INVALID_ARG;
    if (!out && !(parts <= 2 && q_amax)) return SLN_ERR_INVALID_ARG;   // out == NULL: amax-only pass
    const SplitScale q = {q_scale, q_amax, q_saturated};
    if (layout == SLN_WEIGHTS_TILED256H) {
        const long total = sln_conv_tiled_weight_elems(O, I, KH, KW, layout);
        hipLaunchKernelGGL(split_weights_tiledh_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           w, O, I, KH, KW, s_o, s_i, s_kh, s_kw, flip, (__bf16 *)out, q);
        return sln_launch_status();
    }
    if (layout == SLN_WEIGHTS_TILED256) {
        const long total = sln_conv_tiled_weight_elems(O, I, KH, KW, layout);
        hipLaunchKernelGGL(split_weights_tiled_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           w, O, I, KH, KW, s_o, s_i, s_kh, s_kw, flip, parts, (__bf16 *)out, q);
        return sln_launch_status();
    }
    const long total = (long)O * KH * KW * I_pad;
    hipLaunchKernelGGL(split_weights_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, O,
                       I, I_pad, KH, KW, s_o, s_i, s_kh, s_kw, flip, parts, (__bf16 *)out, q);
    return sln_launch_status();
}

extern "C" int sln_act_split_f32(const float *x, int64_t M, int C, int C_pad, int parts, uint16_t *out,
                                 const float *q_scale, float *q_amax, int32_t *q_saturated,
                                 sln_stream_t stream) {
    sln_enter();
    if (M < 0 || C < 1 || C_pad < C || (C_pad & 7) || parts < 1 || parts > 3) return SLN_ERR_INVALID_ARG;
    if (M == 0) return SLN_OK;
    if (!x || (!out && !(parts <= 2 && q_amax))) return SLN_ERR_INVALID_ARG;
    const SplitScale q = {q_scale, q_amax, q_saturated};
    if (parts == 1)
        hipLaunchKernelGGL(act_split_kernel<1>, dim3(ew_grid(M * (C_pad / 4))), dim3(256), 0,
                           (hipStream_t)stream, x, (long)M, C, C_pad, (__bf16 *)out, q);
    else if (parts == 2)
        hipLaunchKernelGGL(act_split_kernel<2>, dim3(ew_grid(M * (C_pad / 4))), dim3(256), 0,
                           (hipStream_t)stream, x, (long)M, C, C_pad, (__bf16 *)out, q);
    else
        hipLaunchKernelGGL(act_split_kernel<3>, dim3(ew_grid(M * (C_pad / 4))), dim3(256), 0,
                           (hipStream_t)stream, x, (long)M, C, C_pad, (__bf16 *)out, q);
    return sln_launch_status();
}

extern "C" int sln_conv_split_weights_batch_f32(const sln_split_desc_t *descs, const int32_t *chunk_entry,
                                               const int64_t *chunk_first, int n_chunks, int chunk_elems,
                                               sln_stream_t stream) {
    sln_enter();
    if (n_chunks < 0 || chunk_elems < 256) return SLN_ERR_INVALID_ARG;
    if (n_chunks == 0) return SLN_OK;
    if (!descs || !chunk_entry || !chunk_first) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(split_weights_batch_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, descs,
                       chunk_entry, chunk_first, chunk_elems);
    return sln_launch_status();
}

extern "C" int sln_im2col_split_f32(const float *x, int N, int H, int W, int C, int KH, int KW, int stride_h,
                                    int stride_w, int pad_top, int pad_left, int OH, int OW, int K_pad,
                                    int parts, uint16_t *out, int64_t out_rows, int64_t row0,
                                    const float *q_scale, float *q_amax, int32_t *q_saturated,
                                    sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 1 || KH < 1 || KW < 1 || KH > 255 || KW > 255 || stride_h < 1 ||
        stride_w < 1 || OH < 1 || OW < 1 || parts < 1 || parts > 3)
        return SLN_ERR_INVALID_ARG;
    if (K_pad < KH * KW * C || (K_pad & 7) || K_pad > 4096) return SLN_ERR_INVALID_ARG;
    const long M = (long)N * OH * OW;
    if (row0 < 0 || row0 + M > out_rows) return SLN_ERR_INVALID_ARG;
    if (M == 0) return SLN_OK;
    if (!x || (!out && !(parts <= 2 && q_amax))) return SLN_ERR_INVALID_ARG;
    const SplitScale q = {q_scale, q_amax, q_saturated};
    const size_t lds = sizeof(int) * 2 * (size_t)K_pad;
    const long pstride = (long)out_rows * K_pad;
    if (parts == 1)
        hipLaunchKernelGGL(im2col_split_kernel<1>, dim3(ew_grid(M * (K_pad / 4))), dim3(256), lds,
                           (hipStream_t)stream, x, N, H, W, C, KH, KW, stride_h, stride_w, pad_top, pad_left, OH,
                           OW, K_pad, pstride, (long)row0, (__bf16 *)out, q);
    else if (parts == 2)
        hipLaunchKernelGGL(im2col_split_kernel<2>, dim3(ew_grid(M * (K_pad / 4))), dim3(256), lds,
                           (hipStream_t)stream, x, N, H, W, C, KH, KW, stride_h, stride_w, pad_top, pad_left, OH,
                           OW, K_pad, pstride, (long)row0, (__bf16 *)out, q);
    else
        hipLaunchKernelGGL(im2col_split_kernel<3>, dim3(ew_grid(M * (K_pad / 4))), dim3(256), lds,
                           (hipStream_t)stream, x, N, H, W, C, KH, KW, stride_h, stride_w, pad_top, pad_left, OH,
                           OW, K_pad, pstride, (long)row0, (__bf16 *)out, q);
    return sln_launch_status();
}

extern "C" int sln_col2im_f32(const float *cols, int N, int H, int W, int C, int KH, int KW, int stride_h,
                              int stride_w, int pad_top, int pad_left, int OH, int OW, int K_pad, float *gx,
                              sln_stream_t stream) {
    sln_enter();
    if (N < 0 || H < 1 || W < 1 || C < 1 || KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1 || OH < 1 ||
        OW < 1 || K_pad < KH * KW * C)
        return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!cols || !gx) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(col2im_kernel, dim3(ew_grid((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, cols,
                       N, H, W, C, KH, KW, stride_h, stride_w, pad_top, pad_left, OH, OW, K_pad, gx);
    return sln_launch_status();
}

extern "C" int sln_conv_grad_prep_f32(const float *gy, const float *y, const uint16_t *y_part0,
                                      const float *scale, int64_t M,
                                      int C, int C_pad, int parts, float *gu, uint16_t *gz_parts,
                                      float *gbias, const float *q_scale, float *q_amax,
                                      int32_t *q_saturated, sln_stream_t stream) {
    sln_enter();
    // (parts | SLN_SUMS_PREZEROED: the caller hands gbias over already zero -- slices of one arena it clears once
    // per step -- and the fill launch per layer is saved)
    const bool sums_zero = (parts & SLN_SUMS_PREZEROED) != 0;
    parts &= ~SLN_SUMS_PREZEROED;
    if (M < 0 || C < 1 || C_pad < C || (C_pad & 7) || parts < 1 || parts > 3) return SLN_ERR_INVALID_ARG;
    if (!gy || (!gz_parts && !(parts <= 2 && q_amax))) return SLN_ERR_INVALID_ARG;
    if (y_part0 && (y || parts > 2)) return SLN_ERR_INVALID_ARG;   // one ReLU pattern; fp16 parts only
    hipStream_t st = (hipStream_t)stream;
    if (!sums_zero && gz_parts && gbias && hipMemsetAsync(gbias, 0, sizeof(float) * C, st) != hipSuccess)
        return SLN_ERR_LAUNCH;
    if (M == 0) return SLN_OK;
    int tw = 1;
    while (tw < C_pad / 4 && tw < 256) tw <<= 1;
    long grid = sln_div_up(M, (long)(256 / tw) * 2);
    if (grid > 2048) grid = 2048;
    const SplitScale q = {q_scale, q_amax, q_saturated};
    const PoolSrc nopool = {};
    if (parts == 1)
        hipLaunchKernelGGL(grad_prep_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, gy, y,
                           (const __bf16 *)y_part0, scale, (long)M, C, C_pad, gu, (__bf16 *)gz_parts, gbias, q, nopool);
    else if (parts == 2)
        hipLaunchKernelGGL(grad_prep_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, gy, y,
                           (const __bf16 *)y_part0, scale, (long)M, C, C_pad, gu, (__bf16 *)gz_parts, gbias, q, nopool);
    else
        hipLaunchKernelGGL(grad_prep_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, gy, y,
                           (const __bf16 *)nullptr, scale, (long)M, C, C_pad, gu, (__bf16 *)gz_parts, gbias, q, nopool);
    return sln_launch_status();
}

// The same preparation for a layer whose output went through a max-pool (K x K, stride S, clipped windows:
// sln_maxpool_fwd_f32's geometry and winning taps): the layer's gradient is gathered from the POOLED gradient
// g_pool [N,OH,OW,C] on the fly -- no pool-backward launch, no fp32 gradient map of the layer's output.
// y: the layer's fp32 output (its ReLU pattern) or NULL; C % 8 == 0.
extern "C" int sln_conv_grad_prep_pooled_f32(const float *g_pool, const uint8_t *argmax, int N, int H, int W, int K,
                                             int S, int pad_top, int pad_left, int OH, int OW, const float *y,
                                             const float *scale, int C, int parts, uint16_t *gz_parts, float *gbias,
                                             const float *q_scale, float *q_amax, int32_t *q_saturated,
                                             sln_stream_t stream) {
    sln_enter();
    const bool sums_zero = (parts & SLN_SUMS_PREZEROED) != 0;
    parts &= ~SLN_SUMS_PREZEROED;
    if (N < 0 || H < 1 || W < 1 || C < 8 || (C & 7) || parts < 1 || parts > 3 || K < 1 || K > 15 || S < 1 || pad_top < 0 ||
        pad_left < 0 || pad_top >= K || pad_left >= K || OH < 1 || OW < 1)
        return SLN_ERR_INVALID_ARG;
    if (!g_pool || !argmax || (!gz_parts && !(parts <= 2 && q_amax))) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!sums_zero && gz_parts && gbias && hipMemsetAsync(gbias, 0, sizeof(float) * C, st) != hipSuccess)
        return SLN_ERR_LAUNCH;
    const long M = (long)N * H * W;
    if (M == 0) return SLN_OK;
    int tw = 1;
    while (tw < C / 4 && tw < 256) tw <<= 1;
    long grid = sln_div_up(M, (long)(256 / tw) * 2);
    if (grid > 2048) grid = 2048;
    const SplitScale q = {q_scale, q_amax, q_saturated};
    const PoolSrc ps = {g_pool, argmax, H, W, K, S, pad_top, pad_left, OH, OW};
    if (parts == 1)
        hipLaunchKernelGGL(grad_prep_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, g_pool, y, (const __bf16 *)nullptr,
                           scale, M, C, C, (float *)nullptr, (__bf16 *)gz_parts, gbias, q, ps);
    else if (parts == 2)
        hipLaunchKernelGGL(grad_prep_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, g_pool, y, (const __bf16 *)nullptr,
                           scale, M, C, C, (float *)nullptr, (__bf16 *)gz_parts, gbias, q, ps);
    else
        hipLaunchKernelGGL(grad_prep_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, g_pool, y, (const __bf16 *)nullptr,
                           scale, M, C, C, (float *)nullptr, (__bf16 *)gz_parts, gbias, q, ps);
    return sln_launch_status();
}

extern "C" int sln_scale_update_f32(float *amax, float *scale, float *history, int32_t *cursor, int n,
                                    int64_t history_stride, int window, int target_log2, sln_stream_t stream) {
    sln_enter();
    if (n < 0 || target_log2 < -14 || target_log2 > 15) return SLN_ERR_INVALID_ARG;
    if (history && (!cursor || window < 1 || window > 1024 || history_stride < n)) return SLN_ERR_INVALID_ARG;
    if (n == 0) return SLN_OK;
    if (!amax || !scale) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scale_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, amax, scale,
                       history, cursor, n, (long)history_stride, window, target_log2, (const signed char *)nullptr);
    return sln_launch_status();
}

// The same with a per-slot table of EXTRA head room (bits, 0 ... 8) below target_log2: gradient tensors spike by more
// than the 2^5 an activation's scale leaves (the RPN class-logit gradient of a level whose anchors were hardly drawn in
// the window), and their absolute precision floor, 2^-25 of the scaled range, is far below what a weight-gradient sum
// resolves.
extern "C" int sln_scale_update_headroom_f32(float *amax, float *scale, float *history, int32_t *cursor,
                                             const int8_t *headroom, int n, int64_t history_stride, int window,
                                             int target_log2, sln_stream_t stream) {
    sln_enter();
    if (n < 0 || target_log2 < -14 || target_log2 > 15) return SLN_ERR_INVALID_ARG;
    if (history && (!cursor || window < 1 || window > 1024 || history_stride < n)) return SLN_ERR_INVALID_ARG;
    if (n == 0) return SLN_OK;
    if (!amax || !scale) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scale_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, amax, scale,
                       history, cursor, n, (long)history_stride, window, target_log2, (const signed char *)headroom);
    return sln_launch_status();
}

// Which forward kernel sln_conv2d_fwd(_ms)_f32 uses for M output pixels, Cout channels and a
// reduction of K = KH*KW*Cin: 256 = conv_fwd256_kernel (one 256x256 tile per CU, LDS-DMA pipeline)
// when Cout fills >= 176 of the last 256 columns, the tiles come in (nearly) whole rounds of the 256 CUs
// and K is long enough to amortise the pipeline; else 128.  SLN_CONV_TILE256 = 0 never,
// 1 (default) by this rule, 2 always (tests); read on every call.
// Which forward kernel the last sln_conv2d_fwd*_f32 call of this thread launched (profiling labels): 0 the 128^2
// kernel, 1 conv_fwd256_kernel, 2 conv_fwd256h_kernel, 3 conv_fwd128x256h_kernel, 4 conv_fwd256h_kernel's tap-row instances,
// 5 conv_fwd_kernel's ROW3 instances.
static thread_local int sln_last_fwd_kernel = 0;
extern "C" int sln_conv_fwd_last_kernel(void) { return sln_last_fwd_kernel; }

extern "C" int sln_conv_fwd_tile(int64_t M, int Cout, int64_t K, int parts) {
    const int mode = sln_knob("SLN_CONV_TILE256", 1);
    if (M < 1 || Cout < 1 || M > 2147483647L - T2 || parts == 1) return BM;     // (single fp16 part: the generic kernel)
    if (mode == 2) return T2;
    if (mode != 1) return BM;
    const long nb2 = sln_div_up(M, T2) * (long)sln_div_up(Cout, T2);
    const double fill = (double)nb2 / (double)(sln_div_up(nb2, 256) * 256L);
    // at least 176 of the last 256 columns in use (ASPP's Cout = 182: +6...11 % per launch over
    // the 128x128 kernel, which wastes the same share of its second 128-column tile)
    const bool cols = Cout >= 176 && (Cout % T2 == 0 || Cout % T2 >= 160);
    return (cols && fill >= 0.85 && K >= sln_knob("SLN_CONV_TILE256_MINK", 256)) ? T2 : BM;   // K >= 256: 16 stages (1x1 256 -> 1024: +3 %)
}

extern "C" int sln_conv2d_fwd_ms_f32(const uint16_t *x_parts, int nseg, const int32_t *seg_nhw, int Cin,
                                     const uint16_t *w_parts, int w_layout, int parts, int Cout, int KH,
                                     int KW, int stride_h, int stride_w, int dil_h, int dil_w, int pad_top,
                                     int pad_left, int pad_bottom, int pad_right, const float *scale,
                                     const float *shift, const float *residual, int relu,
                                     const float *mask, const float *post_scale, float *y,
                                     uint16_t *y_parts, float *colsum, const float *x_scale,
                                     const float *w_scale, const float *y_q_scale, float *y_q_amax,
                                     int32_t *y_q_saturated, const uint16_t *residual_parts,
                                     const float *residual_scale, const uint16_t *mask_part0,
                                     sln_stream_t stream) {
    sln_enter();
    if ((residual_parts && residual) || (mask_part0 && mask)) return SLN_ERR_INVALID_ARG;
    if (residual_parts && (mask || mask_part0)) return SLN_ERR_UNSUPPORTED;   // (no such reader on the path)
    if (!x_parts || !w_parts || (!y && !y_parts) || !seg_nhw || nseg < 1 || nseg > SLN_MAX_SEG || Cin < 1 || Cout < 1 ||
        KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1 || dil_h < 1 || dil_w < 1)
        return SLN_ERR_INVALID_ARG;
    if (parts < 1 || parts > 3) return SLN_ERR_INVALID_ARG;
    if (Cin % 8 != 0) return SLN_ERR_UNSUPPORTED;  // 16-B vector loads along (padded) channels
    ConvParams p;
    long M = 0, Min = 0;
    for (int q = 0; q < nseg; ++q) {
        const int N = seg_nhw[3 * q], H = seg_nhw[3 * q + 1], W = seg_nhw[3 * q + 2];
        if (N < 0 || H < 1 || W < 1) return SLN_ERR_INVALID_ARG;
        const int OH = (H + pad_top + pad_bottom - dil_h * (KH - 1) - 1) / stride_h + 1;
        const int OW = (W + pad_left + pad_right - dil_w * (KW - 1) - 1) / stride_w + 1;
        if (OH < 1 || OW < 1) return SLN_ERR_INVALID_ARG;
        if (M > 2147483647L - BM || Min > 2147483647L) return SLN_ERR_UNSUPPORTED;
        p.segH[q] = H; p.segW[q] = W; p.segOH[q] = OH; p.segOW[q] = OW;
        p.seg_m0[q] = (int)M; p.seg_x0[q] = (int)Min;
        M += (long)N * OH * OW;
        Min += (long)N * H * W;
    }
    for (int q = nseg; q < SLN_MAX_SEG; ++q) {
        p.segH[q] = p.segW[q] = p.segOH[q] = p.segOW[q] = 1;
        p.seg_m0[q] = 2147483647; p.seg_x0[q] = 0;
    }
    const bool sums_zero = (relu & SLN_SUMS_PREZEROED) != 0;      // colsum arrives zeroed (see sln_amodal.h)
    relu &= 1;
    if (colsum && !sums_zero && hipMemsetAsync(colsum, 0, sizeof(float) * Cout, (hipStream_t)stream) != hipSuccess)
        return SLN_ERR_LAUNCH;
    if (M == 0) return SLN_OK;
    if (M > 2147483647L - BM || Min > 2147483647L) return SLN_ERR_UNSUPPORTED;
    p.nseg = nseg;
    p.x = (const __bf16 *)x_parts; p.w = (const __bf16 *)w_parts;
    p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.yparts = (__bf16 *)y_parts; p.mask = mask; p.colsum = colsum; p.post_scale = post_scale;
    p.x_scale = x_scale; p.w_scale = w_scale;
    p.yq.scale = y_q_scale; p.yq.amax = y_q_amax; p.yq.saturated = y_q_saturated;
    p.res_parts = (const __bf16 *)residual_parts; p.res_scale = residual_scale;
    p.mask_part0 = (const __bf16 *)mask_part0;
    p.Cop = (Cout + 7) / 8 * 8;
    p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.sh = stride_h; p.sw = stride_w; p.dh = dil_h; p.dw = dil_w; p.pt = pad_top; p.pl = pad_left;
    p.relu = relu;
    p.M = (int)M;
    p.dbg = sln_knob("SLN_CONV_DBG", 0);
    p.Ktot = KH * KW * Cin;
    p.x_part_stride = Min * Cin;
    p.w_part_stride = (long)Cout * p.Ktot;
    p.y_part_stride = M * p.Cop;
    p.cin_chunks = sln_div_up(Cin, BK);
    const bool use256 = sln_conv_fwd_tile(M, Cout, KH * KW * Cin, parts) == T2;
    // the 256x256 kernel DMAs the weights in its own LDS-image order, the 128x128 kernel reads rows
    if (w_layout != sln_conv_fwd_weights_layout(M, Cout, Cin, KH * KW, parts, Min)) return SLN_ERR_INVALID_ARG;
    p.w_tiled = w_layout;
    p.tapmode = 0;
    if (residual_parts || mask_part0) {
        // parts-only operands of the epilogue exist in the fixed-feature slab only: fp16 x 2, whole 16-B row
        // groups, and not the round-1 256^2 kernel (operands beyond the 4-GiB buffer range)
        if (parts > 2 || (Cout & 7) || w_layout == SLN_WEIGHTS_TILED256) return SLN_ERR_UNSUPPORTED;
        p.dbg &= ~16;
    }
    const long gm2 = sln_div_up(M, T2), gn2 = sln_div_up(Cout, T2);
    const long nb2 = gm2 * gn2;
    sln_last_fwd_kernel = use256 ? (w_layout == SLN_WEIGHTS_TILED256H ? 2 : 1) : 0;
    if (use256) {
        p.gm = (int)gm2; p.gn = (int)gn2;
        if (w_layout == SLN_WEIGHTS_TILED256H) {
            const bool stamp = sln_knob("SLN_CONV_STAMP", 0) != 0;
            const dim3 g2((unsigned)nb2), b2(512);
            const bool m16 = sln_knob("SLN_CONV_MFMA16", 1) != 0;
            // (the four-phase body and the stamped 32x32x16 body of round 2 are no longer instantiated: build time)
            if (stamp) hipLaunchKernelGGL((conv_fwd256h_kernel<true, 2, 16>), g2, b2, 0, (hipStream_t)stream, p);
            else if (m16) {
                // the instance that carries only the epilogue this launch needs (SLN_CONV_EPI=0: the all-in-one instance)
                const bool w8 = epilogue_is_plain(p) && !(p.dbg & 16) && epilogue_is_w8(p) && !(p.dbg & 256) &&
                                sln_knob("SLN_CONV_EPI", 1) != 0;
                // pointwise layers whose epilogue ADDS A SHORTCUT (forward expand convolutions, the block-input data
                // gradients): the 128 x 256 kernel, two blocks per CU at different phases.  Same-box table
                // (profiles/r4_l_hbm_layers_128x256.txt): K = 256 -> 1024 with a parts / fp32 shortcut -11 %, with
                // shortcut + mask -15 %, C5's 512 -> 2048 -9 ... -17 %; WITHOUT a shortcut the tile's epilogue is
                // short and the second copy of the weight stream costs more than the overlap returns (K = 1024 ->
                // 256 parts-only +16 %): those stay on the 256^2 kernel.  SLN_CONV_TILE128H: 0 never, 2 whenever the
                // eight-channel epilogue applies (tests: also 3x3).
                const int k128 = sln_knob("SLN_CONV_TILE128H", 1);
                if (w8 && (k128 == 2 || (k128 == 1 && KH * KW == 1 && (p.res_parts || p.residual)))) {
                    const long gm1 = sln_div_up(M, 128);
                    if (gm1 * gn2 > 2147483647L) return SLN_ERR_UNSUPPORTED;
                    p.gm = (int)gm1;
                    const dim3 g1((unsigned)(gm1 * gn2)), b1(256);
                    sln_last_fwd_kernel = 3;
#define SLN_L128(E) hipLaunchKernelGGL((conv_fwd128x256h_kernel<E>), g1, b1, 0, (hipStream_t)stream, p)
                    if (p.res_parts) SLN_L128(2);
                    else if (p.residual && p.mask_part0) SLN_L128(4);
                    else if (p.residual) SLN_L128(5);
                    else if (p.mask_part0) SLN_L128(3);
                    else SLN_L128(1);
#undef SLN_L128
                    return sln_launch_status();
                }
                // 3-wide kernels, stride 1, on maps of 32 ... 256 columns in whole tiles: activation stages per kernel ROW
                const int Wr = p.segW[0];
                const bool row = nseg == 1 && KW == 3 && stride_h == 1 && stride_w == 1 && pad_left == dil_w && dil_w <= 8 &&
                                 p.segOW[0] == Wr && (Wr == 32 || Wr == 64 || Wr == 128 || Wr == 256) && M % T2 == 0 &&
                                 !(p.dbg & ~(8 | 16 | 256 | 8192 | 16384 | 32768 | 65536 | 131072)) &&
                                 sln_knob("SLN_CONV_TAPROW", 1) != 0;
                // ... and on maps of exactly one tile with at most 40 columns x dilation (the mask head's 16 x 16 rois): per kernel COLUMN
                const bool colm = !row && nseg == 1 && KH == 3 && stride_h == 1 && stride_w == 1 && pad_top == dil_h &&
                                  p.segOH[0] == p.segH[0] && p.segOW[0] == Wr && (long)p.segH[0] * Wr == T2 && dil_h * Wr <= 40 &&
                                  !(p.dbg & ~(8 | 16 | 256 | 8192 | 16384 | 32768 | 65536 | 131072)) &&
                                  sln_knob("SLN_CONV_TAPROW", 1) != 0;
                p.tapmode = row ? 1 : colm ? 2 : 0;
#define SLN_L256(E, R) hipLaunchKernelGGL((conv_fwd256h_kernel<false, 2, 16, E, R>), g2, b2, 0, (hipStream_t)stream, p)
                if (row || colm) {
                    sln_last_fwd_kernel = 4;
                    if (!w8) hipLaunchKernelGGL((conv_fwd256h_kernel<false, 2, 16, 0, true>), g2, b2, 0, (hipStream_t)stream, p);
                    else if (p.res_parts) SLN_L256(2, true);
                    else if (p.residual && p.mask_part0) SLN_L256(4, true);
                    else if (p.residual) SLN_L256(5, true);
                    else if (p.mask_part0) SLN_L256(3, true);
                    else SLN_L256(1, true);
                    return sln_launch_status();
                }
                if (!w8) hipLaunchKernelGGL((conv_fwd256h_kernel<false, 2, 16>), g2, b2, 0, (hipStream_t)stream, p);
                else if (p.res_parts) SLN_L256(2, false);
                else if (p.residual && p.mask_part0) SLN_L256(4, false);
                else if (p.residual) SLN_L256(5, false);
                else if (p.mask_part0) SLN_L256(3, false);
                else SLN_L256(1, false);
#undef SLN_L256
            }
            else hipLaunchKernelGGL((conv_fwd256h_kernel<false, 2>), g2, b2, 0, (hipStream_t)stream, p);
        }
        else if (parts == 2)
            hipLaunchKernelGGL(conv_fwd256_kernel<2>, dim3((unsigned)nb2), dim3(512), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(conv_fwd256_kernel<3>, dim3((unsigned)nb2), dim3(512), 0, (hipStream_t)stream, p);
        return sln_launch_status();
    }
    // 64-wide N tile when the whole output is at most 64 channels wide (SLN_CONV_BN64=0 disables)
    const bool narrow = Cout <= 64 && sln_knob("SLN_CONV_BN64", 1) != 0;
    p.gm = sln_div_up(M, BM);
    p.gn = sln_div_up(Cout, narrow ? 64 : BN);
    const long nblk = (long)p.gm * p.gn;
    if (nblk > 2147483647L) return SLN_ERR_UNSUPPORTED;
    const dim3 g((unsigned)nblk), b(256);
    const bool w8 = parts == 2 && epilogue_is_plain(p) && !(p.dbg & 16) && epilogue_is_w8(p) && !(p.dbg & 256) &&
                    sln_knob("SLN_CONV_EPI", 1) != 0;
    // (round 5) 3-wide kernels, stride 1, on maps whose rows are whole 128-pixel tiles: k-steps per kernel ROW
    // (conv_fwd_kernel's ROW3 instances; SLN_CONV_ROW3=0: the per-tap loop, A/B)
    const bool row3 = parts == 2 && w8 && nseg == 1 && KW == 3 && stride_h == 1 && stride_w == 1 && pad_left == dil_w &&
                      dil_w <= 8 && p.segOW[0] == p.segW[0] && p.segW[0] % BM == 0 && sln_knob("SLN_CONV_ROW3", 1) != 0;
    if (row3) {
        sln_last_fwd_kernel = 5;
        // SLN_CONV_ROW3_NARROW=1 (A/B): 64-wide tiles for 128 output channels as well (43 KB of LDS: three blocks per CU
        // instead of two, the activation rows staged by both column tiles)
        const bool narrow3 = narrow || (Cout <= 128 && sln_knob("SLN_CONV_ROW3_NARROW", 0) != 0);
        p.gn = sln_div_up(Cout, narrow3 ? 64 : BN);
        const dim3 g3((unsigned)((long)p.gm * p.gn));
#define SLN_LR3(E) do { if (narrow3) hipLaunchKernelGGL((conv_fwd_kernel<2, 64, E, true>), g3, b, 0, (hipStream_t)stream, p); \
                        else hipLaunchKernelGGL((conv_fwd_kernel<2, 128, E, true>), g3, b, 0, (hipStream_t)stream, p); } while (0)
        if (p.res_parts) SLN_LR3(2);
        else if (p.residual && p.mask_part0) SLN_LR3(4);
        else if (p.residual) SLN_LR3(5);
        else if (p.mask_part0) SLN_LR3(3);
        else SLN_LR3(1);
#undef SLN_LR3
        return sln_launch_status();
    }
    if (parts == 1 && narrow) hipLaunchKernelGGL((conv_fwd_kernel<1, 64>), g, b, 0, (hipStream_t)stream, p);
    else if (parts == 1) hipLaunchKernelGGL((conv_fwd_kernel<1, 128>), g, b, 0, (hipStream_t)stream, p);
    else if (parts == 2 && narrow && w8 && sln_knob("SLN_CONV_EPI64", 1) != 0) {
        // (round 5) the 64-wide tile with ONE epilogue kind per instance, like the 128-wide one below: the all-in-one
        // instance carries every variant behind run-time branches
        if (p.res_parts) hipLaunchKernelGGL((conv_fwd_kernel<2, 64, 2>), g, b, 0, (hipStream_t)stream, p);
        else if (p.residual && p.mask_part0) hipLaunchKernelGGL((conv_fwd_kernel<2, 64, 4>), g, b, 0, (hipStream_t)stream, p);
        else if (p.residual) hipLaunchKernelGGL((conv_fwd_kernel<2, 64, 5>), g, b, 0, (hipStream_t)stream, p);
        else if (p.mask_part0) hipLaunchKernelGGL((conv_fwd_kernel<2, 64, 3>), g, b, 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_fwd_kernel<2, 64, 1>), g, b, 0, (hipStream_t)stream, p);
    }
    else if (parts == 2 && narrow) hipLaunchKernelGGL((conv_fwd_kernel<2, 64>), g, b, 0, (hipStream_t)stream, p);
    else if (parts == 2 && w8) {       // the instance that carries only the epilogue this launch needs
        if (p.res_parts) hipLaunchKernelGGL((conv_fwd_kernel<2, 128, 2>), g, b, 0, (hipStream_t)stream, p);
        else if (p.residual && p.mask_part0) hipLaunchKernelGGL((conv_fwd_kernel<2, 128, 4>), g, b, 0, (hipStream_t)stream, p);
        else if (p.residual) hipLaunchKernelGGL((conv_fwd_kernel<2, 128, 5>), g, b, 0, (hipStream_t)stream, p);
        else if (p.mask_part0) hipLaunchKernelGGL((conv_fwd_kernel<2, 128, 3>), g, b, 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_fwd_kernel<2, 128, 1>), g, b, 0, (hipStream_t)stream, p);
    }
    else if (parts == 2) hipLaunchKernelGGL((conv_fwd_kernel<2, 128>), g, b, 0, (hipStream_t)stream, p);
    else if (narrow) hipLaunchKernelGGL((conv_fwd_kernel<3, 64>), g, b, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_fwd_kernel<3, 128>), g, b, 0, (hipStream_t)stream, p);
    return sln_launch_status();
}

extern "C" int sln_conv2d_fwd_f32(const uint16_t *x_parts, int N, int H, int W, int Cin,
                                  const uint16_t *w_parts, int w_layout, int parts, int Cout, int KH, int KW,
                                  int stride_h, int stride_w, int dil_h, int dil_w, int pad_top,
                                  int pad_left, int OH, int OW, const float *scale, const float *shift,
                                  const float *residual, int relu, float *y, uint16_t *y_parts,
                                  const float *x_scale, const float *w_scale, const float *y_q_scale,
                                  float *y_q_amax, int32_t *y_q_saturated, sln_stream_t stream) {
    // one image group; the caller's OH/OW fix the bottom/right padding
    if (N < 0 || H < 1 || W < 1 || OH < 1 || OW < 1 || KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1 ||
        dil_h < 1 || dil_w < 1) {
        sln_enter();
        return SLN_ERR_INVALID_ARG;
    }
    const int32_t seg[3] = {N, H, W};
    const int pad_bottom = (OH - 1) * stride_h + dil_h * (KH - 1) + 1 - H - pad_top;
    const int pad_right = (OW - 1) * stride_w + dil_w * (KW - 1) + 1 - W - pad_left;
    return sln_conv2d_fwd_ms_f32(x_parts, 1, seg, Cin, w_parts, w_layout, parts, Cout, KH, KW, stride_h, stride_w,
                                 dil_h, dil_w, pad_top, pad_left, pad_bottom, pad_right, scale, shift,
                                 residual, relu, nullptr, nullptr, y, y_parts, nullptr, x_scale, w_scale,
                                 y_q_scale, y_q_amax, y_q_saturated, nullptr, nullptr, nullptr, stream);
}

// Which weight-gradient kernel sln_conv2d_wgrad_f32 uses: 256 = conv_wgrad256_kernel (one (tap, 256 Cout
// x 256 Cin) tile per block, LDS-DMA pipeline, split-K sized to one round of the 256 CUs) when both
// channel counts fill >= 176 (>= 160 of the last 256) columns, the tap tiles fit one round and there
// are enough pixels; else 128.  SLN_WGRAD_TILE256 = 0 never, 1 (default) by this rule, 2 always.
// Which weight-gradient kernel the PROCESS's last sln_conv2d_wgrad_f32 call launched (weight gradients are launched from
// autograd's worker thread, the label is read from the caller's): 0 conv_wgrad_kernel (a block per tap), 1 the 256 x 256
// kernels, 2 conv_wgrad_kernel's ROW3 instances (a block per kernel row).
static std::atomic<int> sln_last_wgrad_kernel{0};
extern "C" int sln_conv_wgrad_last_kernel(void) { return sln_last_wgrad_kernel.load(); }

extern "C" int sln_conv_wgrad_tile(int64_t M, int Cout, int Cin, int taps, int parts) {
    const int mode = sln_knob("SLN_WGRAD_TILE256", 1);
    if (M < 1 || Cout < 1 || Cin < 1 || taps < 1 || parts == 1) return BM;      // (single fp16 part: the generic kernel)
    if (mode == 2) return T2;
    if (mode != 1) return BM;
    const long nt2 = (long)sln_div_up(Cout, T2) * sln_div_up(Cin, T2) * taps;
    const bool wide = Cout >= 176 && (Cout % T2 == 0 || Cout % T2 >= 160) && Cin >= 176 &&
                      (Cin % T2 == 0 || Cin % T2 >= 160);
    return (wide && nt2 <= 256 && M >= 256L * 64) ? T2 : BM;
}

extern "C" size_t sln_conv_wgrad_workspace_bytes(int64_t M, int Cout, int Cin, int taps, int parts) {
    if (M < 1 || Cout < 1 || Cin < 1 || taps < 1) return 0;
    const WgradPlan w = wgrad_plan(M, Cout, Cin, taps, parts);
    // (one slab even for a single pixel range: gw_layout 1 goes through the reduce pass)
    return sizeof(float) * (size_t)w.ksplit * Cout * taps * Cin;
}

extern "C" int sln_conv2d_wgrad_f32(const uint16_t *gz_parts, int Cout, int Cout_pad,
                                    const uint16_t *x_parts, int N, int H, int W, int Cin, int Cin_pad,
                                    int parts, int KH, int KW, int stride_h, int stride_w, int dil_h,
                                    int dil_w, int pad_top, int pad_left, int OH, int OW, float *gw,
                                    const float *gz_scale, const float *x_scale, void *workspace,
                                    size_t workspace_bytes, int gw_layout, sln_stream_t stream) {
    sln_enter();
    const bool defer = (gw_layout & SLN_WGRAD_DEFER_REDUCE) != 0;     // partial sums only: the caller batches the reduce
    gw_layout &= ~SLN_WGRAD_DEFER_REDUCE;
    if (gw_layout != 0 && gw_layout != 1) return SLN_ERR_INVALID_ARG;
    if (defer && (!workspace || N == 0)) return SLN_ERR_INVALID_ARG;
    if (!gz_parts || !x_parts || (!gw && !defer) || N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 ||
        OH < 1 || OW < 1 || Cin_pad < Cin || Cout_pad < Cout || (Cin_pad & 7) || (Cout_pad & 7))
        return SLN_ERR_INVALID_ARG;
    if (parts < 1 || parts > 3) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)N * OH * OW;
    const size_t gw_elems = (size_t)Cout * KH * KW * Cin;
    if (N == 0) return hipMemsetAsync(gw, 0, sizeof(float) * gw_elems, st) == hipSuccess ? SLN_OK : SLN_ERR_LAUNCH;
    if (M > 2147483647L - BK) return SLN_ERR_UNSUPPORTED;
    const WgradPlan w = wgrad_plan(M, Cout, Cin, KH * KW, parts);
    // two-phase (deterministic) when the caller lends a workspace; fp32 atomics into the zeroed gw else
    // gw_layout 1 ([Cout][Cin][KH][KW], the parameter's own order) is produced by the reduce pass: it needs the
    // workspace even for a single pixel range (1x1 kernels: the two orders coincide)
    const bool transpose = gw_layout == 1 && KH * KW > 1;
    if (transpose && !workspace) return SLN_ERR_WORKSPACE;
    const size_t need = (w.ksplit > 1 || transpose || defer) ? sizeof(float) * (size_t)w.ksplit * gw_elems : 0;
    const bool two_phase = workspace != nullptr && (w.ksplit > 1 || transpose || defer);
    if (two_phase && workspace_bytes < need) return SLN_ERR_WORKSPACE;
    const bool direct = w.ksplit == 1 && workspace != nullptr && !transpose;   // a single pixel range: plain stores into gw
    if (!two_phase && !direct &&
        hipMemsetAsync(gw, 0, sizeof(float) * gw_elems, st) != hipSuccess)
        return SLN_ERR_LAUNCH;
    WgradParams p;
    p.gz = (const __bf16 *)gz_parts; p.x = (const __bf16 *)x_parts; p.gw = gw;
    p.gz_scale = gz_scale; p.x_scale = x_scale;
    p.partial = two_phase ? (float *)workspace : (direct ? gw : nullptr);
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cip = Cin_pad; p.Cout = Cout; p.Cop = Cout_pad;
    p.KH = KH; p.KW = KW; p.sh = stride_h; p.sw = stride_w; p.dh = dil_h; p.dw = dil_w;
    p.pt = pad_top; p.pl = pad_left; p.OH = OH; p.OW = OW;
    p.M = (int)M;
    p.gz_part_stride = M * Cout_pad;
    p.x_part_stride = (long)N * H * W * Cin_pad;
    p.gm = w.gm; p.gn_per_tap = w.gn; p.ksplit = w.ksplit; p.pix_per_split = w.pps;
    p.xcd_wgrad = sln_knob("SLN_WGRAD_XCD", 3);      // bit 0: the 256^2 kernels, bit 1: the 128^2 kernel
    const long nblk = (long)w.gm * w.gn * KH * KW * w.ksplit;
    if (nblk > 2147483647L) return SLN_ERR_UNSUPPORTED;
    sln_last_wgrad_kernel = w.tile == T2 ? 1 : 0;
    if (w.tile == T2) {
        // the fp16 kernel addresses with 32-bit byte offsets and divides pixel indices in fp32
        const bool h = parts == 2 && sln_knob("SLN_WGRAD_F16_KERNEL", 1) && M < (1L << 24) &&
                       p.gz_part_stride * 2 < 4294967295L && p.x_part_stride * 2 < 4294967295L;
        if (h)
            if (sln_knob("SLN_WGRAD_MFMA16", 0))      // (same-box A/B on the train step: no difference, 85.6 / 85.7 img/s)
                hipLaunchKernelGGL(conv_wgrad256h_kernel<16>, dim3((unsigned)nblk), dim3(512), 0, st, p);
            else
                hipLaunchKernelGGL(conv_wgrad256h_kernel<32>, dim3((unsigned)nblk), dim3(512), 0, st, p);
        else if (parts == 2)
            hipLaunchKernelGGL(conv_wgrad256_kernel<2>, dim3((unsigned)nblk), dim3(512), 0, st, p);
        else
            hipLaunchKernelGGL(conv_wgrad256_kernel<3>, dim3((unsigned)nblk), dim3(512), 0, st, p);
    } else {
        // (round 5) 3-wide kernels, stride 1, rows of whole 32-pixel k-steps: a block per kernel ROW (conv_wgrad_kernel's
        // ROW3 instances, 64-wide Cin tiles; the pixel ranges and slabs are the plan's; SLN_WGRAD_ROW3=0: per tap, A/B)
        const bool row3 = parts == 2 && KW == 3 && stride_h == 1 && stride_w == 1 && OW == W && OW % BK == 0 &&
                          pad_left == dil_w && dil_w <= 8 && w.pps % BK == 0 && sln_knob("SLN_WGRAD_ROW3", 1) != 0;
        if (row3) {
            sln_last_wgrad_kernel = 2;
            p.gn_per_tap = sln_div_up(Cin, 64);
            // SLN_WGRAD_ROW3_TM64=1 (A/B): 64 x 64 tiles whatever Cout (more, lighter blocks)
            const bool tm64 = w.tm == 64 || sln_knob("SLN_WGRAD_ROW3_TM64", 0) != 0;
            if (tm64) p.gm = sln_div_up(Cout, 64);
            const long nb3 = (long)p.gm * p.gn_per_tap * KH * w.ksplit;
            const dim3 g3((unsigned)nb3), b3(256);
            if (tm64) hipLaunchKernelGGL((conv_wgrad_kernel<2, 64, 64, true>), g3, b3, 0, st, p);
            else hipLaunchKernelGGL((conv_wgrad_kernel<2, 128, 64, true>), g3, b3, 0, st, p);
            if (two_phase && !defer)
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((gw_elems + 127) / 128)), dim3(256), 0, st,
                                   (const float *)workspace, w.ksplit, (long)gw_elems, gw, transpose ? KH * KW : 0, Cin);
            return sln_launch_status();
        }
        const dim3 g((unsigned)nblk), b(256);
        const int TMs = w.tm, TNs = w.tn;
#define SLN_WG(PP, A, B) hipLaunchKernelGGL((conv_wgrad_kernel<PP, A, B>), g, b, 0, st, p)
        if (parts == 1) {
            if (TMs == 64 && TNs == 64) SLN_WG(1, 64, 64); else if (TMs == 64) SLN_WG(1, 64, 128);
            else if (TNs == 64) SLN_WG(1, 128, 64); else SLN_WG(1, 128, 128);
        } else if (parts == 2) {
            if (TMs == 64 && TNs == 64) SLN_WG(2, 64, 64); else if (TMs == 64) SLN_WG(2, 64, 128);
            else if (TNs == 64) SLN_WG(2, 128, 64); else SLN_WG(2, 128, 128);
        } else {
            if (TMs == 64 && TNs == 64) SLN_WG(3, 64, 64); else if (TMs == 64) SLN_WG(3, 64, 128);
            else if (TNs == 64) SLN_WG(3, 128, 64); else SLN_WG(3, 128, 128);
        }
#undef SLN_WG
    }
    if (two_phase && !defer)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((gw_elems + 127) / 128)), dim3(256), 0, st,
                           (const float *)workspace, w.ksplit, (long)gw_elems, gw, transpose ? KH * KW : 0, Cin);
    return sln_launch_status();
}

extern "C" int sln_conv_wgrad_ksplit(int64_t M, int Cout, int Cin, int taps, int parts) {
    if (M < 1 || Cout < 1 || Cin < 1 || taps < 1) return 0;
    return wgrad_plan(M, Cout, Cin, taps, parts).ksplit;
}

extern "C" int sln_wgrad_reduce_batch_f32(const sln_wgrad_reduce_desc_t *descs, int n, sln_stream_t stream) {
    sln_enter();
    if (n < 0 || n > SLN_WGRAD_REDUCE_BATCH || (n > 0 && !descs)) return SLN_ERR_INVALID_ARG;
    if (n == 0) return SLN_OK;
    WgradReduceBatch b;
    b.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const sln_wgrad_reduce_desc_t &d = descs[i];
        if (!d.partial || !d.gw || d.n < 1 || d.ksplit < 1 || d.taps < 0 || d.Cin < 1) return SLN_ERR_INVALID_ARG;
        b.d[i] = d;
        b.first[i] = blocks;
        blocks += (d.n + 127) / 128;
    }
    b.first[n] = blocks;
    if (blocks > 2147483647L) return SLN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    return sln_launch_status();
}
