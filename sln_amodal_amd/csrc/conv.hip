// Implicit-GEMM convolution for gfx950 on the bf16 matrix cores with fp32-class
// accuracy ("split-bf16"): every fp32 operand x is represented as a sum of P bf16
// parts, x ~= x0 + x1 (+ x2), xi = bf16_rne(x - x0 - ... - x(i-1)); the product
// A*B is accumulated in fp32 from the dominant part-pairs:
//   P = 2 : a0*b0 + a0*b1 + a1*b0                     (rel. error ~4e-6 per layer)
//   P = 3 : ... + a1*b1 + a0*b2 + a2*b0               (rel. error ~1e-7, fp32 class)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32-input MFMA
// (MI355X_MICROARCH.md), so 3 / 6 bf16 MFMAs per fp32-equivalent product still
// leave 5.3x / 2.7x the fp32-MFMA roofline.  Accumulation is fp32 inside the MFMA.
//
// GEMM view (NHWC activations): M = N*OH*OW output pixels, N = Cout,
// K = KH*KW*Cin walked tap by tap in 32-channel chunks.
//   A[m][k] = x[n, oh*s - pad + kh*d, ow*s - pad + kw*d, ci]   (fp32 in HBM, split
//             on the fly while staging to LDS; out-of-image taps are zero = the
//             reference's SamePad2d / conv padding)
//   B[k][n] = w[cout][kh][kw][ci], pre-split once per weight update into
//             [P][Cout][KH*KW*Cin] bf16 (sln_conv_split_weights_f32).
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of
// 32x32 (64 accumulator VGPRs).  LDS rows are 32 bf16 + 8 pad (80 B): the
// ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct 4-bank groups
// (conflict-free).  One LDS stage + register prefetch of the next k-step.
// Epilogue: y = relu?( acc*scale[c] + shift[c] + residual ) -- bias and the frozen
// BatchNorm affine are folded into scale/shift by the caller.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BM 128
#define BN 128
#define BK 32
#define LDK 40  // padded row length in bf16 (80 B)

struct ConvParams {
    const float *x;
    const __bf16 *w;      // [P][Cout][Ktot]
    const float *scale;   // [Cout] or null (=1)
    const float *shift;   // [Cout] or null (=0)
    const float *residual;  // [M][Cout] or null
    float *y;             // [M][Cout]
    long w_part_stride;
    int N, H, W, Cin, Cout, KH, KW, sh, sw, dh, dw, pt, pl, OH, OW, relu;
    int M, Ktot, cin_chunks, gm, gn;
};

__device__ __forceinline__ void split4(const float4 v, bf16x4 *parts, int P) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (p >= P) break;
        bf16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)r[j];
            r[j] -= (float)h[j];
        }
        parts[p] = h;
    }
}

template <int P>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[P][BM][LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[P][BN][LDK];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs; give each
    // XCD a contiguous run of tiles so the Cout-tiles of one pixel block share an L2.
    const int nblk = p.gm * p.gn;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = (bid / p.gn) * BM;
    const int n0 = (bid % p.gn) * BN;

    // ---- per-thread A rows: 4 rows (t>>3)+32*i, channel quad (t&7)*4 ----
    const int acol = (t & 7) * 4;
    int a_ih0[4], a_iw0[4];
    long a_nbase[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (t >> 3) + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / (p.OH * p.OW);
        const int rem = mm - n * (p.OH * p.OW);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        a_ih0[i] = oh * p.sh - p.pt;
        a_iw0[i] = ow * p.sw - p.pl;
        a_nbase[i] = (long)n * p.H * p.W;
    }
    // ---- per-thread B row: cout n0 + (t>>1), k half (t&1)*16 ----
    const int brow = t >> 1, bhalf = (t & 1) * 16;
    const bool b_ok = (n0 + brow) < p.Cout;
    const __bf16 *bptr = p.w + (long)(n0 + brow) * p.Ktot;

    float4 ra[4];
    bf16x8 rb[P][2];
    const int nk = p.KH * p.KW * p.cin_chunks;

    auto load_tile = [&](int ks) {
        const int tap = ks / p.cin_chunks;
        const int ci0 = (ks - tap * p.cin_chunks) * BK;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int ci = ci0 + acol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = a_ih0[i] + kh * p.dh, iw = a_iw0[i] + kw * p.dw;
            const bool ok = a_ok[i] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && ci < p.Cin;
            ra[i] = ok ? *(const float4 *)(p.x + ((a_nbase[i] + (long)ih * p.W + iw) * p.Cin + ci))
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const long koff = (long)tap * p.Cin + ci0 + bhalf;
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = b_ok && (ci0 + bhalf + 8 * q) < p.Cin;
                bf16x8 z = {};
                rb[pp][q] = ok ? *(const bf16x8 *)(bptr + pp * p.w_part_stride + koff + 8 * q) : z;
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 parts[3];
            split4(ra[i], parts, P);
            const int row = (t >> 3) + 32 * i;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) *(bf16x4 *)&sA[pp][row][acol] = parts[pp];
        }
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < 2; ++q) *(bf16x8 *)&sB[pp][brow][bhalf + 8 * q] = rb[pp][q];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile();
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_tile(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2][P], b[2][P];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    a[i][pp] = *(const bf16x8 *)&sA[pp][wr * 64 + i * 32 + frow][kk * 16 + fk];
                    b[i][pp] = *(const bf16x8 *)&sB[pp][wc * 64 + i * 32 + frow][kk * 16 + fk];
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    if (P == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();
        if (ks + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wc * 64 + j * 32 + (lane & 31);
        if (c >= p.Cout) continue;
        const float sc = p.scale ? p.scale[c] : 1.0f;
        const float sf = p.shift ? p.shift[c] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = acc[i][j][r] * sc + sf;
                const long o = (long)m * p.Cout + c;
                if (p.residual) v += p.residual[o];
                if (p.relu) v = fmaxf(v, 0.0f);
                p.y[o] = v;
            }
    }
}

// ---- weight preparation: fp32 weights (any strides) -> [P][O][KH][KW][I] bf16 parts.
// flip=1 mirrors the taps (used, with O/I swapped through the strides, to express
// the data-gradient convolution as a forward convolution).
__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ w, int O, int I,
                                                            int KH, int KW, long s_o, long s_i,
                                                            long s_kh, long s_kw, int flip, int P,
                                                            __bf16 *__restrict__ out) {
    const long total = (long)O * KH * KW * I;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % I);
        long r = idx / I;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH);
        const int o = (int)(r / KH);
        const int skh = flip ? KH - 1 - kh : kh, skw = flip ? KW - 1 - kw : kw;
        float v = w[o * s_o + i * s_i + skh * s_kh + skw * s_kw];
        for (int pp = 0; pp < P; ++pp) {
            const __bf16 h = (__bf16)v;
            out[(long)pp * total + idx] = h;
            v -= (float)h;
        }
    }
}

extern "C" int sln_conv_split_weights_f32(const float *w, int O, int I, int KH, int KW, long s_o,
                                          long s_i, long s_kh, long s_kw, int flip, int parts,
                                          uint16_t *out, sln_stream_t stream) {
    sln_enter();
    if (!w || !out || O < 1 || I < 1 || KH < 1 || KW < 1 || parts < 2 || parts > 3) return SLN_ERR_INVALID_ARG;
    const long total = (long)O * KH * KW * I;
    int gx = sln_div_up(total, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(split_weights_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, w, O, I, KH, KW,
                       s_o, s_i, s_kh, s_kw, flip, parts, (__bf16 *)out);
    return sln_launch_status();
}

extern "C" int sln_conv2d_fwd_f32(const float *x, int N, int H, int W, int Cin, const uint16_t *w_parts,
                                  int parts, int Cout, int KH, int KW, int stride_h, int stride_w,
                                  int dil_h, int dil_w, int pad_top, int pad_left, int OH, int OW,
                                  const float *scale, const float *shift, const float *residual,
                                  int relu, float *y, sln_stream_t stream) {
    sln_enter();
    if (!x || !w_parts || !y || N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 ||
        stride_h < 1 || stride_w < 1 || dil_h < 1 || dil_w < 1 || OH < 1 || OW < 1)
        return SLN_ERR_INVALID_ARG;
    if (parts != 2 && parts != 3) return SLN_ERR_INVALID_ARG;
    if (Cin % 8 != 0) return SLN_ERR_UNSUPPORTED;  // 16-B vector loads along channels
    if (N == 0) return SLN_OK;
    ConvParams p;
    p.x = x; p.w = (const __bf16 *)w_parts; p.scale = scale; p.shift = shift; p.residual = residual;
    p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.sh = stride_h; p.sw = stride_w; p.dh = dil_h; p.dw = dil_w; p.pt = pad_top; p.pl = pad_left;
    p.OH = OH; p.OW = OW; p.relu = relu;
    const long M = (long)N * OH * OW;
    if (M > 2147483647L - BM) return SLN_ERR_UNSUPPORTED;
    p.M = (int)M;
    p.Ktot = KH * KW * Cin;
    p.w_part_stride = (long)Cout * p.Ktot;
    p.cin_chunks = sln_div_up(Cin, BK);
    p.gm = sln_div_up(M, BM);
    p.gn = sln_div_up(Cout, BN);
    const long nblk = (long)p.gm * p.gn;
    if (nblk > 2147483647L) return SLN_ERR_UNSUPPORTED;
    if (parts == 2)
        hipLaunchKernelGGL(conv_fwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv_fwd_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    return sln_launch_status();
}
