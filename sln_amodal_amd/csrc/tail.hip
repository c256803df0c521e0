// Inference tail on the device (gfx950): full-size binary masks from the 32x32 head outputs, and
// their COCO run-length encoding, so that evaluation stays in HBM up to the COCO-json boundary.
//
//   unmold_masks_kernel  utils.py:447-465 (unmold_mask) for every detection of an image at once:
//       scipy.misc.imresize(mask, (y2-y1, x2-x1), 'bilinear') / 255 >= 0.5 pasted at the box.
//       imresize = min-max "bytescale" to uint8 in float32 + Pillow's 8-bit BILINEAR resample
//       (libImaging/Resample.c: double-precision tap weights rounded to 22-bit fixed point, a
//       horizontal pass into a uint8 temporary, then a vertical pass).  Both passes are restated
//       exactly, so the masks are bit-identical to the reference's.  The output is written
//       column-major ([N, W, H], y fastest) -- the np.asfortranarray layout the reference hands to
//       pycocotools (amodal_train.py:397) -- so the run-length kernel streams it.
//   rle_encode_kernel    cocoapi/common/maskApi.c:33-42 (rleEncode), one 1024-thread block per mask.
//   sln_rle_to_string    maskApi.c:204-216 (rleToString), host side: a few thousand counts per mask.
//
// Compiled with -ffp-contract=off: the tap weights must round once per operation like Pillow's
// x86-64 build.
#include "common.h"
#include <string.h>

#define UM_THREADS 256
#define UM_COLS 64            // output columns (contiguous H-byte lines) per block
#define UM_MAX_M 64           // head mask side limit (the reference's is 32 after the deconv)
#define UM_YT 1024            // rows per vertical tile: 4 per thread
#define UM_PRECISION_BITS 22  // Resample.c: 32 - 8 - 2

__device__ __forceinline__ double um_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output sample xx of an axis resized
// in_size -> out_size over the whole input.  kk[0..ksize) receives the fixed-point taps (zero padded),
// the return value is the first input tap.
__device__ int um_coeffs(int in_size, int out_size, int xx, int ksize, int *kk, int *count) {
    const float in0 = 0.0f, in1 = (float)in_size;
    const double scale = (double)(in1 - in0) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = in0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += um_bilinear((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            double w = um_bilinear((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            v = w < 0 ? (int)(-0.5 + w * (1 << UM_PRECISION_BITS)) : (int)(0.5 + w * (1 << UM_PRECISION_BITS));
        }
        kk[x] = v;
    }
    *count = xmax;
    return xmin;
}

__device__ __forceinline__ int um_ksize(int in_size, int out_size) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double support = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(support) * 2 + 1;
}

__device__ __forceinline__ unsigned um_clip8(int v) {
    v >>= UM_PRECISION_BITS;
    return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Zero `n` bytes at p (block-cooperative).
__device__ void um_zero(uint8_t *p, long n) {
    const int t = threadIdx.x;
    long head = (long)((16 - ((uintptr_t)p & 15)) & 15);
    if (head > n) head = n;
    for (long i = t; i < head; i += UM_THREADS) p[i] = 0;
    uint4 *q = (uint4 *)(p + head);
    const long nv = (n - head) >> 4;
    for (long i = t; i < nv; i += UM_THREADS) q[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long i = head + (nv << 4) + t; i < n; i += UM_THREADS) p[i] = 0;
}

// grid (ceil(W / UM_COLS), N).  LDS: the bytescaled head mask, the horizontally resampled strip
// [mh][UM_COLS] and the tap tables of the strip's columns / the current tile's rows.
__global__ __launch_bounds__(UM_THREADS) void unmold_masks_kernel(
    const float *__restrict__ masks, const int32_t *__restrict__ class_ids,
    const int32_t *__restrict__ boxes, int C, int mh, int mw, int H, int W, uint8_t *__restrict__ full) {
    __shared__ uint8_t s_byt[UM_MAX_M * UM_MAX_M];
    __shared__ uint8_t s_tmp[UM_MAX_M * UM_COLS];
    __shared__ int s_k[3 * UM_YT > 5 * UM_MAX_M + 3 * UM_COLS ? 3 * UM_YT : 5 * UM_MAX_M + 3 * UM_COLS];
    __shared__ int s_min[UM_YT];
    __shared__ float s_red[2 * (UM_THREADS / SLN_WAVE)];
    const int t = threadIdx.x, n = blockIdx.y;
    const int x0 = blockIdx.x * UM_COLS;
    const int ncols = min(UM_COLS, W - x0);
    uint8_t *out = full + ((long)n * W + x0) * H;      // the strip is one contiguous byte range
    const int y1 = boxes[n * 4 + 0], x1 = boxes[n * 4 + 1], y2 = boxes[n * 4 + 2], x2 = boxes[n * 4 + 3];
    const int oh = y2 - y1, ow = x2 - x1;
    const bool valid = oh > 0 && ow > 0 && y1 >= 0 && x1 >= 0 && y2 <= H && x2 <= W;
    const int cx_lo = max(x0, x1), cx_hi = min(x0 + ncols, x2);   // strip columns inside the box
    if (!valid || cx_lo >= cx_hi) {
        um_zero(out, (long)ncols * H);
        return;
    }
    // ---- bytescale (scipy.misc.pilutil.bytescale, float32) ----
    int cls = class_ids ? class_ids[n] : 0;
    const float *m = masks + ((long)n * C + cls) * mh * mw;
    const int npx = mh * mw;
    float lo = m[0], hi = m[0];
    for (int i = t; i < npx; i += UM_THREADS) {
        const float v = m[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((t & 63) == 0) { s_red[t >> 6] = lo; s_red[4 + (t >> 6)] = hi; }
    __syncthreads();
    lo = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
    hi = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
    float cscale = hi - lo;
    if (cscale == 0.0f) cscale = 1.0f;
    const float scale = (float)(255.0 / (double)cscale);
    for (int i = t; i < npx; i += UM_THREADS) {
        float v = (m[i] - lo) * scale;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        s_byt[i] = (uint8_t)(v + 0.5f);
    }
    // ---- horizontal pass for the strip's box columns -> s_tmp[r][c] ----
    const int ksh = um_ksize(mw, ow);
    const int nin = cx_hi - cx_lo;                       // <= UM_COLS; ksh > 3 only when ow < mw
    if (t < nin) {
        int cnt;
        s_min[t] = um_coeffs(mw, ow, cx_lo - x1 + t, ksh, &s_k[t * ksh], &cnt);
    }
    __syncthreads();
    for (int i = t; i < mh * nin; i += UM_THREADS) {
        const int r = i / nin, c = i - r * nin;
        const int *k = &s_k[c * ksh];
        const uint8_t *row = &s_byt[r * mw + s_min[c]];
        int acc = 1 << (UM_PRECISION_BITS - 1);
        for (int x = 0; x < ksh; ++x) {
            const int kx = k[x];
            if (kx) acc += (int)row[x] * kx;             // taps past the count are zero-padded
        }
        s_tmp[r * UM_COLS + c] = (uint8_t)um_clip8(acc);
    }
    __syncthreads();
    // ---- columns of the strip outside the box ----
    if (cx_lo > x0) um_zero(out, (long)(cx_lo - x0) * H);
    if (cx_hi < x0 + ncols) um_zero(out + (long)(cx_hi - x0) * H, (long)(x0 + ncols - cx_hi) * H);
    // ---- vertical pass, threshold, paste: tiles of UM_YT rows, 4 consecutive rows per thread ----
    const int ksv = um_ksize(mh, oh);
    const bool packed = (H & 3) == 0;
    for (int ty = 0; ty < H; ty += UM_YT) {
        __syncthreads();
        for (int j = 0; j < 4; ++j) {
            const int ry = 4 * t + j, y = ty + ry;
            if (y >= y1 && y < y2 && y < H) {
                int cnt;
                s_min[ry] = um_coeffs(mh, oh, y - y1, ksv, &s_k[(ry - max(0, y1 - ty)) * ksv], &cnt);
            }
        }
        __syncthreads();
        const int ybase = ty + 4 * t;
        if (ybase >= H) continue;
        for (int c = 0; c < nin; ++c) {
            unsigned word = 0u;
            for (int j = 0; j < 4; ++j) {
                const int ry = 4 * t + j, y = ty + ry;
                unsigned bit = 0u;
                if (y >= y1 && y < y2) {
                    const int *k = &s_k[(ry - max(0, y1 - ty)) * ksv];
                    const uint8_t *col = &s_tmp[s_min[ry] * UM_COLS + c];
                    int acc = 1 << (UM_PRECISION_BITS - 1);
                    for (int x = 0; x < ksv; ++x) {
                        const int kx = k[x];
                        if (kx) acc += (int)col[x * UM_COLS] * kx;
                    }
                    // (r.astype(float32) / 255.0) >= 0.5  <=>  r >= 128
                    bit = ((float)um_clip8(acc) / 255.0f) >= 0.5f ? 1u : 0u;
                }
                word |= bit << (8 * j);
            }
            uint8_t *dst = out + (long)(cx_lo - x0 + c) * H + ybase;
            if (packed) {
                *(unsigned *)dst = word;
            } else {
                for (int j = 0; j < 4 && ybase + j < H; ++j) dst[j] = (uint8_t)(word >> (8 * j));
            }
        }
    }
}

extern "C" int sln_unmold_masks_u8(const float *masks, const int32_t *class_ids, const int32_t *boxes,
                                   int N, int C, int mh, int mw, int H, int W, uint8_t *full,
                                   sln_stream_t stream) {
    if (N < 0 || C <= 0 || mh <= 0 || mw <= 0 || H <= 0 || W <= 0) return SLN_ERR_INVALID_ARG;
    if (mh > UM_MAX_M || mw > UM_MAX_M) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!masks || !boxes || !full) return SLN_ERR_INVALID_ARG;
    sln_enter();
    dim3 grid(sln_div_up(W, UM_COLS), N);
    hipLaunchKernelGGL(unmold_masks_kernel, grid, dim3(UM_THREADS), 0, (hipStream_t)stream, masks, class_ids, boxes, C,
                       mh, mw, H, W, full);
    return sln_launch_status();
}

// ----------------------------------------------------------------------------------------------
// Run-length encoding (maskApi.c:33-42).  A run boundary sits at every j with mask[j] != mask[j-1]
// (mask[-1] := 0), so  counts = diff([0, boundaries..., a]).  Pass 1 writes the boundary positions
// in order (16 bytes per thread, block scan of the per-thread boundary counts); pass 2 turns
// positions into differences in place, from the top chunk down so that every predecessor is read
// before it is overwritten.
// ----------------------------------------------------------------------------------------------
#define RLE_THREADS 1024

__global__ __launch_bounds__(RLE_THREADS) void rle_encode_kernel(const uint8_t *__restrict__ masks, long a,
                                                                 int max_runs, uint32_t *__restrict__ counts,
                                                                 int32_t *__restrict__ num_runs) {
    __shared__ unsigned s_wave[RLE_THREADS / SLN_WAVE];
    __shared__ unsigned s_total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint8_t *mk = masks + (long)blockIdx.x * a;
    uint32_t *pos = counts + (long)blockIdx.x * max_runs;
    const bool aligned = (((uintptr_t)mk) & 15) == 0;
    unsigned carry = 0u;
    for (long base = 0; base < a; base += (long)RLE_THREADS * 16) {
        const long j0 = base + (long)t * 16;
        unsigned char b[16];
        unsigned flags = 0u;
        if (j0 < a) {
            const int nb = (int)min(16L, a - j0);
            if (aligned && nb == 16) {
                const uint4 v = *(const uint4 *)(mk + j0);
                memcpy(b, &v, 16);
            } else {
                for (int i = 0; i < 16; ++i) b[i] = i < nb ? mk[j0 + i] : 0;
            }
            unsigned char prev = j0 ? mk[j0 - 1] : 0;
            for (int i = 0; i < nb; ++i) {
                flags |= (unsigned)(b[i] != prev) << i;
                prev = b[i];
            }
        }
        const unsigned cnt = __popc(flags);
        unsigned inc = cnt;                                // inclusive wave scan
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        if (t == 0) {
            unsigned run = 0u;
            for (int w = 0; w < RLE_THREADS / SLN_WAVE; ++w) {
                const unsigned c = s_wave[w];
                s_wave[w] = run;
                run += c;
            }
            s_total = run;
        }
        __syncthreads();
        unsigned k = carry + s_wave[wave] + inc - cnt;
        while (flags) {
            const int i = __ffs(flags) - 1;
            flags &= flags - 1;
            if (k < (unsigned)max_runs) pos[k] = (uint32_t)(j0 + i);
            ++k;
        }
        carry += s_total;
        __syncthreads();
    }
    const unsigned T = carry;                              // boundaries; runs = T + 1
    if (t == 0) num_runs[blockIdx.x] = (int32_t)(T + 1u);
    if (T + 1u > (unsigned)max_runs) return;               // caller re-runs with a larger capacity
    __syncthreads();
    for (long top = (long)T; top >= 0; top -= RLE_THREADS) {
        const long k = top - t;
        uint32_t cur = 0u, prv = 0u;
        if (k >= 0) {
            cur = k == (long)T ? (uint32_t)a : pos[k];
            prv = k == 0 ? 0u : pos[k - 1];
        }
        __syncthreads();
        if (k >= 0) pos[k] = cur - prv;
    }
}

extern "C" int sln_rle_encode_u8(const uint8_t *masks, int N, int64_t a, int max_runs, uint32_t *counts,
                                 int32_t *num_runs, sln_stream_t stream) {
    if (N < 0 || a < 0 || a > 0xffffffffLL || max_runs < 1) return SLN_ERR_INVALID_ARG;
    if (N == 0) return SLN_OK;
    if (!counts || !num_runs || (a > 0 && !masks)) return SLN_ERR_INVALID_ARG;
    sln_enter();
    hipLaunchKernelGGL(rle_encode_kernel, dim3(N), dim3(RLE_THREADS), 0, (hipStream_t)stream, masks, (long)a, max_runs,
                       counts, num_runs);
    return sln_launch_status();
}

// Host side: compressed string of one mask's counts (maskApi.c:204-216).  Values from the fourth
// on are differences to the count two places back; 5 payload bits per character, low group first,
// 0x20 = "another group follows", offset 48.  Returns the length written (no terminator), or
// -SLN_ERR_INVALID_ARG when `cap` is too small (6 characters per count always suffice).
extern "C" int64_t sln_rle_to_string(const uint32_t *counts, int64_t m, char *out, int64_t cap) {
    if (m < 0 || (m > 0 && (!counts || !out))) return -SLN_ERR_INVALID_ARG;
    int64_t p = 0;
    for (int64_t i = 0; i < m; ++i) {
        long long x = (long long)counts[i];
        if (i > 2) x -= (long long)counts[i - 2];
        for (;;) {
            int c = (int)(x & 0x1f);
            x >>= 5;
            const bool more = (c & 0x10) ? (x != -1) : (x != 0);
            if (more) c |= 0x20;
            if (p >= cap) return -SLN_ERR_INVALID_ARG;
            out[p++] = (char)(c + 48);
            if (!more) break;
        }
    }
    return p;
}

// The same for the N masks of a batch in ONE call (round 6: the evaluation hand-off runs on a worker thread next to
// the next batch's forward; a ctypes call releases the interpreter lock for its whole duration, a Python loop of N
// small calls does not): row n of `counts` (row pitch `row_stride` words) holds num_runs[n] counts; the strings are
// written back to back into `out` and offsets[n] .. offsets[n + 1] delimits string n (offsets has N + 1 entries).
// Returns the total length, or -SLN_ERR_INVALID_ARG (cap too small: 6 characters per count always suffice).
extern "C" int64_t sln_rle_to_strings(const uint32_t *counts, int64_t row_stride, const int32_t *num_runs, int N,
                                      char *out, int64_t cap, int64_t *offsets) {
    if (N < 0 || row_stride < 0 || (N > 0 && (!counts || !num_runs || !out || !offsets))) return -SLN_ERR_INVALID_ARG;
    int64_t p = 0;
    for (int n = 0; n < N; ++n) {
        offsets[n] = p;
        const int64_t m = num_runs[n];
        if (m < 0 || m > row_stride) return -SLN_ERR_INVALID_ARG;
        const int64_t w = sln_rle_to_string(counts + (int64_t)n * row_stride, m, out + p, cap - p);
        if (w < 0) return w;
        p += w;
    }
    if (N > 0) offsets[N] = p;
    return p;
}

// Host side inverse (maskApi.c:218-231): counts from the compressed string.  Returns the number of
// counts, or -SLN_ERR_INVALID_ARG when `cap` is too small (one count per character always suffices).
extern "C" int64_t sln_rle_from_string(const char *s, int64_t len, uint32_t *counts, int64_t cap) {
    if (len < 0 || (len > 0 && (!s || !counts))) return -SLN_ERR_INVALID_ARG;
    int64_t m = 0, p = 0;
    while (p < len) {
        long long x = 0;
        int k = 0;
        for (;;) {
            if (p >= len) return -SLN_ERR_INVALID_ARG;              // truncated group
            const int c = (int)s[p++] - 48;
            x |= (long long)(c & 0x1f) << (5 * k);
            ++k;
            if (!(c & 0x20)) {
                if (c & 0x10) x |= (long long)(~0ULL << (5 * k));
                break;
            }
        }
        if (m > 2) x += (long long)counts[m - 2];
        if (m >= cap) return -SLN_ERR_INVALID_ARG;
        counts[m++] = (uint32_t)x;
    }
    return m;
}
