// Batched top-k selection in visiting order for the proposal front end (gfx950).
//
// The reference sorts all A = 261 888 foreground scores of an image and keeps the first 6000
// (modal/Functions.py:133-147: scores.sort(descending=True), then [:pre_nms_limit]).  Here the rest is
// never sorted, and an image is spread over the whole chip (round 2: the first version ran one block per
// image, 16 CUs busy for 0.93 ms):
//   1. radix select of the k-th largest key in three passes (11 + 11 + 10 bits): every 8192-element
//      segment of every image is a block that histograms its digit in LDS and adds the non-empty bins to
//      the image's global histogram; a pass re-derives the digits chosen so far from the earlier
//      histograms (every block redundantly: 2048 bins, one small scan);
//   2. per segment, the count of elements above the threshold T and of ties (== T);
//   3. scatter: elements above T anywhere into [0, n_gt) (order irrelevant, they are sorted next), the
//      first `ties` elements equal to T -- lowest index first: ordered ballot ranks inside a segment on
//      top of the preceding segments' tie counts -- behind them: exactly k candidates per image;
//   4. one block per image sorts its k candidates in LDS (bitonic, 256 ... 8192 slots) by (score
//      descending, index ascending) and writes the anchor indices.
// The result is exactly torch.sort(..., descending=True, stable=True)[1][:k] (ties broken by the lower
// index; the reference's sort is unstable, so ties are unspecified there).  Keys are the usual
// order-preserving integer image of the floats (NaN above +inf, like torch; -0.0 == +0.0).
#include "common.h"

#define TK_SEG 8192           // elements per segment block
#define TK_T 256              // threads of the segment kernels
#define TK_BINS 2048
#define TK_SORT_T 1024
#define TK_MAX_K 8192

__device__ __forceinline__ unsigned tk_key(float v) {
    unsigned u = __float_as_uint(v);
    if ((u << 1) == 0u) u = 0u;          // -0.0 compares equal to +0.0 in a sort: one key for both
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned tk_digit(unsigned key, int pass) {
    return pass == 0 ? key >> 21 : (pass == 1 ? (key >> 10) & 2047u : key & 1023u);
}

// Block-wide (TK_T threads): the digit d of `hist` (TK_BINS bins, scanned from the top) in which the
// need-th largest element lies, and how many elements of that bin are still to be taken.
__device__ void tk_find(const unsigned *__restrict__ hist, unsigned need, unsigned *s_scan, unsigned *s_out) {
    const int t = threadIdx.x;
    constexpr int PER = TK_BINS / TK_T;                 // 8 bins per thread, thread 0 owns the top ones
    unsigned loc[PER], sum = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        loc[i] = hist[TK_BINS - 1 - (t * PER + i)];
        sum += loc[i];
    }
    s_scan[t] = sum;
    __syncthreads();
    if (t == 0) {                                       // 256-entry serial scan: a few hundred cycles
        unsigned run = 0u;
        for (int i = 0; i < TK_T; ++i) {
            const unsigned c = s_scan[i];
            s_scan[i] = run;
            run += c;
        }
        s_out[0] = 0u;                                  // defaults when fewer than `need` elements exist
        s_out[1] = 0u;
    }
    __syncthreads();
    unsigned before = s_scan[t];
    if (before < need && need <= before + sum) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (before < need && need <= before + loc[i]) {
                s_out[0] = (unsigned)(TK_BINS - 1 - (t * PER + i));
                s_out[1] = need - before;
            }
            before += loc[i];
        }
    }
    __syncthreads();
}

// The threshold state after `passes` completed passes: prefix key bits and the remaining need.
__device__ void tk_state(const unsigned *__restrict__ ghist_b, int passes, unsigned k, unsigned *s_scan,
                         unsigned *s_out, unsigned &prefix, unsigned &need) {
    prefix = 0u;
    need = k;
    for (int p = 0; p < passes; ++p) {
        tk_find(ghist_b + p * TK_BINS, need, s_scan, s_out);
        const unsigned d = s_out[0];
        need = s_out[1];
        prefix |= p == 0 ? d << 21 : (p == 1 ? d << 10 : d);
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned tk_prefix_mask(int passes) {
    return passes == 0 ? 0u : (passes == 1 ? 0xFFE00000u : (passes == 2 ? 0xFFFFFC00u : 0xFFFFFFFFu));
}

// grid (segments, B).  ghist [B][3][TK_BINS], zeroed by the launcher.
__global__ __launch_bounds__(TK_T) void tk_hist_kernel(const float *__restrict__ scores, long stride_b,
                                                       long stride_a, int A, int k, int pass,
                                                       unsigned *__restrict__ ghist) {
    __shared__ unsigned s_hist[TK_BINS];
    __shared__ unsigned s_scan[TK_T];
    __shared__ unsigned s_out[2];
    const int t = threadIdx.x, b = blockIdx.y;
    unsigned *gh = ghist + (long)b * 3 * TK_BINS;
    unsigned prefix, need;
    tk_state(gh, pass, (unsigned)k, s_scan, s_out, prefix, need);
    for (int i = t; i < TK_BINS; i += TK_T) s_hist[i] = 0u;
    __syncthreads();
    const unsigned mask = tk_prefix_mask(pass);
    const float *sc = scores + (long)b * stride_b;
    const int i0 = blockIdx.x * TK_SEG, i1 = min(A, i0 + TK_SEG);
    for (int i = i0 + t; i < i1; i += TK_T) {
        const unsigned key = tk_key(sc[(long)i * stride_a]);
        if ((key & mask) == prefix) atomicAdd(&s_hist[tk_digit(key, pass)], 1u);
    }
    __syncthreads();
    for (int i = t; i < TK_BINS; i += TK_T) {
        const unsigned c = s_hist[i];
        if (c) atomicAdd(&gh[pass * TK_BINS + i], c);
    }
}

// grid (segments, B): segcnt [B][S][2] = (elements above T, elements equal to T) of the segment.
__global__ __launch_bounds__(TK_T) void tk_count_kernel(const float *__restrict__ scores, long stride_b,
                                                        long stride_a, int A, int k,
                                                        const unsigned *__restrict__ ghist,
                                                        unsigned *__restrict__ segcnt) {
    __shared__ unsigned s_scan[TK_T];
    __shared__ unsigned s_out[2];
    __shared__ unsigned s_cnt[2];
    const int t = threadIdx.x, b = blockIdx.y;
    unsigned T, ties;
    tk_state(ghist + (long)b * 3 * TK_BINS, 3, (unsigned)k, s_scan, s_out, T, ties);
    if (t < 2) s_cnt[t] = 0u;
    __syncthreads();
    const float *sc = scores + (long)b * stride_b;
    const int i0 = blockIdx.x * TK_SEG, i1 = min(A, i0 + TK_SEG);
    unsigned gt = 0u, eq = 0u;
    for (int i = i0 + t; i < i1; i += TK_T) {
        const unsigned key = tk_key(sc[(long)i * stride_a]);
        gt += key > T;
        eq += key == T;
    }
    for (int o = 32; o > 0; o >>= 1) {
        gt += __shfl_xor(gt, o);
        eq += __shfl_xor(eq, o);
    }
    if ((t & 63) == 0) { atomicAdd(&s_cnt[0], gt); atomicAdd(&s_cnt[1], eq); }
    __syncthreads();
    if (t < 2) segcnt[((long)b * gridDim.x + blockIdx.x) * 2 + t] = s_cnt[t];
}

// grid (segments, B): the k candidates of every image, cand_key / cand_idx [B][k].
__global__ __launch_bounds__(TK_T) void tk_scatter_kernel(const float *__restrict__ scores, long stride_b,
                                                          long stride_a, int A, int k,
                                                          const unsigned *__restrict__ ghist,
                                                          const unsigned *__restrict__ segcnt,
                                                          unsigned *__restrict__ cand_key,
                                                          int *__restrict__ cand_idx) {
    __shared__ unsigned s_scan[TK_T];
    __shared__ unsigned s_out[2];
    __shared__ unsigned s_wave[TK_T / SLN_WAVE];
    __shared__ unsigned s_gt, s_tie_base, s_base[3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.y;
    unsigned T, ties;
    tk_state(ghist + (long)b * 3 * TK_BINS, 3, (unsigned)k, s_scan, s_out, T, ties);
    if (t == 0) {          // offsets of this segment: elements above T / ties in the segments before it
        unsigned gt_before = 0u, eq_before = 0u, gt_total = 0u;
        for (unsigned s = 0; s < gridDim.x; ++s) {
            const unsigned g = segcnt[((long)b * gridDim.x + s) * 2], e = segcnt[((long)b * gridDim.x + s) * 2 + 1];
            if (s < blockIdx.x) { gt_before += g; eq_before += e; }
            gt_total += g;
        }
        s_base[0] = gt_before; s_base[1] = eq_before; s_base[2] = gt_total;
        s_gt = 0u;
        s_tie_base = 0u;
    }
    __syncthreads();
    const unsigned gt_before = s_base[0], eq_before = s_base[1], gt_total = s_base[2];
    const float *sc = scores + (long)b * stride_b;
    unsigned *ck = cand_key + (long)b * k;
    int *ci = cand_idx + (long)b * k;
    const int i0 = blockIdx.x * TK_SEG, i1 = min(A, i0 + TK_SEG);
    for (int c0 = i0; c0 < i1; c0 += TK_T) {
        const int i = c0 + t;
        const unsigned key = i < i1 ? tk_key(sc[(long)i * stride_a]) : 0u;
        const bool gt = i < i1 && key > T, eq = i < i1 && key == T;
        if (gt) {
            const unsigned pos = gt_before + atomicAdd(&s_gt, 1u);
            ck[pos] = key; ci[pos] = i;
        }
        // ordered rank of this tie: ballot prefix within the wave, wave totals via LDS
        const unsigned long long bal = __ballot(eq);
        const unsigned in_wave = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = 0u, total = 0u;
#pragma unroll
        for (int w = 0; w < TK_T / SLN_WAVE; ++w) {
            const unsigned c = s_wave[w];
            if (w < wave) before += c;
            total += c;
        }
        const unsigned rank = eq_before + s_tie_base + before + in_wave;   // among all ties of the image
        if (eq && rank < ties) {
            ck[gt_total + rank] = key;
            ci[gt_total + rank] = i;
        }
        __syncthreads();
        if (t == 0) s_tie_base += total;
        __syncthreads();
    }
}

// One block per image: bitonic sort of the k candidates by (key descending, index ascending).
template <int SLOTS>
__global__ __launch_bounds__(TK_SORT_T) void tk_sort_kernel(const unsigned *__restrict__ cand_key,
                                                            const int *__restrict__ cand_idx, int k,
                                                            int64_t *__restrict__ order) {
    __shared__ unsigned s_key[SLOTS];
    __shared__ int s_idx[SLOTS];
    const int t = threadIdx.x;
    const unsigned *ck = cand_key + (long)blockIdx.x * k;
    const int *ci = cand_idx + (long)blockIdx.x * k;
    for (int i = t; i < SLOTS; i += TK_SORT_T) {
        s_key[i] = i < k ? ck[i] : 0u;
        s_idx[i] = i < k ? ci[i] : 0x7FFFFFFF;
    }
    __syncthreads();
    for (unsigned size = 2; size <= SLOTS; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned p = t; p < SLOTS / 2; p += TK_SORT_T) {
                const unsigned lo = 2 * p - (p & (stride - 1));
                const unsigned hi = lo + stride;
                const bool up = (lo & size) == 0;            // "ascending" block: best element first
                const unsigned ka = s_key[lo], kb = s_key[hi];
                const int ia = s_idx[lo], ib = s_idx[hi];
                const bool a_first = ka > kb || (ka == kb && ia < ib);   // a precedes b in visiting order
                if (a_first != up) {
                    s_key[lo] = kb; s_key[hi] = ka;
                    s_idx[lo] = ib; s_idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < k; i += TK_SORT_T) order[(long)blockIdx.x * k + i] = s_idx[i];
}

static inline size_t tk_align(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" size_t sln_topk_workspace_bytes(int B, int A, int k) {
    if (B < 1 || A < 1 || k < 1) return 0;
    const size_t S = (size_t)sln_div_up(A, TK_SEG);
    return tk_align((size_t)B * 3 * TK_BINS * 4) + tk_align((size_t)B * S * 2 * 4) + 2 * tk_align((size_t)B * k * 4);
}

/* scores: B rows of A floats, element (b, a) at scores[b*stride_b + a*stride_a] (strides in elements:
 * the foreground column of the [B,A,2] RPN probabilities is read in place).  order [B,k] int64. */
extern "C" int sln_topk_order_f32(const float *scores, int B, int A, long stride_b, long stride_a, int k,
                                  int64_t *order, void *workspace, size_t workspace_bytes,
                                  sln_stream_t stream) {
    sln_enter();
    if (B < 0 || A < 0 || k < 0 || k > A) return SLN_ERR_INVALID_ARG;
    if (k > TK_MAX_K) return SLN_ERR_UNSUPPORTED;
    if (B == 0 || k == 0) return SLN_OK;
    if (!scores || !order) return SLN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < sln_topk_workspace_bytes(B, A, k)) return SLN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int S = sln_div_up(A, TK_SEG);
    unsigned char *ws = (unsigned char *)workspace;
    unsigned *ghist = (unsigned *)ws;
    ws += tk_align((size_t)B * 3 * TK_BINS * 4);
    unsigned *segcnt = (unsigned *)ws;
    ws += tk_align((size_t)B * S * 2 * 4);
    unsigned *cand_key = (unsigned *)ws;
    ws += tk_align((size_t)B * k * 4);
    int *cand_idx = (int *)ws;
    if (hipMemsetAsync(ghist, 0, (size_t)B * 3 * TK_BINS * 4, st) != hipSuccess) return SLN_ERR_LAUNCH;
    const dim3 grid(S, B);
    for (int pass = 0; pass < 3; ++pass)
        hipLaunchKernelGGL(tk_hist_kernel, grid, dim3(TK_T), 0, st, scores, stride_b, stride_a, A, k, pass, ghist);
    hipLaunchKernelGGL(tk_count_kernel, grid, dim3(TK_T), 0, st, scores, stride_b, stride_a, A, k, ghist, segcnt);
    hipLaunchKernelGGL(tk_scatter_kernel, grid, dim3(TK_T), 0, st, scores, stride_b, stride_a, A, k, ghist, segcnt,
                       cand_key, cand_idx);
    if (k <= 256)
        hipLaunchKernelGGL(tk_sort_kernel<256>, dim3(B), dim3(TK_SORT_T), 0, st, cand_key, cand_idx, k, order);
    else if (k <= 1024)
        hipLaunchKernelGGL(tk_sort_kernel<1024>, dim3(B), dim3(TK_SORT_T), 0, st, cand_key, cand_idx, k, order);
    else if (k <= 2048)
        hipLaunchKernelGGL(tk_sort_kernel<2048>, dim3(B), dim3(TK_SORT_T), 0, st, cand_key, cand_idx, k, order);
    else
        hipLaunchKernelGGL(tk_sort_kernel<8192>, dim3(B), dim3(TK_SORT_T), 0, st, cand_key, cand_idx, k, order);
    return sln_launch_status();
}
