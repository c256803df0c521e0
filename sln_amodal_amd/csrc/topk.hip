// Batched top-k selection in visiting order for the proposal front end (gfx950).
//
// The reference sorts all A = 261 888 foreground scores of an image and keeps the first 6000
// (modal/Functions.py:133-147: scores.sort(descending=True), then [:pre_nms_limit]).  Here one
// 1024-thread block per image
//   1. finds the k-th largest key with a 4-pass, 8-bit MSB radix select (LDS histograms),
//   2. collects every element above the threshold plus the lowest-index ties of the threshold
//      (ordered chunk scan), k (key, index) pairs in LDS,
//   3. sorts them in LDS (bitonic, 8192 slots) by (score descending, index ascending)
// and writes the anchor indices: exactly torch.sort(..., descending=True, stable=True)[1][:k] (ties
// broken by the lower index; the reference's sort is unstable, so ties are unspecified there).
// Keys are the usual order-preserving integer image of the floats (NaN above +inf, like torch).
#include "common.h"

#define TK_THREADS 1024
#define TK_SLOTS 8192

__device__ __forceinline__ unsigned tk_key(float v) {
    unsigned u = __float_as_uint(v);
    if ((u << 1) == 0u) u = 0u;          // -0.0 compares equal to +0.0 in a sort: one key for both
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(TK_THREADS) void topk_order_kernel(const float *__restrict__ scores, long stride_b,
                                                                long stride_a, int A, int k,
                                                                int64_t *__restrict__ order) {
    __shared__ unsigned s_key[TK_SLOTS];
    __shared__ int s_idx[TK_SLOTS];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_prefix, s_need, s_count, s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *sc = scores + (long)blockIdx.x * stride_b;

    // ---- 1. radix select: the key T of the k-th largest element, and how many ties of T to take ----
    if (t == 0) { s_prefix = 0u; s_need = (unsigned)k; }
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
        if (t < 256) s_hist[t] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = t; i < A; i += TK_THREADS) {
            const unsigned key = tk_key(sc[(long)i * stride_a]);
            if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            unsigned need = s_need, d = 255u;
            for (;; --d) {                      // from the largest digit down
                const unsigned c = s_hist[d];
                if (c >= need || d == 0u) break;
                need -= c;
            }
            s_prefix = prefix | (d << (8 * pass));
            s_need = need;                      // elements still to take among those with this digit
        }
        mask |= 255u << (8 * pass);
        __syncthreads();
    }
    const unsigned T = s_prefix, ties = s_need;

    // ---- 2. collect: key > T anywhere, key == T in index order until `ties` are taken ----
    if (t == 0) { s_count = 0u; s_base = 0u; }
    for (int i = t; i < TK_SLOTS; i += TK_THREADS) { s_key[i] = 0u; s_idx[i] = 0x7FFFFFFF; }
    __syncthreads();
    for (int i0 = 0; i0 < A; i0 += TK_THREADS) {
        const int i = i0 + t;
        const unsigned key = i < A ? tk_key(sc[(long)i * stride_a]) : 0u;
        const bool gt = i < A && key > T, eq = i < A && key == T;
        if (gt) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            s_key[pos] = key; s_idx[pos] = i;
        }
        // ordered rank of this tie inside the chunk: ballot prefix within the wave, wave totals via LDS
        const unsigned long long bal = __ballot(eq);
        const unsigned in_wave = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = 0u, total = 0u;
        for (int w = 0; w < 16; ++w) {
            const unsigned c = s_wave[w];
            if (w < wave) before += c;
            total += c;
        }
        const unsigned rank = s_base + before + in_wave;      // 0-based rank among all ties so far
        if (eq && rank < ties) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            s_key[pos] = key; s_idx[pos] = i;
        }
        __syncthreads();
        if (t == 0) s_base += total;
        __syncthreads();
    }
    __syncthreads();

    // ---- 3. bitonic sort of the 8192 slots by (key descending, index ascending) ----
    for (unsigned size = 2; size <= TK_SLOTS; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned p = t; p < TK_SLOTS / 2; p += TK_THREADS) {
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
    for (int i = t; i < k; i += TK_THREADS) order[(long)blockIdx.x * k + i] = s_idx[i];
}

/* scores: B rows of A floats, element (b, a) at scores[b*stride_b + a*stride_a] (strides in elements:
 * the foreground column of the [B,A,2] RPN probabilities is read in place).  order [B,k] int64. */
extern "C" int sln_topk_order_f32(const float *scores, int B, int A, long stride_b, long stride_a, int k,
                                  int64_t *order, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || A < 0 || k < 0 || k > A) return SLN_ERR_INVALID_ARG;
    if (k > TK_SLOTS) return SLN_ERR_UNSUPPORTED;
    if (B == 0 || k == 0) return SLN_OK;
    if (!scores || !order) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(topk_order_kernel, dim3(B), dim3(TK_THREADS), 0, (hipStream_t)stream, scores, stride_b,
                       stride_a, A, k, order);
    return sln_launch_status();
}
