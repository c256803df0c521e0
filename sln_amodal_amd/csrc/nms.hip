// Greedy NMS for gfx950: IoU bit-matrix + device-side wave64 reduce.
//
// Semantics: reference CPU path, nms/src/nms.c:33-63 with areas from
// nms/pth_nms.py:16 -- suppress on ovr >= thresh, +1 widths, fp32 IEEE.
// This file is compiled with -ffp-contract=off (see build.py) so that every
// float expression rounds once per operation exactly like the CPU oracle.
//
// Kernel 1 (nms_mask_kernel): one wave per (64-row block, 64-col block) tile of
//   the upper triangle; lane = row box, the 64 column boxes are staged in LDS and
//   read back as broadcasts; emits one u64 suppression word per row.  The wave
//   width (64) is the word width, so a lane owns exactly one word.
// Kernel 2 (nms_reduce_kernel): one wave per image walks the row blocks in score
//   order.  The 64x64 diagonal tile is resolved with scalar readlane ops, the
//   kept rows' words are then OR-ed into a lane-distributed "removed" bitmap with
//   independent (pipelined) loads.  No host round trip: the reference's gpu_nms
//   copies the N x N/64 mask to the host and reduces there (nms_cuda.c:31-58).
#include "common.h"

typedef unsigned long long u64;

#define NMS_MAXW 4  // removed-bitmap words per lane -> N <= 64*64*4 = 16384

__device__ __forceinline__ float box_area(float y1, float x1, float y2, float x2) {
    // pth_nms.py:16: (x2 - x1 + 1) * (y2 - y1 + 1)
    const float w = (x2 - x1) + 1.0f;
    const float h = (y2 - y1) + 1.0f;
    return w * h;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const float *__restrict__ dets, int N,
                                                       int nblk, const int32_t *__restrict__ n_valid,
                                                       float thresh, u64 *__restrict__ mask) {
    __shared__ float s_box[4][SLN_WAVE][5];
    const int b = blockIdx.z;
    const int rb = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cb = blockIdx.x * 4 + wave;
    if (cb < rb || cb >= nblk) return;  // wave-uniform; no barrier below
    const int nv = n_valid ? min(n_valid[b], N) : N;
    const float *d = dets + (size_t)b * N * 5;

    const int col = cb * 64 + lane;
    if (col < nv) {
        const float y1 = d[col * 5 + 0], x1 = d[col * 5 + 1];
        const float y2 = d[col * 5 + 2], x2 = d[col * 5 + 3];
        s_box[wave][lane][0] = y1; s_box[wave][lane][1] = x1;
        s_box[wave][lane][2] = y2; s_box[wave][lane][3] = x2;
        s_box[wave][lane][4] = box_area(y1, x1, y2, x2);
    }
    // same wave wrote and reads s_box[wave]: LDS ops of one wave complete in order
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    const int row = rb * 64 + lane;
    u64 word = 0;
    if (row < nv) {
        const float iy1 = d[row * 5 + 0], ix1 = d[row * 5 + 1];
        const float iy2 = d[row * 5 + 2], ix2 = d[row * 5 + 3];
        const float iarea = box_area(iy1, ix1, iy2, ix2);
        const int ncol = min(64, nv - cb * 64);
        for (int j = 0; j < ncol; ++j) {
            const float yy1 = fmaxf(iy1, s_box[wave][j][0]);
            const float xx1 = fmaxf(ix1, s_box[wave][j][1]);
            const float yy2 = fminf(iy2, s_box[wave][j][2]);
            const float xx2 = fminf(ix2, s_box[wave][j][3]);
            const float w = fmaxf(0.0f, (xx2 - xx1) + 1.0f);
            const float h = fmaxf(0.0f, (yy2 - yy1) + 1.0f);
            const float inter = w * h;
            const float ovr = inter / ((iarea + s_box[wave][j][4]) - inter);
            const bool later = (cb * 64 + j) > row;
            if (later && ovr >= thresh) word |= 1ull << j;
        }
    }
    if (row < N) mask[((size_t)b * N + row) * nblk + cb] = word;
}

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(const u64 *__restrict__ mask, int N, int nblk,
                                                        const int32_t *__restrict__ n_valid,
                                                        int max_out, int64_t *__restrict__ keep,
                                                        int32_t *__restrict__ num_keep) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nv = n_valid ? min(n_valid[b], N) : N;
    const u64 *m = mask + (size_t)b * N * nblk;
    int64_t *kp = keep + (size_t)b * max_out;
    u64 remv[NMS_MAXW];
#pragma unroll
    for (int s = 0; s < NMS_MAXW; ++s) remv[s] = 0;
    int nk = 0;
    const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int rb = 0; rb < nblk && rb * 64 < nv && nk < max_out; ++rb) {
        const int row = rb * 64 + lane;
        const u64 diag = (row < nv) ? m[(size_t)row * nblk + rb] : 0ull;
        // word `rb` of the removed bitmap lives in lane rb%64, slot rb/64
        u64 held = remv[0];
#pragma unroll
        for (int s = 1; s < NMS_MAXW; ++s)
            if ((rb >> 6) == s) held = remv[s];
        u64 cur = readlane64(held, rb & 63);
        const int nrow = min(64, nv - rb * 64);
        if (nrow < 64) cur |= ~0ull << nrow;  // rows past the end count as removed
        u64 alive = 0;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const u64 dt = readlane64(diag, t);
            if (!((cur >> t) & 1ull)) {
                alive |= 1ull << t;
                cur |= dt;
            }
        }
        if ((alive >> lane) & 1ull) {
            const int pos = nk + __popcll(alive & lane_lt);
            if (pos < max_out) kp[pos] = row;
        }
        nk += __popcll(alive);
        // OR the kept rows' words (> rb) into the lane-distributed bitmap,
        // four independent row loads in flight per step.
        u64 bits = alive;
        while (bits) {
            int r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (bits) { r[q] = rb * 64 + __builtin_ctzll(bits); bits &= bits - 1; }
                else r[q] = -1;
            }
#pragma unroll
            for (int s = 0; s < NMS_MAXW; ++s) {
                const int w = lane + 64 * s;
                if (w > rb && w < nblk) {
                    u64 acc = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (r[q] >= 0) acc |= m[(size_t)r[q] * nblk + w];
                    remv[s] |= acc;
                }
            }
        }
    }
    const int kept = min(nk, max_out);
    for (int p = kept + lane; p < max_out; p += 64) kp[p] = -1;
    if (lane == 0) num_keep[b] = kept;
}

extern "C" size_t sln_nms_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    const size_t nblk = (size_t)sln_div_up(N, 64);
    return (size_t)B * (size_t)N * nblk * sizeof(u64);
}

extern "C" int sln_nms_f32(const float *dets, int B, int N, const int32_t *n_valid, float thresh,
                           int max_out, int64_t *keep, int32_t *num_keep, void *workspace,
                           size_t workspace_bytes, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || N < 0 || max_out < 0) return SLN_ERR_INVALID_ARG;
    if (B == 0) return SLN_OK;
    if (!keep && max_out > 0) return SLN_ERR_INVALID_ARG;
    if (!num_keep) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = sln_div_up(N, 64);
    if (nblk > 64 * NMS_MAXW) return SLN_ERR_UNSUPPORTED;
    if (N > 0) {
        if (!dets) return SLN_ERR_INVALID_ARG;
        if (!workspace || workspace_bytes < sln_nms_workspace_bytes(B, N)) return SLN_ERR_WORKSPACE;
        dim3 grid(sln_div_up(nblk, 4), nblk, B);
        hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, st, dets, N, nblk, n_valid, thresh,
                           (u64 *)workspace);
    }
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, (const u64 *)workspace, N, nblk,
                       n_valid, max_out, keep, num_keep);
    return sln_launch_status();
}
