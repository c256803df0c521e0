// Proposal front end on gfx950: gather top-n anchors, decode deltas, clip, emit
// NMS-ready rows; and the post-NMS gather + normalise.
//
// Arithmetic: modal/Functions.py:77-98 (apply_box_deltas), :101-111
// (clip_boxes), :131-135 (deltas * RPN_BBOX_STD_DEV), :172-176 (normalise).
// Compiled with -ffp-contract=off (each torch op in the reference rounds once).
#include "common.h"

__global__ __launch_bounds__(256) void proposal_decode_kernel(
    const float *__restrict__ probs, const float *__restrict__ deltas,
    const float *__restrict__ anchors, const int64_t *__restrict__ order, int A, int n, float s0,
    float s1, float s2, float s3, float win_h, float win_w, float *__restrict__ dets) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long a = order[(size_t)b * n + i];
    float *o = dets + ((size_t)b * n + i) * 5;
    if (a < 0 || a >= A) {  // defensive: emit an empty, lowest-score row
        o[0] = o[1] = o[2] = o[3] = 0.0f; o[4] = -1.0f;
        return;
    }
    const float4 an = *(const float4 *)(anchors + a * 4);
    const float4 dl = *(const float4 *)(deltas + ((size_t)b * A + a) * 4);
    const float score = probs[((size_t)b * A + a) * 2 + 1];
    float height = an.z - an.x, width = an.w - an.y;
    float cy = an.x + 0.5f * height, cx = an.y + 0.5f * width;
    const float dy = dl.x * s0, dx = dl.y * s1, dh = dl.z * s2, dw = dl.w * s3;
    cy = cy + dy * height;
    cx = cx + dx * width;
    height = height * expf(dh);
    width = width * expf(dw);
    float y1 = cy - 0.5f * height, x1 = cx - 0.5f * width;
    float y2 = y1 + height, x2 = x1 + width;
    y1 = fminf(fmaxf(y1, 0.0f), win_h); x1 = fminf(fmaxf(x1, 0.0f), win_w);
    y2 = fminf(fmaxf(y2, 0.0f), win_h); x2 = fminf(fmaxf(x2, 0.0f), win_w);
    o[0] = y1; o[1] = x1; o[2] = y2; o[3] = x2; o[4] = score;
}

__global__ __launch_bounds__(256) void gather_rois_kernel(const float *__restrict__ dets,
                                                          const int64_t *__restrict__ keep,
                                                          const int32_t *__restrict__ num_keep, int N,
                                                          int max_out, float nh, float nw,
                                                          float *__restrict__ rois) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= max_out) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < num_keep[b]) {
        const long k = keep[(size_t)b * max_out + r];
        if (k >= 0 && k < N) {
            const float *d = dets + ((size_t)b * N + k) * 5;
            v = make_float4(d[0] / nh, d[1] / nw, d[2] / nh, d[3] / nw);
        }
    }
    *(float4 *)(rois + ((size_t)b * max_out + r) * 4) = v;
}

extern "C" int sln_proposal_decode_f32(const float *probs, const float *deltas, const float *anchors,
                                       const int64_t *order, int B, int A, int n,
                                       const float *std_dev, float win_h, float win_w, float *dets,
                                       sln_stream_t stream) {
    sln_enter();
    if (B < 0 || A < 0 || n < 0 || !std_dev) return SLN_ERR_INVALID_ARG;
    if (B == 0 || n == 0) return SLN_OK;
    if (!probs || !deltas || !anchors || !order || !dets) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(proposal_decode_kernel, dim3(sln_div_up(n, 256), B), dim3(256), 0,
                       (hipStream_t)stream, probs, deltas, anchors, order, A, n, std_dev[0],
                       std_dev[1], std_dev[2], std_dev[3], win_h, win_w, dets);
    return sln_launch_status();
}

extern "C" int sln_gather_rois_f32(const float *dets, const int64_t *keep, const int32_t *num_keep,
                                   int B, int N, int max_out, float norm_h, float norm_w, float *rois,
                                   sln_stream_t stream) {
    sln_enter();
    if (B < 0 || N < 0 || max_out < 0) return SLN_ERR_INVALID_ARG;
    if (B == 0 || max_out == 0) return SLN_OK;
    if (!keep || !num_keep || !rois || (!dets && N > 0)) return SLN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gather_rois_kernel, dim3(sln_div_up(max_out, 256), B), dim3(256), 0,
                       (hipStream_t)stream, dets, keep, num_keep, N, max_out, norm_h, norm_w, rois);
    return sln_launch_status();
}
