// pyramid_roi_align as ONE launch on gfx950 (NHWC feature maps).
//
// The reference loops over FPN levels in Python, with .any()/nonzero() host
// syncs, per-level crop_and_resize launches, a cat and an index_select to restore
// roi order (modal/modals.py:66-108).  Here every roi carries its level; a wave
// picks the level's map and samples it with crop_and_resize.c's arithmetic
// (:44-106), writing straight into the roi's slot -- optionally at a channel
// offset of a wider output row (fused torch.cat for the mask head,
// modal/modals.py:481).  Compiled with -ffp-contract=off.
#include "common.h"

struct PyrMaps {
    const float *ptr[4];
    int H[4], W[4];
};
struct PyrGrads {
    float *ptr[4];
    int H[4], W[4];
};

__device__ __forceinline__ bool pyr_sample(const float *box, int H, int W, int ch, int cw, int y,
                                           int x, int &top, int &bot, int &lft, int &rgt, float &yl,
                                           float &xl) {
    const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
    const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
    const float ws = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
    const float in_y = (ch > 1) ? y1 * (float)(H - 1) + (float)y * hs
                                : (float)(0.5 * (double)(y1 + y2) * (double)(H - 1));
    const float in_x = (cw > 1) ? x1 * (float)(W - 1) + (float)x * ws
                                : (float)(0.5 * (double)(x1 + x2) * (double)(W - 1));
    if ((in_y < 0 || in_y > (float)(H - 1)) || (in_x < 0 || in_x > (float)(W - 1))) return false;
    const float fy = floorf(in_y), fx = floorf(in_x);
    top = (int)fy; bot = (int)ceilf(in_y); lft = (int)fx; rgt = (int)ceilf(in_x);
    yl = in_y - fy; xl = in_x - fx;
    return true;
}

template <int VEC>
__global__ __launch_bounds__(256) void pyr_fwd_kernel(PyrMaps maps, int B, int C,
                                                      const float *__restrict__ boxes,
                                                      const int32_t *__restrict__ box_ind,
                                                      const int32_t *__restrict__ level, int K,
                                                      int ch, int cw, float extrap,
                                                      float *__restrict__ out, int out_cstride,
                                                      int out_coffset) {
    const int lane = threadIdx.x & 63;
    const long nsamp = (long)K * ch * cw;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long sidx = wave0; sidx < nsamp; sidx += nwaves) {
        const int x = (int)(sidx % cw);
        const int y = (int)((sidx / cw) % ch);
        const int k = (int)(sidx / ((long)cw * ch));
        const int bi = box_ind[k];
        const int lv = level[k] - 2;
        float *o = out + sidx * out_cstride + out_coffset;
        if (bi < 0 || bi >= B || lv < 0 || lv > 3) {  // padded roi slot
            for (int c = lane; c < C; c += 64) o[c] = 0.0f;
            continue;
        }
        const int H = maps.H[lv], W = maps.W[lv];
        int top, bot, lft, rgt;
        float yl, xl;
        if (!pyr_sample(boxes + (size_t)k * 4, H, W, ch, cw, y, x, top, bot, lft, rgt, yl, xl)) {
            for (int c = lane; c < C; c += 64) o[c] = extrap;
            continue;
        }
        const float *img = maps.ptr[lv] + (size_t)bi * H * W * C;
        const float *ptl = img + ((size_t)top * W + lft) * C, *ptr = img + ((size_t)top * W + rgt) * C;
        const float *pbl = img + ((size_t)bot * W + lft) * C, *pbr = img + ((size_t)bot * W + rgt) * C;
        if (VEC == 4) {
            for (int c = lane * 4; c < C; c += 256) {
                const float4 tl = *(const float4 *)(ptl + c), tr = *(const float4 *)(ptr + c);
                const float4 bl = *(const float4 *)(pbl + c), br = *(const float4 *)(pbr + c);
                float4 r;
                float t, b2;
                t = tl.x + (tr.x - tl.x) * xl; b2 = bl.x + (br.x - bl.x) * xl; r.x = t + (b2 - t) * yl;
                t = tl.y + (tr.y - tl.y) * xl; b2 = bl.y + (br.y - bl.y) * xl; r.y = t + (b2 - t) * yl;
                t = tl.z + (tr.z - tl.z) * xl; b2 = bl.z + (br.z - bl.z) * xl; r.z = t + (b2 - t) * yl;
                t = tl.w + (tr.w - tl.w) * xl; b2 = bl.w + (br.w - bl.w) * xl; r.w = t + (b2 - t) * yl;
                *(float4 *)(o + c) = r;
            }
        } else {
            for (int c = lane; c < C; c += 64) {
                const float t = ptl[c] + (ptr[c] - ptl[c]) * xl;
                const float b2 = pbl[c] + (pbr[c] - pbl[c]) * xl;
                o[c] = t + (b2 - t) * yl;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pyr_bwd_kernel(PyrGrads gm, int B, int C,
                                                      const float *__restrict__ grads,
                                                      int g_cstride, int g_coffset,
                                                      const float *__restrict__ boxes,
                                                      const int32_t *__restrict__ box_ind,
                                                      const int32_t *__restrict__ level, int K,
                                                      int ch, int cw) {
    const int lane = threadIdx.x & 63;
    const long nsamp = (long)K * ch * cw;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long sidx = wave0; sidx < nsamp; sidx += nwaves) {
        const int x = (int)(sidx % cw);
        const int y = (int)((sidx / cw) % ch);
        const int k = (int)(sidx / ((long)cw * ch));
        const int bi = box_ind[k];
        const int lv = level[k] - 2;
        if (bi < 0 || bi >= B || lv < 0 || lv > 3) continue;
        const int H = gm.H[lv], W = gm.W[lv];
        int top, bot, lft, rgt;
        float yl, xl;
        if (!pyr_sample(boxes + (size_t)k * 4, H, W, ch, cw, y, x, top, bot, lft, rgt, yl, xl)) continue;
        const float *g = grads + sidx * g_cstride + g_coffset;
        float *img = gm.ptr[lv] + (size_t)bi * H * W * C;
        float *ptl = img + ((size_t)top * W + lft) * C, *ptr = img + ((size_t)top * W + rgt) * C;
        float *pbl = img + ((size_t)bot * W + lft) * C, *pbr = img + ((size_t)bot * W + rgt) * C;
        for (int c = lane; c < C; c += 64) {
            const float gv = g[c];
            const float dtop = (1 - yl) * gv, dbot = yl * gv;
            atomicAdd(ptl + c, (1 - xl) * dtop);
            atomicAdd(ptr + c, xl * dtop);
            atomicAdd(pbl + c, (1 - xl) * dbot);
            atomicAdd(pbr + c, xl * dbot);
        }
    }
}

static inline int pyr_grid(long nsamp) {
    long blocks = (nsamp + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int sln_pyramid_crop_fwd_f32(const float *const *maps, const int32_t *map_hw, int B, int C,
                                        const float *boxes, const int32_t *box_ind,
                                        const int32_t *level, int K, int ch, int cw,
                                        float extrapolation_value, float *out, int out_cstride,
                                        int out_coffset, sln_stream_t stream) {
    sln_enter();
    if (!maps || !map_hw || B < 0 || C < 1 || K < 0 || ch < 1 || cw < 1) return SLN_ERR_INVALID_ARG;
    if (out_cstride < C + out_coffset || out_coffset < 0) return SLN_ERR_INVALID_ARG;
    if (K == 0) return SLN_OK;
    if (!boxes || !box_ind || !level || !out) return SLN_ERR_INVALID_ARG;
    PyrMaps pm;
    for (int i = 0; i < 4; ++i) {
        pm.ptr[i] = maps[i]; pm.H[i] = map_hw[2 * i]; pm.W[i] = map_hw[2 * i + 1];
        if (!pm.ptr[i] || pm.H[i] < 1 || pm.W[i] < 1) return SLN_ERR_INVALID_ARG;
    }
    const long nsamp = (long)K * ch * cw;
    const bool vec = (C % 4 == 0) && (out_cstride % 4 == 0) && (out_coffset % 4 == 0) &&
                     ((((size_t)out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(pyr_fwd_kernel<4>, dim3(pyr_grid(nsamp)), dim3(256), 0, (hipStream_t)stream,
                           pm, B, C, boxes, box_ind, level, K, ch, cw, extrapolation_value, out,
                           out_cstride, out_coffset);
    else
        hipLaunchKernelGGL(pyr_fwd_kernel<1>, dim3(pyr_grid(nsamp)), dim3(256), 0, (hipStream_t)stream,
                           pm, B, C, boxes, box_ind, level, K, ch, cw, extrapolation_value, out,
                           out_cstride, out_coffset);
    return sln_launch_status();
}

extern "C" int sln_pyramid_crop_bwd_f32(const float *grads, int g_cstride, int g_coffset,
                                        const float *boxes, const int32_t *box_ind,
                                        const int32_t *level, int K, int ch, int cw, int B, int C,
                                        float *const *grad_maps, const int32_t *map_hw,
                                        sln_stream_t stream) {
    sln_enter();
    if (!grad_maps || !map_hw || B < 0 || C < 1 || K < 0 || ch < 1 || cw < 1) return SLN_ERR_INVALID_ARG;
    if (g_cstride < C + g_coffset || g_coffset < 0) return SLN_ERR_INVALID_ARG;
    PyrGrads gm;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 4; ++i) {
        gm.ptr[i] = grad_maps[i]; gm.H[i] = map_hw[2 * i]; gm.W[i] = map_hw[2 * i + 1];
        if (!gm.ptr[i] || gm.H[i] < 1 || gm.W[i] < 1) return SLN_ERR_INVALID_ARG;
        if (B > 0 && hipMemsetAsync(gm.ptr[i], 0, sizeof(float) * (size_t)B * gm.H[i] * gm.W[i] * C, st) !=
                         hipSuccess)
            return SLN_ERR_LAUNCH;
    }
    if (K == 0 || B == 0) return SLN_OK;
    if (!grads || !boxes || !box_ind || !level) return SLN_ERR_INVALID_ARG;
    const long nsamp = (long)K * ch * cw;
    hipLaunchKernelGGL(pyr_bwd_kernel, dim3(pyr_grid(nsamp)), dim3(256), 0, st, gm, B, C, grads,
                       g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
    return sln_launch_status();
}
