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

// One axis of crop_and_resize.c's sample position (:56-63, :71-78): input coordinate
// of crop bin i, its two integer taps and the lerp fraction.  false = outside the map
// (the bin is extrapolated in forward, skipped in backward).
__device__ __forceinline__ bool pyr_axis(float a1, float a2, int n, int cn, int i, int &lo, int &hi,
                                         float &frac) {
    const float s = (cn > 1) ? (a2 - a1) * (float)(n - 1) / (float)(cn - 1) : 0.0f;
    const float in = (cn > 1) ? a1 * (float)(n - 1) + (float)i * s
                              : (float)(0.5 * (double)(a1 + a2) * (double)(n - 1));
    if (in < 0 || in > (float)(n - 1)) return false;
    const float f = floorf(in);
    lo = (int)f; hi = (int)ceilf(in); frac = in - f;
    return true;
}

__device__ __forceinline__ bool pyr_sample(const float *box, int H, int W, int ch, int cw, int y,
                                           int x, int &top, int &bot, int &lft, int &rgt, float &yl,
                                           float &xl) {
    const bool oky = pyr_axis(box[0], box[2], H, ch, y, top, bot, yl);
    const bool okx = pyr_axis(box[1], box[3], W, cw, x, lft, rgt, xl);
    return oky && okx;
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

// Backward for crops with more bins than the roi has pixels (the 16x16 mask-head
// crops): per-sample scatter would issue 4 atomics per (bin, channel), most of them
// landing on the same few map pixels.  Bilinear sampling is separable, so a block
// (one roi x 64 channels) inverts the tap lists per axis in LDS -- for every map row /
// column of the roi's footprint, which bins touch it and with what weight -- and then
// GATHERS each footprint pixel's sum from the crop gradient and issues ONE atomic per
// (pixel, channel); atomics are still needed because rois overlap.  The individual
// products wx*(wy*g) are the reference's (crop_and_resize.c:169-186); only the order
// of the additions differs, which the reference's own atomics leave unspecified.
#define PYR_MAXS 32   // max bins per axis on this path
#define PYR_MAXP 48   // max footprint rows / columns on this path
__global__ __launch_bounds__(256) void pyr_bwd_patch_kernel(PyrGrads gm, int B, int C,
                                                            const float *__restrict__ grads,
                                                            int g_cstride, int g_coffset,
                                                            const float *__restrict__ boxes,
                                                            const int32_t *__restrict__ box_ind,
                                                            const int32_t *__restrict__ level,
                                                            int K, int ch, int cw) {
    __shared__ int s_lo[2][PYR_MAXS], s_hi[2][PYR_MAXS], s_ok[2][PYR_MAXS];
    __shared__ float s_fr[2][PYR_MAXS];
    __shared__ int s_start[2][PYR_MAXP + 2], s_cur[2][PYR_MAXP + 1];
    __shared__ int s_ei[2][2 * PYR_MAXS];
    __shared__ float s_ew[2][2 * PYR_MAXS];
    __shared__ int s_ext[4];
    const int k = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int t = threadIdx.x, wv = t >> 6;
    const int bi = box_ind[k];
    const int lv = level[k] - 2;
    if (bi < 0 || bi >= B || lv < 0 || lv > 3) return;
    const int H = gm.H[lv], W = gm.W[lv];
    const float *box = boxes + (size_t)k * 4;
    if (t < 64) {
        if (t < ch) {
            int lo = 0, hi = 0; float fr = 0;
            const bool ok = pyr_axis(box[0], box[2], H, ch, t, lo, hi, fr);
            s_lo[0][t] = lo; s_hi[0][t] = hi; s_fr[0][t] = fr; s_ok[0][t] = ok;
        }
    } else if (t < 128) {
        const int i = t - 64;
        if (i < cw) {
            int lo = 0, hi = 0; float fr = 0;
            const bool ok = pyr_axis(box[1], box[3], W, cw, i, lo, hi, fr);
            s_lo[1][i] = lo; s_hi[1][i] = hi; s_fr[1][i] = fr; s_ok[1][i] = ok;
        }
    }
    __syncthreads();
    if (t < 2) {   // axis t: footprint extent, then a counting sort of the taps by map row/col
        const int n = t == 0 ? ch : cw;
        int mn = 0x7fffffff, mx = -1;
        for (int i = 0; i < n; ++i)
            if (s_ok[t][i]) { mn = min(mn, s_lo[t][i]); mx = max(mx, s_hi[t][i]); }
        const int ext = mx >= 0 ? mx - mn + 1 : 0;
        s_ext[2 * t] = mn; s_ext[2 * t + 1] = ext;
        if (ext > 0 && ext <= PYR_MAXP) {
            for (int p = 0; p <= ext; ++p) s_start[t][p] = 0;
            for (int i = 0; i < n; ++i)
                if (s_ok[t][i]) {
                    s_start[t][s_lo[t][i] - mn + 1]++;
                    if (s_hi[t][i] != s_lo[t][i]) s_start[t][s_hi[t][i] - mn + 1]++;
                }
            for (int p = 1; p <= ext; ++p) s_start[t][p] += s_start[t][p - 1];
            for (int p = 0; p < ext; ++p) s_cur[t][p] = s_start[t][p];
            for (int i = 0; i < n; ++i)
                if (s_ok[t][i]) {
                    int e = s_cur[t][s_lo[t][i] - mn]++;
                    s_ei[t][e] = i; s_ew[t][e] = 1 - s_fr[t][i];
                    if (s_hi[t][i] != s_lo[t][i]) {   // hi == lo: frac is 0, the second tap adds 0
                        e = s_cur[t][s_hi[t][i] - mn]++;
                        s_ei[t][e] = i; s_ew[t][e] = s_fr[t][i];
                    }
                }
        }
    }
    __syncthreads();
    const int p0y = s_ext[0], ph = s_ext[1], p0x = s_ext[2], pw = s_ext[3];
    if (ph <= 0 || pw <= 0 || c >= C) return;
    const float *g = grads + (size_t)k * ch * cw * g_cstride + g_coffset + c;
    float *img = gm.ptr[lv] + (size_t)bi * H * W * C + c;
    if (ph > PYR_MAXP || pw > PYR_MAXP || ph * pw > 2 * ch * cw) {
        // footprint larger than the crop: per-bin scatter is the cheaper side
        for (int s = wv; s < ch * cw; s += 4) {
            const int y = s / cw, x = s - y * cw;
            if (!s_ok[0][y] || !s_ok[1][x]) continue;
            const float yl = s_fr[0][y], xl = s_fr[1][x];
            const float gv = g[(size_t)s * g_cstride];
            const float dtop = (1 - yl) * gv, dbot = yl * gv;
            float *rt = img + (size_t)s_lo[0][y] * W * C, *rb = img + (size_t)s_hi[0][y] * W * C;
            const size_t l = (size_t)s_lo[1][x] * C, r = (size_t)s_hi[1][x] * C;
            atomicAdd(rt + l, (1 - xl) * dtop);
            atomicAdd(rt + r, xl * dtop);
            atomicAdd(rb + l, (1 - xl) * dbot);
            atomicAdd(rb + r, xl * dbot);
        }
        return;
    }
    for (int p = wv; p < ph * pw; p += 4) {
        const int pr = p / pw, pc = p - pr * pw;
        const int ys = s_start[0][pr], ye = s_start[0][pr + 1];
        const int xs = s_start[1][pc], xe = s_start[1][pc + 1];
        if (ys == ye || xs == xe) continue;
        float acc = 0.0f;
        for (int e = ys; e < ye; ++e) {
            const float wy = s_ew[0][e];
            const float *gy = g + (size_t)s_ei[0][e] * cw * g_cstride;
            for (int f = xs; f < xe; ++f) acc += s_ew[1][f] * (wy * gy[(size_t)s_ei[1][f] * g_cstride]);
        }
        atomicAdd(img + ((size_t)(p0y + pr) * W + (p0x + pc)) * C, acc);
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
                                        float *const *grad_maps, const int32_t *map_hw, int accumulate,
                                        sln_stream_t stream) {
    sln_enter();
    if (!grad_maps || !map_hw || B < 0 || C < 1 || K < 0 || ch < 1 || cw < 1) return SLN_ERR_INVALID_ARG;
    if (g_cstride < C + g_coffset || g_coffset < 0) return SLN_ERR_INVALID_ARG;
    PyrGrads gm;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 4; ++i) {
        gm.ptr[i] = grad_maps[i]; gm.H[i] = map_hw[2 * i]; gm.W[i] = map_hw[2 * i + 1];
        if (!gm.ptr[i] || gm.H[i] < 1 || gm.W[i] < 1) return SLN_ERR_INVALID_ARG;
        if (!accumulate && B > 0 && hipMemsetAsync(gm.ptr[i], 0, sizeof(float) * (size_t)B * gm.H[i] * gm.W[i] * C, st) !=
                         hipSuccess)
            return SLN_ERR_LAUNCH;
    }
    if (K == 0 || B == 0) return SLN_OK;
    if (!grads || !boxes || !box_ind || !level) return SLN_ERR_INVALID_ARG;
    const long nsamp = (long)K * ch * cw;
    if (ch * cw >= 64 && ch <= PYR_MAXS && cw <= PYR_MAXS)   // dense crops: footprint gather
        hipLaunchKernelGGL(pyr_bwd_patch_kernel, dim3(K, (C + 63) / 64), dim3(256), 0, st, gm, B, C,
                           grads, g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
    else
        hipLaunchKernelGGL(pyr_bwd_kernel, dim3(pyr_grid(nsamp)), dim3(256), 0, st, gm, B, C, grads,
                           g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
    return sln_launch_status();
}
