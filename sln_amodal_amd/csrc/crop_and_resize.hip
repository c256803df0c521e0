// crop_and_resize (TF-style RoIAlign: one bilinear sample per output bin) for
// gfx950, forward + backward, NCHW (reference layout) and NHWC (channels-last,
// what the HIP conv stack produces and the layout this op is HBM-efficient in).
//
// Arithmetic: roialign/roi_align/src/crop_and_resize.c:44-106 (fwd), :196-247
// (bwd).  Compiled with -ffp-contract=off: the sample coordinate
// in = lo*(size-1) + idx*scale decides the floor/ceil taps and must round like
// the CPU path (no FMA); the crop==1 branch is evaluated in double
// (crop_and_resize.c:56).
//
// HBM model: forward 20 B per output element (4 taps + 1 store),
// backward 36 B (1 load + 4 atomic RMW).  In NHWC a sample's 4 taps are 4
// contiguous C-float runs and the output sample is one contiguous run, so every
// wave instruction moves whole 256 B-1 KiB segments; in NCHW each tap of each
// channel lives on its own cache line (up to 32x read amplification), which is
// why the pipeline keeps feature maps channels-last and the NCHW kernels exist
// for drop-in compatibility.
#include "common.h"

struct Sample {
    int top, bot, lft, rgt;
    float yl, xl;
    bool valid;
};

__device__ __forceinline__ float in_coord(float lo, float hi, int size, int crop, int idx,
                                          float scale) {
    if (crop > 1) return lo * (float)(size - 1) + (float)idx * scale;
    return (float)(0.5 * (double)(lo + hi) * (double)(size - 1));
}

__device__ __forceinline__ Sample make_sample(const float *__restrict__ box, int H, int W, int ch,
                                              int cw, int y, int x) {
    const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
    const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
    const float ws = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
    const float in_y = in_coord(y1, y2, H, ch, y, hs);
    const float in_x = in_coord(x1, x2, W, cw, x, ws);
    Sample s;
    // written as the reference writes it (a NaN coordinate is NOT extrapolated there)
    s.valid = !(in_y < 0 || in_y > (float)(H - 1)) && !(in_x < 0 || in_x > (float)(W - 1));
    const float fy = floorf(in_y), fx = floorf(in_x);
    s.top = (int)fy; s.bot = (int)ceilf(in_y);
    s.lft = (int)fx; s.rgt = (int)ceilf(in_x);
    s.yl = in_y - fy;
    s.xl = in_x - fx;
    return s;
}

__device__ __forceinline__ float lerp2(float tl, float tr, float bl, float br, float xl, float yl) {
    const float t = tl + (tr - tl) * xl;
    const float b = bl + (br - bl) * xl;
    return t + (b - t) * yl;
}

// ------------------------------------------------------------------ NHWC forward
// One wave per output sample (k, y, x); lanes sweep the channel run.  VEC = 4
// uses 16-B loads/stores (C % 4 == 0), VEC = 1 is the ragged-C path (C = 183).
template <int VEC>
__global__ __launch_bounds__(256) void car_fwd_nhwc(const float *__restrict__ image, int B, int C,
                                                    int H, int W, const float *__restrict__ boxes,
                                                    const int32_t *__restrict__ box_ind, int K,
                                                    int ch, int cw, float extrap,
                                                    float *__restrict__ crops, int32_t *err) {
    const int lane = threadIdx.x & 63;
    const long nsamp = (long)K * ch * cw;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long sidx = wave0; sidx < nsamp; sidx += nwaves) {
        const int x = (int)(sidx % cw);
        const int y = (int)((sidx / cw) % ch);
        const int k = (int)(sidx / ((long)cw * ch));
        const int bi = box_ind[k];
        float *out = crops + sidx * C;
        if (bi < 0 || bi >= B) {
            if (lane == 0 && err) atomicOr(err, 1);
            for (int c = lane; c < C; c += 64) out[c] = 0.0f;
            continue;
        }
        const Sample s = make_sample(boxes + (size_t)k * 4, H, W, ch, cw, y, x);
        if (!s.valid) {
            for (int c = lane; c < C; c += 64) out[c] = extrap;
            continue;
        }
        const float *img = image + (size_t)bi * H * W * C;
        const float *ptl = img + ((size_t)s.top * W + s.lft) * C;
        const float *ptr = img + ((size_t)s.top * W + s.rgt) * C;
        const float *pbl = img + ((size_t)s.bot * W + s.lft) * C;
        const float *pbr = img + ((size_t)s.bot * W + s.rgt) * C;
        if (VEC == 4) {
            for (int c = lane * 4; c < C; c += 256) {
                const float4 tl = *(const float4 *)(ptl + c), tr = *(const float4 *)(ptr + c);
                const float4 bl = *(const float4 *)(pbl + c), br = *(const float4 *)(pbr + c);
                float4 o;
                o.x = lerp2(tl.x, tr.x, bl.x, br.x, s.xl, s.yl);
                o.y = lerp2(tl.y, tr.y, bl.y, br.y, s.xl, s.yl);
                o.z = lerp2(tl.z, tr.z, bl.z, br.z, s.xl, s.yl);
                o.w = lerp2(tl.w, tr.w, bl.w, br.w, s.xl, s.yl);
                *(float4 *)(out + c) = o;
            }
        } else {
            for (int c = lane; c < C; c += 64)
                out[c] = lerp2(ptl[c], ptr[c], pbl[c], pbr[c], s.xl, s.yl);
        }
    }
}

// ----------------------------------------------------------------- NHWC backward
// Same wave-per-sample mapping; lane l owns channels l, l+64, ... so that every
// atomic wave-instruction adds 256 contiguous bytes (the full-rate shape for
// global_atomic_add_f32 on gfx950).
__global__ __launch_bounds__(256) void car_bwd_nhwc(const float *__restrict__ grads,
                                                    const float *__restrict__ boxes,
                                                    const int32_t *__restrict__ box_ind, int K,
                                                    int ch, int cw, int B, int C, int H, int W,
                                                    float *__restrict__ gimg, int32_t *err) {
    const int lane = threadIdx.x & 63;
    const long nsamp = (long)K * ch * cw;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    for (long sidx = wave0; sidx < nsamp; sidx += nwaves) {
        const int x = (int)(sidx % cw);
        const int y = (int)((sidx / cw) % ch);
        const int k = (int)(sidx / ((long)cw * ch));
        const int bi = box_ind[k];
        if (bi < 0 || bi >= B) {
            if (lane == 0 && err) atomicOr(err, 1);
            continue;
        }
        const Sample s = make_sample(boxes + (size_t)k * 4, H, W, ch, cw, y, x);
        if (!s.valid) continue;
        const float *g = grads + sidx * C;
        float *img = gimg + (size_t)bi * H * W * C;
        float *ptl = img + ((size_t)s.top * W + s.lft) * C;
        float *ptr = img + ((size_t)s.top * W + s.rgt) * C;
        float *pbl = img + ((size_t)s.bot * W + s.lft) * C;
        float *pbr = img + ((size_t)s.bot * W + s.rgt) * C;
        for (int c = lane; c < C; c += 64) {
            const float gv = g[c];
            const float dtop = (1 - s.yl) * gv;
            const float dbot = s.yl * gv;
            atomicAdd(ptl + c, (1 - s.xl) * dtop);
            atomicAdd(ptr + c, s.xl * dtop);
            atomicAdd(pbl + c, (1 - s.xl) * dbot);
            atomicAdd(pbr + c, s.xl * dbot);
        }
    }
}

// ------------------------------------------------------------------ NCHW kernels
// Reference layout.  One block per (box, 32-channel slab): the ch x cw sample
// table is computed once into LDS, then threads sweep (d, y, x) with x fastest so
// that stores are contiguous ([C,ch,cw] of one box is one contiguous region).
#define CAR_MAX_TAB 1024  // ch*cw entries cached in LDS; larger crops recompute

__global__ __launch_bounds__(256) void car_fwd_nchw(const float *__restrict__ image, int B, int C,
                                                    int H, int W, const float *__restrict__ boxes,
                                                    const int32_t *__restrict__ box_ind, int K,
                                                    int ch, int cw, float extrap,
                                                    float *__restrict__ crops, int32_t *err) {
    __shared__ int s_off[CAR_MAX_TAB];     // top*W+lft, or -1 when extrapolated
    __shared__ int s_dx_dy[CAR_MAX_TAB];   // (rgt-lft) | (bot-top) << 1
    __shared__ float s_xl[CAR_MAX_TAB], s_yl[CAR_MAX_TAB];
    const int k = blockIdx.x;
    const int c0 = blockIdx.y * 32;
    const int nc = min(32, C - c0);
    const int bi = box_ind[k];
    const int ns = ch * cw;
    float *out = crops + ((size_t)k * C + c0) * ns;
    if (bi < 0 || bi >= B) {
        if (threadIdx.x == 0 && err) atomicOr(err, 1);
        for (int i = threadIdx.x; i < nc * ns; i += 256) out[i] = 0.0f;
        return;
    }
    const bool tab = ns <= CAR_MAX_TAB;
    if (tab) {
        for (int i = threadIdx.x; i < ns; i += 256) {
            const Sample s = make_sample(boxes + (size_t)k * 4, H, W, ch, cw, i / cw, i % cw);
            s_off[i] = s.valid ? s.top * W + s.lft : -1;
            s_dx_dy[i] = (s.rgt - s.lft) | ((s.bot - s.top) << 1);
            s_xl[i] = s.xl;
            s_yl[i] = s.yl;
        }
        __syncthreads();
    }
    const float *img = image + ((size_t)bi * C + c0) * H * W;
    for (int i = threadIdx.x; i < nc * ns; i += 256) {
        const int d = i / ns, si = i - d * ns;
        int off, dxdy;
        float xl, yl;
        if (tab) {
            off = s_off[si]; dxdy = s_dx_dy[si]; xl = s_xl[si]; yl = s_yl[si];
        } else {
            const Sample s = make_sample(boxes + (size_t)k * 4, H, W, ch, cw, si / cw, si % cw);
            off = s.valid ? s.top * W + s.lft : -1;
            dxdy = (s.rgt - s.lft) | ((s.bot - s.top) << 1);
            xl = s.xl; yl = s.yl;
        }
        if (off < 0) { out[i] = extrap; continue; }
        const float *p = img + (size_t)d * H * W + off;
        const int dx = dxdy & 1, dy = (dxdy >> 1) * W;
        out[i] = lerp2(p[0], p[dx], p[dy], p[dy + dx], xl, yl);
    }
}

__global__ __launch_bounds__(256) void car_bwd_nchw(const float *__restrict__ grads,
                                                    const float *__restrict__ boxes,
                                                    const int32_t *__restrict__ box_ind, int K,
                                                    int ch, int cw, int B, int C, int H, int W,
                                                    float *__restrict__ gimg, int32_t *err) {
    const int k = blockIdx.x;
    const int c0 = blockIdx.y * 32;
    const int nc = min(32, C - c0);
    const int bi = box_ind[k];
    const int ns = ch * cw;
    if (bi < 0 || bi >= B) {
        if (threadIdx.x == 0 && err) atomicOr(err, 1);
        return;
    }
    const float *g = grads + ((size_t)k * C + c0) * ns;
    float *img = gimg + ((size_t)bi * C + c0) * H * W;
    for (int i = threadIdx.x; i < nc * ns; i += 256) {
        const int d = i / ns, si = i - d * ns;
        const Sample s = make_sample(boxes + (size_t)k * 4, H, W, ch, cw, si / cw, si % cw);
        if (!s.valid) continue;
        float *p = img + (size_t)d * H * W;
        const float gv = g[i];
        const float dtop = (1 - s.yl) * gv;
        const float dbot = s.yl * gv;
        atomicAdd(p + s.top * W + s.lft, (1 - s.xl) * dtop);
        atomicAdd(p + s.top * W + s.rgt, s.xl * dtop);
        atomicAdd(p + s.bot * W + s.lft, (1 - s.xl) * dbot);
        atomicAdd(p + s.bot * W + s.rgt, s.xl * dbot);
    }
}

static inline int car_grid(long nsamp) {
    // memory-bound, grid-stride: ~8 blocks per CU (256 CUs) is enough to fill HBM
    long blocks = (nsamp + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int sln_crop_and_resize_fwd_f32(const float *image, int B, int C, int H, int W,
                                           int layout, const float *boxes, const int32_t *box_ind,
                                           int K, int ch, int cw, float extrap, float *crops,
                                           int32_t *err_flag, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || C < 0 || H < 1 || W < 1 || K < 0 || ch < 1 || cw < 1) return SLN_ERR_INVALID_ARG;
    if (K == 0 || C == 0) return SLN_OK;
    if (!image || !boxes || !box_ind || !crops) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (layout == SLN_LAYOUT_NHWC) {
        const long nsamp = (long)K * ch * cw;
        if (C % 4 == 0)
            hipLaunchKernelGGL(car_fwd_nhwc<4>, dim3(car_grid(nsamp)), dim3(256), 0, st, image, B, C,
                               H, W, boxes, box_ind, K, ch, cw, extrap, crops, err_flag);
        else
            hipLaunchKernelGGL(car_fwd_nhwc<1>, dim3(car_grid(nsamp)), dim3(256), 0, st, image, B, C,
                               H, W, boxes, box_ind, K, ch, cw, extrap, crops, err_flag);
    } else if (layout == SLN_LAYOUT_NCHW) {
        if (K > 2147483647 / 1 || sln_div_up(C, 32) > 65535) return SLN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(car_fwd_nchw, dim3(K, sln_div_up(C, 32)), dim3(256), 0, st, image, B, C, H,
                           W, boxes, box_ind, K, ch, cw, extrap, crops, err_flag);
    } else {
        return SLN_ERR_INVALID_ARG;
    }
    return sln_launch_status();
}

extern "C" int sln_crop_and_resize_bwd_f32(const float *grads, const float *boxes,
                                           const int32_t *box_ind, int K, int ch, int cw, int B,
                                           int C, int H, int W, int layout, float *grad_image,
                                           int32_t *err_flag, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || C < 0 || H < 1 || W < 1 || K < 0 || ch < 1 || cw < 1) return SLN_ERR_INVALID_ARG;
    if (layout != SLN_LAYOUT_NHWC && layout != SLN_LAYOUT_NCHW) return SLN_ERR_INVALID_ARG;
    if (B == 0 || C == 0) return SLN_OK;
    if (!grad_image) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // the callee zeroes the gradient image (crop_and_resize.c:182)
    if (hipMemsetAsync(grad_image, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess)
        return SLN_ERR_LAUNCH;
    if (K == 0) return SLN_OK;
    if (!grads || !boxes || !box_ind) return SLN_ERR_INVALID_ARG;
    if (layout == SLN_LAYOUT_NHWC) {
        const long nsamp = (long)K * ch * cw;
        hipLaunchKernelGGL(car_bwd_nhwc, dim3(car_grid(nsamp)), dim3(256), 0, st, grads, boxes,
                           box_ind, K, ch, cw, B, C, H, W, grad_image, err_flag);
    } else {
        if (sln_div_up(C, 32) > 65535) return SLN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(car_bwd_nchw, dim3(K, sln_div_up(C, 32)), dim3(256), 0, st, grads, boxes,
                           box_ind, K, ch, cw, B, C, H, W, grad_image, err_flag);
    }
    return sln_launch_status();
}
