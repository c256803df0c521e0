// Sem-dist ("layer") target decode on gfx950: uint64 occlusion labels ->
// per-instance bit planes, and the fused decode+crop+round mask-target kernel.
//
// Closed form of amodal_train.py:236-271 + modal/Functions.py:1012-1095:
//   low-word bit i  -> object i visible here  -> plane 0
//   high-word bit i -> object i occluded here -> plane min(rank+1, L-1),
//                      rank = popcount(hi & ((1<<i)-1))
// Integer/bit work, HBM-bound: 8 B read per pixel, N*L B written per pixel.
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned decode_byte(u64 v, int i, int l, int L) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    unsigned r = (l == 0) ? ((lo >> i) & 1u) : 0u;
    if ((hi >> i) & 1u) {
        const int rank = __popc(hi & ((1u << i) - 1u));
        const int pl = (rank + 1 >= L - 1) ? (L - 1) : (rank + 1);
        r |= (pl == l) ? 1u : 0u;
    }
    return r;
}

// max_objectID (Functions.py:1074-1079): OR together "highest set low-word bit"
// one-hot words, then count trailing ones.  One block per image slice; partial
// ORs are combined with an atomicOr into tops[b]; a second tiny kernel converts.
__global__ __launch_bounds__(256) void label_tops_kernel(const u64 *__restrict__ label, long npix,
                                                         unsigned *__restrict__ tops) {
    const int b = blockIdx.y;
    const u64 *lab = label + (size_t)b * npix;
    unsigned acc = 0;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const unsigned vis = (unsigned)lab[p];
        if (vis) acc |= 1u << (31 - __clz(vis));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(tops + b, acc);
}

// tops and n_obj alias (in-place conversion): no __restrict__ here.
__global__ void label_count_kernel(const unsigned *tops, int B, int32_t *n_obj) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const unsigned t = tops[b];
        n_obj[b] = (t == 0xffffffffu) ? 32 : (__ffs((int)~t) - 1);
    }
}

// Four pixels per thread: 32 B of labels in, one packed u32 (4 mask bytes) per
// (plane, object) out -> every store instruction writes 256 contiguous bytes.
__global__ __launch_bounds__(256) void label_decode_kernel(const u64 *__restrict__ label, long npix,
                                                           int L, int N, uint8_t *__restrict__ planes) {
    const int b = blockIdx.y;
    const u64 *lab = label + (size_t)b * npix;
    uint8_t *out = planes + (size_t)b * L * N * npix;
    const long nquad = (npix + 3) / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (long)gridDim.x * 256) {
        const long p0 = q * 4;
        u64 v[4];
        if (p0 + 3 < npix) {
            if ((((size_t)(lab + p0)) & 15) == 0) {
                const ulonglong2 a = *(const ulonglong2 *)(lab + p0);
                const ulonglong2 c = *(const ulonglong2 *)(lab + p0 + 2);
                v[0] = a.x; v[1] = a.y; v[2] = c.x; v[3] = c.y;
            } else {
                v[0] = lab[p0]; v[1] = lab[p0 + 1]; v[2] = lab[p0 + 2]; v[3] = lab[p0 + 3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (p0 + j < npix) ? lab[p0 + j] : 0ull;
        }
        const bool any = (v[0] | v[1] | v[2] | v[3]) != 0ull;
        for (int l = 0; l < L; ++l)
            for (int i = 0; i < N; ++i) {
                unsigned w = 0;
                if (any) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) w |= decode_byte(v[j], i, l, L) << (8 * j);
                }
                uint8_t *dst = out + ((size_t)l * N + i) * npix + p0;
                if (p0 + 3 < npix && (((size_t)dst) & 3) == 0) {
                    *(unsigned *)dst = w;
                } else {
                    for (int j = 0; j < 4 && p0 + j < npix; ++j) dst[j] = (uint8_t)(w >> (8 * j));
                }
            }
    }
}

// Fused mask targets: masks[k,l,y,x] = round_half_even(bilinear(plane_l of
// object roi_obj[k] in image roi_img[k])) with crop_and_resize.c's sampling
// (same coordinate arithmetic, -ffp-contract=off).  Reads 4 labels (32 B) per
// output sample instead of materialising [L,P,H,W] float planes.
__global__ __launch_bounds__(256) void mask_targets_kernel(const u64 *__restrict__ label, int B, int H,
                                                           int W, int L, const float *__restrict__ rois,
                                                           const int32_t *__restrict__ roi_img,
                                                           const int32_t *__restrict__ roi_obj, int K,
                                                           int mh, int mw, float *__restrict__ masks) {
    const long total = (long)K * mh * mw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % mw);
        const int y = (int)((idx / mw) % mh);
        const int k = (int)(idx / ((long)mw * mh));
        const int bi = roi_img[k], obj = roi_obj[k];
        float *out = masks + (size_t)k * L * mh * mw + (size_t)y * mw + x;
        bool ok = bi >= 0 && bi < B && obj >= 0 && obj < 32;
        const float *box = rois + (size_t)k * 4;
        const float y1 = box[0], x1 = box[1], y2 = box[2], x2 = box[3];
        const float hs = (mh > 1) ? (y2 - y1) * (float)(H - 1) / (float)(mh - 1) : 0.0f;
        const float ws = (mw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(mw - 1) : 0.0f;
        const float in_y = (mh > 1) ? y1 * (float)(H - 1) + (float)y * hs
                                    : (float)(0.5 * (double)(y1 + y2) * (double)(H - 1));
        const float in_x = (mw > 1) ? x1 * (float)(W - 1) + (float)x * ws
                                    : (float)(0.5 * (double)(x1 + x2) * (double)(W - 1));
        ok = ok && !(in_y < 0 || in_y > (float)(H - 1)) && !(in_x < 0 || in_x > (float)(W - 1));
        if (!ok) {  // extrapolation_value 0 (Functions.py:339)
            for (int l = 0; l < L; ++l) out[(size_t)l * mh * mw] = 0.0f;
            continue;
        }
        const float fy = floorf(in_y), fx = floorf(in_x);
        const int top = (int)fy, bot = (int)ceilf(in_y), lft = (int)fx, rgt = (int)ceilf(in_x);
        const float yl = in_y - fy, xl = in_x - fx;
        const u64 *lab = label + (size_t)bi * H * W;
        const u64 vtl = lab[(size_t)top * W + lft], vtr = lab[(size_t)top * W + rgt];
        const u64 vbl = lab[(size_t)bot * W + lft], vbr = lab[(size_t)bot * W + rgt];
        for (int l = 0; l < L; ++l) {
            const float tl = (float)decode_byte(vtl, obj, l, L), tr = (float)decode_byte(vtr, obj, l, L);
            const float bl = (float)decode_byte(vbl, obj, l, L), br = (float)decode_byte(vbr, obj, l, L);
            const float t = tl + (tr - tl) * xl;
            const float bt = bl + (br - bl) * xl;
            out[(size_t)l * mh * mw] = rintf(t + (bt - t) * yl);  // half-to-even
        }
    }
}

extern "C" int sln_label_num_objects_u64(const uint64_t *label, int B, int64_t npix, int32_t *n_obj,
                                         sln_stream_t stream) {
    sln_enter();
    if (B < 0 || npix < 0) return SLN_ERR_INVALID_ARG;
    if (B == 0) return SLN_OK;
    if (!n_obj || (!label && npix > 0)) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // n_obj doubles as the OR accumulator (same size), then is converted in place
    if (hipMemsetAsync(n_obj, 0, sizeof(int32_t) * (size_t)B, st) != hipSuccess) return SLN_ERR_LAUNCH;
    if (npix > 0) {
        int gx = sln_div_up(npix, 256 * 8);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(label_tops_kernel, dim3(gx, B), dim3(256), 0, st, (const u64 *)label,
                           (long)npix, (unsigned *)n_obj);
    }
    hipLaunchKernelGGL(label_count_kernel, dim3(sln_div_up(B, 64)), dim3(64), 0, st,
                       (const unsigned *)n_obj, B, n_obj);
    return sln_launch_status();
}

// ---------------------------------------------------------------- loader front end (round 5)
// Nearest-neighbour zoom of the uint64 labels of a batch to the network size, on the device: the reference's
// utils.resize_layer (utils.py:358-362) is scipy.ndimage.zoom(order=0) of the decoded planes on a DataLoader worker
// (12 ms per 1024^2 image on a host core); the label itself is zoomed here instead (the decode is per pixel, so
// decoding the zoomed label equals zooming the decoded planes) with scipy's index maps, which the host computes
// once per image size (utils.zoom_nearest_index: OH + OW integers, -1 = the constant fill 0).  A horizontal flip
// (Functions.py:713) is the reversed xs map.  Image b's label lies at src + b * src_stride with row pitch
// src_hw[2b+1]; 16 B written per lane (two output pixels), gathered 8-B reads.
__global__ __launch_bounds__(256) void label_zoom_kernel(const u64 *__restrict__ src, long src_stride,
                                                         const int32_t *__restrict__ src_hw,
                                                         const int32_t *__restrict__ ys,
                                                         const int32_t *__restrict__ xs, int OH, int OW,
                                                         u64 *__restrict__ out) {
    const int b = blockIdx.z, oy = blockIdx.y;
    const int W0 = src_hw[2 * b + 1];
    const int sy = ys[(long)b * OH + oy];
    const u64 *row = src + (long)b * src_stride + (long)(sy < 0 ? 0 : sy) * W0;
    u64 *dst = out + ((long)b * OH + oy) * OW;
    const int32_t *xb = xs + (long)b * OW;
    for (int ox = (blockIdx.x * 256 + threadIdx.x) * 2; ox < OW; ox += gridDim.x * 512) {
        const int x0 = xb[ox], x1 = ox + 1 < OW ? xb[ox + 1] : -1;
        //  src_hw is device-side metadata: a row that disagrees with the packed stride (a worker's error
        // path, another batch's sizes) must read the constant fill, not the next image's label or past the buffer
        const long i0 = (long)(sy < 0 ? 0 : sy) * W0 + x0, i1 = (long)(sy < 0 ? 0 : sy) * W0 + x1;
        const u64 v0 = (sy >= 0 && x0 >= 0 && x0 < W0 && i0 < src_stride) ? row[x0] : 0ull;
        const u64 v1 = (sy >= 0 && x1 >= 0 && x1 < W0 && i1 < src_stride) ? row[x1] : 0ull;
        if (ox + 1 < OW && (OW & 1) == 0) {
            *(ulonglong2 *)(dst + ox) = make_ulonglong2(v0, v1);
        } else {
            dst[ox] = v0;
            if (ox + 1 < OW) dst[ox + 1] = v1;
        }
    }
}

extern "C" int sln_label_zoom_u64(const uint64_t *src, int64_t src_stride, const int32_t *src_hw, const int32_t *ys,
                                  const int32_t *xs, int B, int OH, int OW, uint64_t *out, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || OH < 0 || OW < 0 || src_stride < 0) return SLN_ERR_INVALID_ARG;
    if (B == 0 || OH == 0 || OW == 0) return SLN_OK;
    if (!src || !src_hw || !ys || !xs || !out || B > 65535 || OH > 65535) return SLN_ERR_INVALID_ARG;
    int gx = sln_div_up(OW, 512);
    hipLaunchKernelGGL(label_zoom_kernel, dim3(gx, OH, B), dim3(256), 0, (hipStream_t)stream, (const u64 *)src,
                       (long)src_stride, src_hw, ys, xs, OH, OW, (u64 *)out);
    return sln_launch_status();
}

// Object counts of labels of DIFFERENT sizes packed at a fixed stride (the original, un-zoomed labels of a batch:
// load_layer2 counts before the resize): image b holds src_hw[2b] * src_hw[2b+1] valid pixels.
__global__ __launch_bounds__(256) void label_tops_ragged_kernel(const u64 *__restrict__ label, long stride,
                                                                const int32_t *__restrict__ src_hw,
                                                                unsigned *__restrict__ tops) {
    const int b = blockIdx.y;
    const u64 *lab = label + (size_t)b * stride;
    long npix = (long)src_hw[2 * b] * src_hw[2 * b + 1];
    if (npix > stride) npix = stride;          // (never past this image's slot of the packed batch)
    if (npix < 0) npix = 0;
    unsigned acc = 0;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const unsigned vis = (unsigned)lab[p];
        if (vis) acc |= 1u << (31 - __clz(vis));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(tops + b, acc);
}

extern "C" int sln_label_num_objects_ragged_u64(const uint64_t *label, int B, int64_t stride, const int32_t *src_hw,
                                                int32_t *n_obj, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || stride < 0) return SLN_ERR_INVALID_ARG;
    if (B == 0) return SLN_OK;
    if (!n_obj || !label || !src_hw || B > 65535) return SLN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(n_obj, 0, sizeof(int32_t) * (size_t)B, st) != hipSuccess) return SLN_ERR_LAUNCH;
    int gx = sln_div_up(stride > 0 ? stride : 1, 256 * 8);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(label_tops_ragged_kernel, dim3(gx, B), dim3(256), 0, st, (const u64 *)label, (long)stride,
                       src_hw, (unsigned *)n_obj);
    hipLaunchKernelGGL(label_count_kernel, dim3(sln_div_up(B, 64)), dim3(64), 0, st, (const unsigned *)n_obj, B, n_obj);
    return sln_launch_status();
}

extern "C" int sln_label_decode_u64(const uint64_t *label, int B, int H, int W, int L, int N,
                                    uint8_t *planes, sln_stream_t stream) {
    sln_enter();
    if (B < 0 || H < 0 || W < 0 || L < 1 || N < 0 || N > 32) return SLN_ERR_INVALID_ARG;
    const long npix = (long)H * W;
    if (B == 0 || N == 0 || npix == 0) return SLN_OK;
    if (!label || !planes) return SLN_ERR_INVALID_ARG;
    int gx = sln_div_up(npix, 256 * 4);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(label_decode_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream,
                       (const u64 *)label, npix, L, N, planes);
    return sln_launch_status();
}

extern "C" int sln_mask_targets_u64(const uint64_t *label, int B, int H, int W, int L,
                                    const float *rois, const int32_t *roi_img,
                                    const int32_t *roi_obj, int K, int mh, int mw, float *masks,
                                    sln_stream_t stream) {
    sln_enter();
    if (B < 0 || H < 1 || W < 1 || L < 1 || K < 0 || mh < 1 || mw < 1) return SLN_ERR_INVALID_ARG;
    if (K == 0) return SLN_OK;
    if (!label || !rois || !roi_img || !roi_obj || !masks) return SLN_ERR_INVALID_ARG;
    const long total = (long)K * mh * mw;
    int gx = sln_div_up(total, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(mask_targets_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream,
                       (const u64 *)label, B, H, W, L, rois, roi_img, roi_obj, K, mh, mw, masks);
    return sln_launch_status();
}
