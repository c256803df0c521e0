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
template <int VEC>
__global__ __launch_bounds__(256) void pyr_bwd_patch_kernel(PyrGrads gm, int B, int C,
                                                            const float *__restrict__ grads,
                                                            int g_cstride, int g_coffset,
                                                            const float *__restrict__ boxes,
                                                            const int32_t *__restrict__ box_ind,
                                                            const int32_t *__restrict__ level,
                                                            int K, int ch, int cw) {
    // VEC channels per lane (4: 16-byte loads, a block = one roi x 256 channels).  Everything but the loads and
    // the adds is the same for all lanes of a wave (the pixel and its taps), so that part -- LDS list reads,
    // address arithmetic: 0.34 ms of 0.57 with one channel per lane -- is paid once per VEC x 64 channels.
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    __shared__ float s_tr[VEC > 1 ? 4 : 1][2][64 * VEC];   // per wave: two pixels' sums, re-read channel-contiguous
    __shared__ int s_lo[2][PYR_MAXS], s_hi[2][PYR_MAXS], s_ok[2][PYR_MAXS];
    __shared__ float s_fr[2][PYR_MAXS];
    __shared__ int s_start[2][PYR_MAXP + 2], s_cur[2][PYR_MAXP + 1];
    __shared__ int s_ei[2][2 * PYR_MAXS];
    __shared__ float s_ew[2][2 * PYR_MAXS];
    __shared__ int s_ext[4];
    const int k = blockIdx.x, lane = threadIdx.x & 63, c0 = blockIdx.y * 64 * VEC, c = c0 + lane * VEC;
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
    if (ph <= 0 || pw <= 0) return;
    const bool cok = c < C;                               // C % VEC == 0: a lane's channels are all in or all out
    const float *g = grads + (size_t)k * ch * cw * g_cstride + g_coffset + (cok ? c : 0);
    float *img = gm.ptr[lv] + (size_t)bi * H * W * C;
    if (ph > PYR_MAXP || pw > PYR_MAXP || ph * pw > 2 * ch * cw) {
        // footprint larger than the crop: per-bin scatter is the cheaper side
        if (!cok) return;
        for (int s = wv + 4 * blockIdx.z; s < ch * cw; s += 4 * gridDim.z) {
            const int y = s / cw, x = s - y * cw;
            if (!s_ok[0][y] || !s_ok[1][x]) continue;
            const float yl = s_fr[0][y], xl = s_fr[1][x];
            const vecf gq = *(const vecf *)(g + (size_t)s * g_cstride);
            float *rt = img + (size_t)s_lo[0][y] * W * C + c, *rb = img + (size_t)s_hi[0][y] * W * C + c;
            const size_t l = (size_t)s_lo[1][x] * C, r = (size_t)s_hi[1][x] * C;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float gv = gq[i];
                const float dtop = (1 - yl) * gv, dbot = yl * gv;
                atomicAdd(rt + l + i, (1 - xl) * dtop);
                atomicAdd(rt + r + i, xl * dtop);
                atomicAdd(rb + l + i, (1 - xl) * dbot);
                atomicAdd(rb + r + i, xl * dbot);
            }
        }
        return;
    }
    // two footprint pixels per wave and trip, their taps (ny x nx, mostly 2 x 2) flattened and fetched four at a
    // time: up to eight loads in flight per lane.  Tail taps repeat the first one with weight 0.
    const int npix = ph * pw;
    for (int p = wv + 8 * blockIdx.z; p < npix; p += 8 * gridDim.z) {   // blockIdx.z: a share of the pixels
        int ys[2], xs[2], nx[2], nt[2];
        vecf acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = min(p + 4 * u, npix - 1);
            const int pr = q / pw, pc = q - pr * pw;
            ys[u] = s_start[0][pr]; xs[u] = s_start[1][pc];
            nx[u] = s_start[1][pc + 1] - xs[u];
            nt[u] = (p + 4 * u < npix) ? (s_start[0][pr + 1] - ys[u]) * nx[u] : 0;
            acc[u] = (vecf)(0.0f);
        }
        const int trips = (max(nt[0], nt[1]) + 3) >> 2;
        int e[2] = {0, 0}, f[2] = {0, 0};
        for (int b = 0; b < trips; ++b) {
            vecf gv[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool live = 4 * b + v < nt[u];
                    const int ye = ys[u] + (live ? e[u] : 0), xf = xs[u] + (live ? f[u] : 0);
                    const float wy = s_ew[0][ye], wx = s_ew[1][xf];
                    vecf x = (vecf)(0.0f);
                    if (nt[u]) x = *(const vecf *)(g + ((size_t)s_ei[0][ye] * cw + s_ei[1][xf]) * g_cstride);
                    gv[u][v] = live ? wx * (wy * x) : (vecf)(0.0f);
                    if (++f[u] == nx[u]) { f[u] = 0; ++e[u]; }
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u] += gv[u][v];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!nt[u]) continue;                          // wave-uniform
            const int q = p + 4 * u;
            const int pr = q / pw, pc = q - pr * pw;
            float *dst = img + ((size_t)(p0y + pr) * W + (p0x + pc)) * C;
            if constexpr (VEC == 1) {
                if (cok) atomicAdd(dst + c, acc[u][0]);
            } else {
                // lane L holds channels c0 + 4L..4L+3; through LDS so that each atomic instruction covers 64
                // consecutive channels (256 B, two lines) instead of 64 pieces at 16-byte stride
                *(vecf *)&s_tr[wv][u][lane * VEC] = acc[u];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int cc = c0 + 64 * i + lane;
                    if (cc < C) atomicAdd(dst + cc, s_tr[wv][u][64 * i + lane]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// ---------------------------------------------------------------- backward as a write-once gather
// The scatter kernels above need the four gradient maps zeroed first (1.43 GB at 16 x 1024^2) and add with
// atomics in an order that changes from run to run.  Here every 8x8 tile of every map is a block that GATHERS
// what the rois of its image contribute to its 64 pixels and writes them once -- zeros included: no zero fill, no
// atomics, the same bits on every run; several crops of the same maps (classifier 7x7 + mask 16x16) go through
// one launch.  Two kernels: (1) per (image, level) the list of rois, in roi order (one wave each, ballot
// compaction); (2) the gather: rois of the list whose footprint meets the tile are found in parallel, then for
// each of them the per-axis taps (pyr_axis: the reference's arithmetic, crop_and_resize.c:56-78) are inverted
// per map row / column of the tile in LDS, and every wave sums w_x * (w_y * g) over the bins that touch its
// pixels (lanes = channels).  The individual products are the reference's (:169-186); the order of the additions
// is fixed (list order, then bin order).
#define PYG_T 8            // tile edge
#define PYG_MAXSRC 4
struct PygSrc {
    const float *grads;    // [K][ch][cw][cstride], channels coffset..coffset+C
    const float *boxes;    // [K][4]
    const int32_t *box_ind, *level;
    int K, ch, cw, cstride, coffset, first;   // first = index of this source's roi 0 in the joint numbering
};
struct PygSrcs {
    PygSrc s[PYG_MAXSRC];
    int n, total;          // total = sum of K
};

// lists [B][4][total] joint roi numbers (source-major), counts [B][4]; grid = B * 4 waves
__global__ __launch_bounds__(64) void pyg_list_kernel(PygSrcs src, int B, int32_t *__restrict__ lists,
                                                      int32_t *__restrict__ counts) {
    const int b = blockIdx.x >> 2, lv = blockIdx.x & 3, lane = threadIdx.x;
    int32_t *out = lists + ((size_t)b * 4 + lv) * src.total;
    int n = 0;
    for (int si = 0; si < src.n; ++si) {
        const PygSrc &q = src.s[si];
        for (int k0 = 0; k0 < q.K; k0 += 64) {
            const int k = k0 + lane;
            const bool hit = k < q.K && q.box_ind[k] == b && q.level[k] - 2 == lv;
            const unsigned long long m = __ballot(hit);
            if (hit) out[n + __popcll(m & ((1ull << lane) - 1ull))] = q.first + k;
            n += __popcll(m);
        }
    }
    if (lane == 0) counts[b * 4 + lv] = n;
}

struct PygMaps {
    float *ptr[4];
    int H[4], W[4];
    int tiles_x[4], first_block[5];   // per level: tiles per row, first block id (blocks = B * ty * tx * cgroups)
};

__global__ __launch_bounds__(256) void pyg_gather_kernel(PygMaps gm, PygSrcs src, int B, int C, int cgroups,
                                                         const int32_t *__restrict__ lists,
                                                         const int32_t *__restrict__ counts) {
    __shared__ int s_hit[1024];                        // joint roi numbers of the list that meet this tile
    __shared__ int s_nhit, s_wcnt[4];
    __shared__ int s_lo[2][32], s_hi[2][32], s_ok[2][32];
    __shared__ float s_fr[2][32];
    __shared__ int s_n[2][PYG_T], s_bin[2][PYG_T][64];
    __shared__ float s_w[2][PYG_T][64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int lv = 0;
    while (lv < 3 && (int)blockIdx.x >= gm.first_block[lv + 1]) ++lv;
    int rest = blockIdx.x - gm.first_block[lv];
    const int H = gm.H[lv], W = gm.W[lv];
    const int tx_n = gm.tiles_x[lv], ty_n = (H + PYG_T - 1) / PYG_T;
    const int cg = rest % cgroups; rest /= cgroups;
    const int tx = rest % tx_n; rest /= tx_n;
    const int ty = rest % ty_n;
    const int b = rest / ty_n;
    const int r0 = ty * PYG_T, c0 = tx * PYG_T;
    const int c = cg * 64 + lane;
    const int cnt = counts[b * 4 + lv];
    const int32_t *list = lists + ((size_t)b * 4 + lv) * src.total;

    auto locate = [&](int joint, int &si, int &k) {
        si = 0;
        while (si + 1 < src.n && joint >= src.s[si + 1].first) ++si;
        k = joint - src.s[si].first;
    };
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int seg = 0; seg < cnt; seg += 1024) {      // (the hit list holds 1024 entries: longer lists in segments)
    // ---- rois whose (conservative) footprint meets the tile, in list order
    if (t == 0) s_nhit = 0;
    __syncthreads();
    const int seg_end = min(cnt, seg + 1024);
    for (int base = seg; base < seg_end; base += 256) {
        bool hit = false;
        int joint = 0;
        if (base + t < seg_end) {
            joint = list[base + t];
            int si, k;
            locate(joint, si, k);
            const PygSrc &q = src.s[si];
            const float *bx = q.boxes + (size_t)k * 4;
            // input coordinates of the first and last bin per axis (crop == 1: the centre, pyr_axis)
            const float ya = bx[0] * (float)(H - 1), yb = bx[2] * (float)(H - 1);
            const float xa = bx[1] * (float)(W - 1), xb = bx[3] * (float)(W - 1);
            const float ylo = fminf(ya, yb), yhi = fmaxf(ya, yb), xlo = fminf(xa, xb), xhi = fmaxf(xa, xb);
            hit = yhi >= (float)(r0 - 1) && ylo <= (float)(r0 + PYG_T) && xhi >= (float)(c0 - 1) &&
                  xlo <= (float)(c0 + PYG_T);
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wcnt[wv] = __popcll(m);
        __syncthreads();
        int off = s_nhit;
        for (int w = 0; w < wv; ++w) off += s_wcnt[w];
        if (hit) s_hit[off + __popcll(m & ((1ull << lane) - 1ull))] = joint;
        __syncthreads();
        if (t == 0) s_nhit += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int nhit = s_nhit;
    for (int h = 0; h < nhit; ++h) {
        int si, k;
        locate(s_hit[h], si, k);
        const PygSrc &q = src.s[si];
        const float *bx = q.boxes + (size_t)k * 4;
        // per-axis taps of every bin (threads 0..63: axis t / 32, bin t % 32)
        if (t < 64) {
            const int ax = t >> 5, i = t & 31, nb = ax ? q.cw : q.ch;
            if (i < nb) {
                int lo = 0, hi = 0; float fr = 0;
                const bool ok = ax ? pyr_axis(bx[1], bx[3], W, q.cw, i, lo, hi, fr)
                                   : pyr_axis(bx[0], bx[2], H, q.ch, i, lo, hi, fr);
                s_lo[ax][i] = lo; s_hi[ax][i] = hi; s_fr[ax][i] = fr; s_ok[ax][i] = ok;
            }
        }
        __syncthreads();
        // inverted per map row / column of the tile (threads 0..15: axis t / 8, line t % 8), bins ascending
        if (t < 2 * PYG_T) {
            const int ax = t / PYG_T, ln = t % PYG_T, nb = ax ? q.cw : q.ch;
            const int pos = (ax ? c0 : r0) + ln;
            int n = 0;
            for (int i = 0; i < nb; ++i) {
                if (!s_ok[ax][i]) continue;
                if (s_lo[ax][i] == pos) { s_bin[ax][ln][n] = i; s_w[ax][ln][n] = 1 - s_fr[ax][i]; ++n; }
                if (s_hi[ax][i] == pos && s_hi[ax][i] != s_lo[ax][i]) {   // hi == lo: frac is 0, adds 0
                    s_bin[ax][ln][n] = i; s_w[ax][ln][n] = s_fr[ax][i]; ++n;
                }
            }
            s_n[ax][ln] = n;
        }
        __syncthreads();
        if (c < C) {
            const float *g = q.grads + (size_t)k * q.ch * q.cw * q.cstride + q.coffset + c;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pp = wv + 4 * j, pr = pp / PYG_T, pc = pp % PYG_T;
                const int ny = s_n[0][pr], nx = s_n[1][pc];
                if (ny == 0 || nx == 0) continue;
                float a = acc[j];
                for (int e = 0; e < ny; ++e) {
                    const float wy = s_w[0][pr][e];
                    const float *gy = g + (size_t)s_bin[0][pr][e] * q.cw * q.cstride;
                    // four loads in flight per wave (a tail entry repeats the last bin with weight 0: + 0 exactly)
                    for (int f = 0; f < nx; f += 4) {
                        float gv[4], wx[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int ff = min(f + u, nx - 1);
                            wx[u] = (f + u < nx) ? s_w[1][pc][ff] : 0.f;
                            gv[u] = gy[(size_t)s_bin[1][pc][ff] * q.cstride];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) a += wx[u] * (wy * gv[u]);
                    }
                }
                acc[j] = a;
            }
        }
        __syncthreads();
    }
    }
    if (c < C) {
        float *img = gm.ptr[lv] + (size_t)b * H * W * C + c;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pp = wv + 4 * j, r = r0 + pp / PYG_T, cc = c0 + pp % PYG_T;
            if (r < H && cc < W) img[((size_t)r * W + cc) * C] = acc[j];
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

// Zero fill of the four gradient maps in ONE launch (four memsets cost 0.21 ms at 16 x 1024^2 -- 1.7 TB/s, the
// three small ones are launch-bound; one grid of 16-byte stores runs at the store rate of the chip).
struct PyrZero {
    float4 *ptr[4];
    long end[4];            // running end of each map in float4 units
};

__global__ void __launch_bounds__(256) pyr_zero_kernel(PyrZero z) {
    // one 16-byte store per thread: the fastest fill shape measured (tools/micro/fill_rate.hip: 6.8 TB/s; a
    // grid-stride loop of 8192 blocks 4.6, hipMemsetAsync 6.5 for one region -- but four launches here)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= z.end[3]) return;
    const int m = (i >= z.end[0]) + (i >= z.end[1]) + (i >= z.end[2]);
    z.ptr[m][i - (m ? z.end[m - 1] : 0)] = make_float4(0.f, 0.f, 0.f, 0.f);
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
    }
    if (!accumulate && B > 0) {
        PyrZero z;
        bool vec = true;
        long run = 0;
        for (int i = 0; i < 4; ++i) {
            const size_t n = (size_t)B * gm.H[i] * gm.W[i] * C;
            vec = vec && n % 4 == 0 && (((size_t)gm.ptr[i]) & 15) == 0;
            z.ptr[i] = (float4 *)gm.ptr[i];
            run += (long)(n / 4);
            z.end[i] = run;
        }
        if (vec) {
            if (run > 0)
                hipLaunchKernelGGL(pyr_zero_kernel, dim3((unsigned)((run + 255) / 256)), dim3(256), 0, st, z);
        } else {
            for (int i = 0; i < 4; ++i)
                if (hipMemsetAsync(gm.ptr[i], 0, sizeof(float) * (size_t)B * gm.H[i] * gm.W[i] * C, st) != hipSuccess)
                    return SLN_ERR_LAUNCH;
        }
    }
    if (K == 0 || B == 0) return sln_launch_status();
    if (!grads || !boxes || !box_ind || !level) return SLN_ERR_INVALID_ARG;
    const long nsamp = (long)K * ch * cw;
    if (ch * cw >= 64 && ch <= PYR_MAXS && cw <= PYR_MAXS) {   // dense crops: footprint gather
        const bool vec = C % 4 == 0 && g_cstride % 4 == 0 && g_coffset % 4 == 0 && (((size_t)grads) & 15) == 0;
        if (vec)
            hipLaunchKernelGGL(pyr_bwd_patch_kernel<4>, dim3(K, (C + 255) / 256, 4), dim3(256), 0, st, gm, B, C,
                               grads, g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
        else
            hipLaunchKernelGGL(pyr_bwd_patch_kernel<1>, dim3(K, (C + 63) / 64), dim3(256), 0, st, gm, B, C,
                               grads, g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
    }
    else
        hipLaunchKernelGGL(pyr_bwd_kernel, dim3(pyr_grid(nsamp)), dim3(256), 0, st, gm, B, C, grads,
                           g_cstride, g_coffset, boxes, box_ind, level, K, ch, cw);
    return sln_launch_status();
}

// Workspace of sln_pyramid_crop_bwd_gather_f32 for `total` rois (all sources) of B images.
extern "C" size_t sln_pyramid_crop_bwd_gather_workspace_bytes(int total_rois, int B) {
    if (total_rois < 0 || B < 0) return 0;
    return sizeof(int32_t) * ((size_t)B * 4 * (size_t)(total_rois > 0 ? total_rois : 1) + (size_t)B * 4);
}

extern "C" int sln_pyramid_crop_bwd_gather_f32(int nsrc, const float *const *grads, const int32_t *g_cstride,
                                               const int32_t *g_coffset, const float *const *boxes,
                                               const int32_t *const *box_ind, const int32_t *const *level,
                                               const int32_t *K, const int32_t *ch, const int32_t *cw, int B, int C,
                                               float *const *grad_maps, const int32_t *map_hw, void *workspace,
                                               size_t workspace_bytes, sln_stream_t stream) {
    sln_enter();
    if (nsrc < 1 || nsrc > PYG_MAXSRC || !grads || !g_cstride || !g_coffset || !boxes || !box_ind || !level || !K ||
        !ch || !cw || !grad_maps || !map_hw || B < 0 || C < 1)
        return SLN_ERR_INVALID_ARG;
    PygSrcs src;
    src.n = nsrc; src.total = 0;
    for (int i = 0; i < nsrc; ++i) {
        PygSrc &q = src.s[i];
        q.grads = grads[i]; q.boxes = boxes[i]; q.box_ind = box_ind[i]; q.level = level[i];
        q.K = K[i]; q.ch = ch[i]; q.cw = cw[i]; q.cstride = g_cstride[i]; q.coffset = g_coffset[i];
        q.first = src.total;
        if (q.K < 0 || q.ch < 1 || q.cw < 1 || q.ch > 32 || q.cw > 32 || q.coffset < 0 || q.cstride < C + q.coffset)
            return SLN_ERR_INVALID_ARG;
        if (q.K > 0 && (!q.grads || !q.boxes || !q.box_ind || !q.level)) return SLN_ERR_INVALID_ARG;
        src.total += q.K;
    }
    if (B == 0) return SLN_OK;
    if (!workspace || workspace_bytes < sln_pyramid_crop_bwd_gather_workspace_bytes(src.total, B))
        return SLN_ERR_WORKSPACE;
    if (src.total < 1) src.total = 1;
    PygMaps gm;
    const int cgroups = (C + 63) / 64;
    long blocks = 0;
    for (int i = 0; i < 4; ++i) {
        gm.ptr[i] = grad_maps[i]; gm.H[i] = map_hw[2 * i]; gm.W[i] = map_hw[2 * i + 1];
        if (!gm.ptr[i] || gm.H[i] < 1 || gm.W[i] < 1) return SLN_ERR_INVALID_ARG;
        gm.tiles_x[i] = (gm.W[i] + PYG_T - 1) / PYG_T;
        gm.first_block[i] = (int)blocks;
        blocks += (long)B * ((gm.H[i] + PYG_T - 1) / PYG_T) * gm.tiles_x[i] * cgroups;
        if (blocks > 2147483647L) return SLN_ERR_UNSUPPORTED;
    }
    gm.first_block[4] = (int)blocks;
    int32_t *lists = (int32_t *)workspace;
    int32_t *counts = lists + (size_t)B * 4 * src.total;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pyg_list_kernel, dim3(B * 4), dim3(64), 0, st, src, B, lists, counts);
    hipLaunchKernelGGL(pyg_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gm, src, B, C, cgroups,
                       (const int32_t *)lists, (const int32_t *)counts);
    return sln_launch_status();
}
