/*
 * sln_oracle.c -- CPU restatement of the SLN-Amodal hot-path native arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (sln_amodal_amd/) may
 * import, link or execute this file.  It is loaded by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- there only as
 * the checker / the CPU baseline, never as the thing shipped.
 *
 * PARITY PINNING STATUS
 *   - orc_nms_f32, orc_crop_and_resize_{fwd,bwd}_f32: "parity unpinned" by a
 *     run of the reference itself.  The reference C sources include <TH/TH.h>
 *     (PyTorch 0.4 TH API), which this image lacks, so they are unbuildable
 *     here and the reference holds no tests or golden vectors for them
 *     (SURVEY.md section 4).  They are restated line by line from the cited
 *     source and cross-checked against independent formulations
 *     (tests/test_oracle_cpu.py: numpy greedy NMS, torch grid_sample).
 *   - orc_label_decode_u64, orc_box_*: pinned against outputs of the reference's
 *     own Python (tools/gen_golden.py -> tests/golden/).
 *   - orc_rle_*: pinned against oracle/_ref/libmaskapi_ref.so, the reference's own
 *     cocoapi/common/maskApi.c compiled by oracle/Makefile (target `ref`), and
 *     against tests/golden/rle.npz generated from it.
 *   - orc_bytescale_f32 / orc_pil_resize_bilinear_u8 / orc_unmold_mask_f32: the
 *     algorithm lives in third-party scipy.misc + Pillow (not vendored by the
 *     reference); pinned against the installed Pillow and tests/golden/unmold.npz.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math, so
 * every float expression rounds exactly once per operation like the
 * reference's x86-64 gcc build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Greedy NMS.                                                                */
/* Follows /root/reference/nms/pth_nms.py:10-24 (areas, order) and            */
/* /root/reference/nms/src/nms.c:33-63 (the suppression loop).                */
/* dets: [n,5] rows (y1,x1,y2,x2,score).  keep: caller-sized [n].             */
/* Score ties: the reference sorts with torch 0.4's unstable sort; this       */
/* restatement (and the HIP path) define the tie-break as index-ascending.    */
/* ------------------------------------------------------------------------ */
typedef struct { float s; int64_t i; } orc_si;

static int orc_cmp_desc(const void *a, const void *b) {
    const orc_si *x = (const orc_si *)a, *y = (const orc_si *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int orc_nms_f32(const float *dets, int64_t n, float thresh, int64_t *keep,
                int64_t *num_out) {
    if (n < 0) return 1;
    float *areas = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    orc_si *order = (orc_si *)malloc(sizeof(orc_si) * (size_t)(n > 0 ? n : 1));
    unsigned char *sup = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!areas || !order || !sup) { free(areas); free(order); free(sup); return 2; }
    for (int64_t i = 0; i < n; ++i) {
        const float y1 = dets[i * 5 + 0], x1 = dets[i * 5 + 1];
        const float y2 = dets[i * 5 + 2], x2 = dets[i * 5 + 3];
        /* pth_nms.py:16  areas = (x2 - x1 + 1) * (y2 - y1 + 1), fp32 elementwise */
        const float w = (x2 - x1) + 1.0f;
        const float h = (y2 - y1) + 1.0f;
        areas[i] = w * h;
        order[i].s = dets[i * 5 + 4];
        order[i].i = i;
    }
    qsort(order, (size_t)n, sizeof(orc_si), orc_cmp_desc);

    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        /* nms.c:40-43 reads columns 0..3 as x1,y1,x2,y2; the caller feeds
           (y1,x1,y2,x2).  IoU is symmetric under the swap; names follow nms.c. */
        const float ix1 = dets[i * 5 + 0], iy1 = dets[i * 5 + 1];
        const float ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
        const float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j].i;
            if (sup[j]) continue;
            const float xx1 = fmaxf(ix1, dets[j * 5 + 0]);
            const float yy1 = fmaxf(iy1, dets[j * 5 + 1]);
            const float xx2 = fminf(ix2, dets[j * 5 + 2]);
            const float yy2 = fminf(iy2, dets[j * 5 + 3]);
            const float w = fmaxf(0.0f, (xx2 - xx1) + 1.0f);
            const float h = fmaxf(0.0f, (yy2 - yy1) + 1.0f);
            const float inter = w * h;
            const float ovr = inter / ((iarea + areas[j]) - inter);
            if (ovr >= thresh) sup[j] = 1; /* nms.c:60 : >= (CPU semantics) */
        }
    }
    *num_out = nk;
    free(areas); free(order); free(sup);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* crop_and_resize forward.                                                   */
/* Follows /root/reference/roialign/roi_align/src/crop_and_resize.c:6-112.    */
/* image [B,C,H,W] f32 NCHW; boxes [K,4] normalised (y1,x1,y2,x2);            */
/* box_ind [K] int32; crops [K,C,ch,cw] (caller allocated, fully written).    */
/* Returns 0, or 3 when a box index is out of range (the reference exit(-1)s).*/
/* ------------------------------------------------------------------------ */
static inline float orc_in_coord(float lo, float hi, int size, int crop, int idx,
                                 float scale) {
    /* crop_and_resize.c:54-56: the crop==1 branch multiplies by the double
       literal 0.5, so it is evaluated in double and narrowed on assignment. */
    if (crop > 1) return lo * (float)(size - 1) + (float)idx * scale;
    return (float)(0.5 * (double)(lo + hi) * (double)(size - 1));
}

int orc_crop_and_resize_fwd_f32(const float *image, int B, int C, int H, int W,
                                const float *boxes, const int32_t *box_ind, int K,
                                int ch, int cw, float extrap, float *crops) {
    const int64_t plane = (int64_t)H * W, img_elems = (int64_t)C * plane;
    const int64_t cplane = (int64_t)ch * cw, crop_elems = (int64_t)C * cplane;
    for (int b = 0; b < K; ++b)
        if (box_ind[b] < 0 || box_ind[b] >= B) return 3;
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < K; ++b) {
        const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
        const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        const int b_in = box_ind[b];
        const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
        const float ws = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
        float *out = crops + crop_elems * b;
        for (int y = 0; y < ch; ++y) {
            const float in_y = orc_in_coord(y1, y2, H, ch, y, hs);
            if (in_y < 0 || in_y > (float)(H - 1)) {
                for (int x = 0; x < cw; ++x)
                    for (int d = 0; d < C; ++d) out[cplane * d + y * cw + x] = extrap;
                continue;
            }
            const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
            const float yl = in_y - (float)top;
            for (int x = 0; x < cw; ++x) {
                const float in_x = orc_in_coord(x1, x2, W, cw, x, ws);
                if (in_x < 0 || in_x > (float)(W - 1)) {
                    for (int d = 0; d < C; ++d) out[cplane * d + y * cw + x] = extrap;
                    continue;
                }
                const int lft = (int)floorf(in_x), rgt = (int)ceilf(in_x);
                const float xl = in_x - (float)lft;
                for (int d = 0; d < C; ++d) {
                    const float *p = image + b_in * img_elems + d * plane;
                    const float tl = p[(int64_t)top * W + lft], tr = p[(int64_t)top * W + rgt];
                    const float bl = p[(int64_t)bot * W + lft], br = p[(int64_t)bot * W + rgt];
                    const float t = tl + (tr - tl) * xl;
                    const float bt = bl + (br - bl) * xl;
                    out[cplane * d + y * cw + x] = t + (bt - t) * yl;
                }
            }
        }
    }
    return 0;
}

/* crop_and_resize backward: crop_and_resize.c:157-252.  Serial over boxes in
 * (b, y, x, d) order, so the fp32 summation order is the reference's. */
int orc_crop_and_resize_bwd_f32(const float *grads, const float *boxes,
                                const int32_t *box_ind, int K, int ch, int cw, int B,
                                int C, int H, int W, float *grads_image) {
    const int64_t plane = (int64_t)H * W, img_elems = (int64_t)C * plane;
    const int64_t cplane = (int64_t)ch * cw, crop_elems = (int64_t)C * cplane;
    memset(grads_image, 0, sizeof(float) * (size_t)B * (size_t)img_elems);
    for (int b = 0; b < K; ++b) {
        const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
        const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        const int b_in = box_ind[b];
        if (b_in < 0 || b_in >= B) return 3;
        const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
        const float ws = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
        for (int y = 0; y < ch; ++y) {
            const float in_y = orc_in_coord(y1, y2, H, ch, y, hs);
            if (in_y < 0 || in_y > (float)(H - 1)) continue;
            const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
            const float yl = in_y - (float)top;
            for (int x = 0; x < cw; ++x) {
                const float in_x = orc_in_coord(x1, x2, W, cw, x, ws);
                if (in_x < 0 || in_x > (float)(W - 1)) continue;
                const int lft = (int)floorf(in_x), rgt = (int)ceilf(in_x);
                const float xl = in_x - (float)lft;
                for (int d = 0; d < C; ++d) {
                    float *p = grads_image + b_in * img_elems + d * plane;
                    const float g = grads[crop_elems * b + cplane * d + y * cw + x];
                    const float dtop = (1 - yl) * g;
                    p[(int64_t)top * W + lft] += (1 - xl) * dtop;
                    p[(int64_t)top * W + rgt] += xl * dtop;
                    const float dbot = yl * g;
                    p[(int64_t)bot * W + lft] += (1 - xl) * dbot;
                    p[(int64_t)bot * W + rgt] += xl * dbot;
                }
            }
        }
    }
    return 0;
}

/* Tap indices only (for "RoI index outputs bit-identical" tests): for each
 * (box, y, x) writes top*W+left as int32, or -1 when extrapolated. */
int orc_crop_and_resize_taps(int H, int W, const float *boxes, int K, int ch, int cw,
                             int32_t *taps) {
    for (int b = 0; b < K; ++b) {
        const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
        const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
        const float ws = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
        for (int y = 0; y < ch; ++y) {
            const float in_y = orc_in_coord(y1, y2, H, ch, y, hs);
            const int ybad = (in_y < 0 || in_y > (float)(H - 1));
            for (int x = 0; x < cw; ++x) {
                const float in_x = orc_in_coord(x1, x2, W, cw, x, ws);
                const int xbad = (in_x < 0 || in_x > (float)(W - 1));
                taps[((int64_t)b * ch + y) * cw + x] =
                    (ybad || xbad) ? -1 : (int32_t)floorf(in_y) * W + (int32_t)floorf(in_x);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Sem-dist ("layer") target decode: uint64 label -> [L, N, H, W] uint8.      */
/* Closed form of /root/reference/amodal_train.py:236-271 (load_layer2) with  */
/* /root/reference/modal/Functions.py:1012-1095:                              */
/*   low word bit i  : object i visible here      -> plane 0                  */
/*   high word bit i : object i occluded here; rank = number of lower-indexed */
/*                     occluded objects at this pixel; plane min(rank+1, L-1) */
/* followed by the [H,W,L,N] -> [L,N,H,W] axis shuffle of                     */
/* Functions.py:735 + model.py:114.  N must be orc_label_num_objects().       */
/* ------------------------------------------------------------------------ */
int orc_label_num_objects(const uint64_t *label, int64_t npix) {
    /* Functions.py:1074-1079 max_objectID: first shift at which no label's low
       word, shifted right, equals exactly 1 -- i.e. (highest set low-word bit
       over all labels) + 1, PROVIDED every lower shift also finds a label whose
       top low-word bit sits there; the loop stops at the first gap. */
    uint32_t tops = 0; /* bit s set <=> some label has its highest low bit at s */
    for (int64_t p = 0; p < npix; ++p) {
        const uint32_t vis = (uint32_t)(label[p] & 0xffffffffu);
        if (vis) tops |= 1u << (31 - __builtin_clz(vis));
    }
    int shift = 0;
    while (shift < 32 && ((tops >> shift) & 1u)) ++shift;
    return shift;
}

int orc_label_decode_u64(const uint64_t *label, int H, int W, int L, int N,
                         uint8_t *planes) {
    if (L < 1 || N < 0 || N > 32) return 1;
    const int64_t npix = (int64_t)H * W;
    memset(planes, 0, (size_t)L * (size_t)N * (size_t)npix);
    for (int64_t p = 0; p < npix; ++p) {
        const uint64_t v = label[p];
        if (!v) continue;
        const uint32_t lo = (uint32_t)(v & 0xffffffffu), hi = (uint32_t)(v >> 32);
        for (int i = 0; i < N; ++i) {
            if ((lo >> i) & 1u) planes[((int64_t)0 * N + i) * npix + p] = 1;
            if ((hi >> i) & 1u) {
                const int rank = __builtin_popcount(hi & ((1u << i) - 1u));
                const int pl = (rank + 1 >= L - 1) ? (L - 1) : (rank + 1);
                planes[((int64_t)pl * N + i) * npix + p] = 1;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Proposal front end: delta decode + clip, fp32, per box.                     */
/* /root/reference/modal/Functions.py:77-98 (apply_box_deltas) and :101-111    */
/* (clip_boxes) with the std-dev scaling of :131-135.  exp() is float expf.    */
/* ------------------------------------------------------------------------ */
int orc_box_decode_clip_f32(const float *anchors, const float *deltas, int64_t n,
                            const float *std_dev, float win_h, float win_w,
                            float *boxes) {
    for (int64_t i = 0; i < n; ++i) {
        const float *a = anchors + i * 4, *d = deltas + i * 4;
        float height = a[2] - a[0], width = a[3] - a[1];
        float cy = a[0] + 0.5f * height, cx = a[1] + 0.5f * width;
        const float dy = d[0] * std_dev[0], dx = d[1] * std_dev[1];
        const float dh = d[2] * std_dev[2], dw = d[3] * std_dev[3];
        cy = cy + dy * height;
        cx = cx + dx * width;
        height = height * expf(dh);
        width = width * expf(dw);
        float y1 = cy - 0.5f * height, x1 = cx - 0.5f * width;
        float y2 = y1 + height, x2 = x1 + width;
        y1 = fminf(fmaxf(y1, 0.0f), win_h); x1 = fminf(fmaxf(x1, 0.0f), win_w);
        y2 = fminf(fmaxf(y2, 0.0f), win_h); x2 = fminf(fmaxf(x2, 0.0f), win_w);
        boxes[i * 4 + 0] = y1; boxes[i * 4 + 1] = x1;
        boxes[i * 4 + 2] = y2; boxes[i * 4 + 3] = x2;
    }
    return 0;
}

/* IoU matrix without the +1 (Functions.py:184-218 bbox_overlaps), fp32. */
int orc_bbox_overlaps_f32(const float *b1, int64_t n1, const float *b2, int64_t n2,
                          float *iou) {
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j = 0; j < n2; ++j) {
            const float *p = b1 + i * 4, *q = b2 + j * 4;
            const float y1 = fmaxf(p[0], q[0]), x1 = fmaxf(p[1], q[1]);
            const float y2 = fminf(p[2], q[2]), x2 = fminf(p[3], q[3]);
            const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
            const float a1 = (p[2] - p[0]) * (p[3] - p[1]);
            const float a2 = (q[2] - q[0]) * (q[3] - q[1]);
            iou[i * n2 + j] = inter / ((a1 + a2) - inter);
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Inference tail.                                    */
/* ------------------------------------------------------------------------ */

/* Run-length encoding of one binary mask stored column-major (the layout
 * np.asfortranarray hands to pycocotools at amodal_train.py:397).
 * Follows /root/reference/cocoapi/common/maskApi.c:33-42 (rleEncode): runs
 * alternate starting with a run of zeros (possibly empty); a run ends wherever
 * the byte value changes.  cnts: caller-sized [a+1].  Returns the run count.
 * Pinned against oracle/_ref/libmaskapi_ref.so (the reference's own maskApi.c
 * compiled here) by tests/test_tail_cpu.py and tests/golden/rle.npz. */
int64_t orc_rle_encode_u8(const uint8_t *mask, int64_t a, uint32_t *cnts) {
    int64_t k = 0;
    uint32_t run = 0;
    uint8_t prev = 0;
    for (int64_t j = 0; j < a; ++j) {
        if (mask[j] != prev) { cnts[k++] = run; run = 0; prev = mask[j]; }
        ++run;
    }
    cnts[k++] = run;
    return k;
}

/* maskApi.c:44-49 (rleDecode) for one mask. */
int orc_rle_decode_u8(const uint32_t *cnts, int64_t m, uint8_t *mask) {
    uint8_t v = 0;
    for (int64_t j = 0; j < m; ++j) {
        for (uint32_t k = 0; k < cnts[j]; ++k) *(mask++) = v;
        v = !v;
    }
    return 0;
}

/* Compressed string of the counts: maskApi.c:204-216 (rleToString).  Counts
 * from the fourth on are stored as the difference to the count two places
 * back; each value is cut into 5-bit groups, low group first, bit 0x20 marks
 * "more groups follow", and 48 is added to land in printable ASCII.
 * s: caller-sized [6*m+1].  Returns the string length (without the NUL). */
int64_t orc_rle_to_string(const uint32_t *cnts, int64_t m, char *s) {
    int64_t p = 0;
    for (int64_t i = 0; i < m; ++i) {
        long x = (long)cnts[i];
        if (i > 2) x -= (long)cnts[i - 2];
        int more = 1;
        while (more) {
            char c = (char)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? (x != -1) : (x != 0);
            if (more) c |= 0x20;
            c += 48;
            s[p++] = c;
        }
    }
    s[p] = 0;
    return p;
}

/* maskApi.c:218-231 (rleFrString).  cnts: caller-sized [strlen(s)]. */
int64_t orc_rle_from_string(const char *s, uint32_t *cnts) {
    int64_t m = 0, p = 0;
    while (s[p]) {
        long x = 0;
        int k = 0, more = 1;
        while (more) {
            const char c = (char)(s[p] - 48);
            x |= (long)(c & 0x1f) << (5 * k);
            more = c & 0x20;
            ++p; ++k;
            if (!more && (c & 0x10)) x |= (long)(~0UL << (5 * k));
        }
        if (m > 2) x += (long)cnts[m - 2];
        cnts[m++] = (uint32_t)x;
    }
    return m;
}

/* ---- unmold_mask: /root/reference/utils.py:447-465 ------------------------
 * scipy.misc.imresize(mask, (h, w), interp='bilinear') / 255 >= 0.5, pasted
 * into a zero image at the box.  imresize is scipy <= 1.2's pilutil:
 * toimage() min-max "bytescale"s the float mask to uint8 (float32 arithmetic
 * on a float32 input: (data - cmin) * f32(255.0 / (cmax - cmin)), clip to
 * [0,255], + 0.5, truncate), PIL Image.resize(BILINEAR), fromimage().
 * Neither scipy.misc.imresize nor Pillow's source is part of /root/reference
 * (requirements: scipy, Pillow, no pins); the resize below restates Pillow's
 * published ImagingResample for 8-bit single-band images (libImaging/
 * Resample.c: precompute_coeffs, normalize_coeffs_8bpc, the horizontal pass
 * into a uint8 temporary, then the vertical pass; PRECISION_BITS = 32-8-2) and
 * is pinned against the installed Pillow (12.2) by tests/test_tail_cpu.py and
 * tests/golden/unmold.npz.
 * -------------------------------------------------------------------------- */
#define ORC_PIL_PRECISION_BITS (32 - 8 - 2)

static inline double orc_pil_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

/* Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size).
 * bounds: [out_size,2] (first tap, tap count); kk: [out_size, ksize] int32.
 * Returns ksize. */
static int orc_pil_coeffs(int in_size, int out_size, int **bounds_out, int32_t **kk_out) {
    const float in0 = 0.0f, in1 = (float)in_size;
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;   /* BILINEAR support = 1.0 */
    const int ksize = (int)ceil(support) * 2 + 1;
    int *bounds = (int *)malloc(sizeof(int) * 2 * (size_t)out_size);
    int32_t *kk = (int32_t *)malloc(sizeof(int32_t) * (size_t)out_size * ksize);
    double *k = (double *)malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = orc_pil_bilinear((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < ksize; ++x) {
            const double v = x < xmax ? k[x] : 0.0;
            kk[(size_t)xx * ksize + x] = v < 0
                ? (int32_t)(-0.5 + v * (1 << ORC_PIL_PRECISION_BITS))
                : (int32_t)(0.5 + v * (1 << ORC_PIL_PRECISION_BITS));
        }
        bounds[xx * 2 + 0] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

static inline uint8_t orc_pil_clip8(int32_t v) {
    v >>= ORC_PIL_PRECISION_BITS;     /* arithmetic shift, like Resample.c's clip8 lookup */
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* PIL Image.resize((ow, oh), BILINEAR) of an 8-bit image [ih, iw]. */
int orc_pil_resize_bilinear_u8(const uint8_t *in, int ih, int iw, int oh, int ow,
                               uint8_t *out) {
    if (oh <= 0 || ow <= 0) return 0;
    int *bh, *bv;
    int32_t *kh, *kv;
    const int ksh = orc_pil_coeffs(iw, ow, &bh, &kh);
    const int ksv = orc_pil_coeffs(ih, oh, &bv, &kv);
    uint8_t *tmp = (uint8_t *)malloc((size_t)ih * ow);
    for (int y = 0; y < ih; ++y)
        for (int xx = 0; xx < ow; ++xx) {
            int32_t ss = 1 << (ORC_PIL_PRECISION_BITS - 1);
            const int xmin = bh[xx * 2], xmax = bh[xx * 2 + 1];
            for (int x = 0; x < xmax; ++x)
                ss += in[(size_t)y * iw + x + xmin] * kh[(size_t)xx * ksh + x];
            tmp[(size_t)y * ow + xx] = orc_pil_clip8(ss);
        }
    for (int yy = 0; yy < oh; ++yy) {
        const int ymin = bv[yy * 2], ymax = bv[yy * 2 + 1];
        for (int xx = 0; xx < ow; ++xx) {
            int32_t ss = 1 << (ORC_PIL_PRECISION_BITS - 1);
            for (int y = 0; y < ymax; ++y)
                ss += tmp[(size_t)(y + ymin) * ow + xx] * kv[(size_t)yy * ksv + y];
            out[(size_t)yy * ow + xx] = orc_pil_clip8(ss);
        }
    }
    free(tmp); free(bh); free(bv); free(kh); free(kv);
    return 0;
}

/* scipy.misc.bytescale (pilutil.py) of a float32 array, defaults low=0 high=255. */
int orc_bytescale_f32(const float *data, int64_t n, uint8_t *out) {
    if (n <= 0) return 0;
    float cmin = data[0], cmax = data[0];
    for (int64_t i = 1; i < n; ++i) {
        if (data[i] < cmin) cmin = data[i];
        if (data[i] > cmax) cmax = data[i];
    }
    float cscale = cmax - cmin;
    if (cscale == 0.0f) cscale = 1.0f;
    const float scale = (float)(255.0 / (double)cscale);
    for (int64_t i = 0; i < n; ++i) {
        float v = (data[i] - cmin) * scale;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        out[i] = (uint8_t)(v + 0.5f);
    }
    return 0;
}

/* utils.py:447-465 (unmold_mask): mask [mh,mw] f32, box (y1,x1,y2,x2) in image
 * pixels -> full [H,W] uint8 (row-major).  The box must lie inside the image
 * (refine_detections clips to the window, model.py:791 scales into the image). */
int orc_unmold_mask_f32(const float *mask, int mh, int mw, int y1, int x1, int y2, int x2,
                        int H, int W, uint8_t *full) {
    memset(full, 0, (size_t)H * W);
    const int oh = y2 - y1, ow = x2 - x1;
    if (oh <= 0 || ow <= 0) return 0;
    if (y1 < 0 || x1 < 0 || y2 > H || x2 > W) return -1;
    uint8_t *byt = (uint8_t *)malloc((size_t)mh * mw);
    uint8_t *res = (uint8_t *)malloc((size_t)oh * ow);
    orc_bytescale_f32(mask, (int64_t)mh * mw, byt);
    orc_pil_resize_bilinear_u8(byt, mh, mw, oh, ow, res);
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x)
            /* (r.astype(f32) / 255.0) >= 0.5  <=>  r >= 128 */
            full[(size_t)(y1 + y) * W + x1 + x] = ((float)res[(size_t)y * ow + x] / 255.0f) >= 0.5f;
    free(byt); free(res);
    return 0;
}
