/*
 * sln_amodal.h -- C ABI of libsln_amodal_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the native ops of SLN-Amodal's detection hot path.  Each
 * entry point replaces one symbol (or one tight group) of the reference's
 * torch.utils.ffi extensions; the replaced interface is cited per function
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / TH types.
 *   - every data pointer is a DEVICE pointer unless marked host.
 *   - every entry point takes the HIP stream to launch on (`void*` = hipStream_t,
 *     NULL = the default stream), enqueues asynchronously and never synchronises
 *     the device or copies to the host (the reference's gpu_nms copies the whole
 *     mask to the host and reduces it there, nms/src/nms_cuda.c:31-58).
 *   - the caller owns and sizes every buffer; nothing is retained after return.
 *   - return value: 0 = SLN_OK, otherwise an SLN_ERR_* code; never exit()s
 *     (the reference CPU crop exit(-1)s on a bad box index,
 *     roialign/roi_align/src/crop_and_resize.c:39-42).
 *   - thread-safe: no global mutable state, and no entry point reads the process
 *     environment: every kernel / tile choice is a pure function of the arguments.
 *     (Only a process started with SLN_DEBUG_KNOBS set -- A/B sessions and the test
 *     suite's forced tile modes -- consults SLN_* tuning variables.)
 */
#ifndef SLN_AMODAL_H
#define SLN_AMODAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLN_OK 0
#define SLN_ERR_INVALID_ARG 1
#define SLN_ERR_WORKSPACE 2
#define SLN_ERR_LAUNCH 3
#define SLN_ERR_UNSUPPORTED 4

/* memory layout of [B,C,H,W]-shaped tensors */
#define SLN_LAYOUT_NCHW 0 /* reference layout                                   */
#define SLN_LAYOUT_NHWC 1 /* channels-last: what the HIP conv stack produces    */

typedef void *sln_stream_t; /* hipStream_t */

int sln_abi_version(void);
const char *sln_error_string(int code);

/* ---------------------------------------------------------------------------
 * Greedy NMS, batched over images, reduce kept on the device.
 * Replaces: int gpu_nms(THLongTensor* keep, THLongTensor* num_out,
 *                       THCudaTensor* boxes, float thresh)
 *           (nms/src/nms_cuda.h, nms/src/nms_cuda.c:17-67,
 *            nms/src/cuda/nms_kernel.cu:26-83)
 * with the *CPU* semantics the north star pins (nms/src/nms.c:55-61):
 * suppress when inter/(area_i+area_j-inter) >= thresh, widths/heights with the
 * legacy +1, fp32 IEEE, areas = (x2-x1+1)*(y2-y1+1) (nms/pth_nms.py:16).
 *
 * dets      [B,N,5] f32 rows (y1,x1,y2,x2,score), ALREADY sorted by score,
 *           descending, per image (like gpu_nms's `boxes`).
 * n_valid   [B] int32 number of leading valid rows per image, or NULL (= N).
 * keep      [B,max_out] int64: indices into the sorted rows, visiting order;
 *           entries past num_keep[b] are set to -1.
 * num_keep  [B] int32: min(#kept, max_out).
 * workspace device scratch of sln_nms_workspace_bytes(B,N) bytes.
 * ------------------------------------------------------------------------- */
size_t sln_nms_workspace_bytes(int B, int N);
int sln_nms_f32(const float *dets, int B, int N, const int32_t *n_valid, float thresh,
                int max_out, int64_t *keep, int32_t *num_keep, void *workspace,
                size_t workspace_bytes, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * crop_and_resize (TensorFlow-style RoIAlign, one bilinear sample per bin).
 * Replaces: crop_and_resize_gpu_forward / crop_and_resize_gpu_backward
 *           (roialign/roi_align/src/crop_and_resize_gpu.h:1-16,
 *            roialign/roi_align/src/cuda/crop_and_resize_kernel.cu:10-221)
 * with the arithmetic of the CPU path
 *           (roialign/roi_align/src/crop_and_resize.c:6-252).
 *
 * image     [B,C,H,W] f32 in `layout`;  boxes [K,4] f32 normalised
 *           (y1,x1,y2,x2);  box_ind [K] int32 in [0,B).
 * crops     [K,C,ch,cw] f32 in the same `layout` (NHWC: [K,ch,cw,C]); every
 *           element is written (no pre-zeroing needed).
 * err_flag  optional device int32; bit 0 is OR-ed in when a box index is out of
 *           range (that box's crop is filled with zeros, its gradient skipped).
 * backward: grad_image [B,C,H,W] is zeroed by the callee, then accumulated with
 *           fp32 atomics (summation order is not deterministic).
 * ------------------------------------------------------------------------- */
int sln_crop_and_resize_fwd_f32(const float *image, int B, int C, int H, int W, int layout,
                                const float *boxes, const int32_t *box_ind, int K, int ch,
                                int cw, float extrapolation_value, float *crops,
                                int32_t *err_flag, sln_stream_t stream);
int sln_crop_and_resize_bwd_f32(const float *grads, const float *boxes,
                                const int32_t *box_ind, int K, int ch, int cw, int B, int C,
                                int H, int W, int layout, float *grad_image,
                                int32_t *err_flag, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Sem-dist ("layer") target decode: per-pixel uint64 occlusion label ->
 * per-instance bit planes.
 * Replaces (Python/numpy in the reference): AmodalDataset.load_layer2
 *           (amodal_train.py:236-271) + get_image_labals, max_objectID,
 *           objectID_to_masks, maskID_to_objectIDs, number_to_index,
 *           objIDs_to_sindistanceLayer (modal/Functions.py:1012-1095) + the axis
 *           shuffle of modal/Functions.py:735 and model.py:114.
 *
 * label     [B,H,W] uint64 (low word: visible object bits, high word: occluded)
 * n_obj     [B] int32 out: object count per image (max_objectID semantics).
 * planes    [B,L,N,H,W] uint8 out, fully written; objects >= n_obj[b] are zero.
 * ------------------------------------------------------------------------- */
int sln_label_num_objects_u64(const uint64_t *label, int B, int64_t npix, int32_t *n_obj,
                              sln_stream_t stream);
/* Loader front end (ABI 11): the nearest-neighbour zoom of the uint64 labels to the network size and the object
 * count of the ORIGINAL labels, on the device.
 * Replaces: utils.resize_layer (utils.py:358-362: scipy.ndimage.zoom(order=0) of the decoded planes, on a
 *           DataLoader worker) + the flip of Functions.py:713-716 for the label, and max_objectID over the
 *           un-resized label (Functions.py:1074-1079 as called by amodal_train.py:236-271).
 * src       image b's label at src + b * src_stride, src_hw[2b] rows of src_hw[2b+1] pixels (device int32 [B,2]).
 * ys, xs    [B,OH] / [B,OW] int32 device: source row / column of every output row / column (scipy's index map,
 *           host-computed once per image size; -1 = the constant fill 0; a flipped image passes xs reversed).
 * out       [B,OH,OW] uint64, fully written. */
int sln_label_zoom_u64(const uint64_t *src, int64_t src_stride, const int32_t *src_hw, const int32_t *ys,
                       const int32_t *xs, int B, int OH, int OW, uint64_t *out, sln_stream_t stream);
int sln_label_num_objects_ragged_u64(const uint64_t *label, int B, int64_t stride, const int32_t *src_hw,
                                     int32_t *n_obj, sln_stream_t stream);
int sln_label_decode_u64(const uint64_t *label, int B, int H, int W, int L, int N,
                         uint8_t *planes, sln_stream_t stream);

/* Fused mask-target generation: decode + crop_and_resize + round-half-even in
 * one pass over the label (never materialises the [L,P,H,W] float planes of
 * modal/Functions.py:329-346).
 * rois [K,4] normalised; roi_img [K] image index; roi_obj [K] object index
 * (the roi's assigned GT instance); masks [K,L,mh,mw] f32 in {0,1}. */
int sln_mask_targets_u64(const uint64_t *label, int B, int H, int W, int L, const float *rois,
                         const int32_t *roi_img, const int32_t *roi_obj, int K, int mh,
                         int mw, float *masks, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Proposal front end (modal/Functions.py:114-178, :77-111): gather the top-n
 * anchors by foreground score, scale deltas by std_dev, decode against the
 * pixel-space anchors, clip to [0,win], emit NMS-ready rows.
 * probs [B,A,2], deltas [B,A,4], anchors [A,4], order [B,n] int64 anchor indices
 * (score-descending); std_dev: 4 host floats.  dets [B,n,5] out.
 * sln_gather_rois: rois[b,r,:] = dets[b,keep[b,r],:4] / (h,w,h,w) for
 * r < num_keep[b], zero padding after (Functions.py:166-176).
 * ------------------------------------------------------------------------- */
int sln_proposal_decode_f32(const float *probs, const float *deltas, const float *anchors,
                            const int64_t *order, int B, int A, int n, const float *std_dev,
                            float win_h, float win_w, float *dets, sln_stream_t stream);
int sln_gather_rois_f32(const float *dets, const int64_t *keep, const int32_t *num_keep, int B,
                        int N, int max_out, float norm_h, float norm_w, float *rois,
                        sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * pyramid_roi_align as one launch (modal/modals.py:20-110): every roi carries
 * its FPN level (2..5, computed by the caller with modals.py:51-64's formula);
 * the roi is sampled from that level's map with crop_and_resize's arithmetic and
 * written to its own slot, so no per-level gather / cat / re-ordering is needed.
 * maps      host array of 4 device pointers, P2..P5, each [B,H_l,W_l,C] NHWC.
 * map_hw    host array of 8 ints (H2,W2,...,H5,W5).
 * box_ind   [K] image index; level [K] int32; rois with box_ind outside [0,B) or
 *           level outside [2,5] are padding: their output is zero / no gradient.
 * out       [K,ch,cw,out_cstride] NHWC rows; this op writes channels
 *           [out_coffset, out_coffset+C) of each row (fused concat,
 *           modal/modals.py:481).
 * backward  grad_maps: 4 device pointers [B,H_l,W_l,C]; accumulate = 0: zeroed by the callee first,
 *           accumulate = 1: added to what they hold (several crops of the same maps -- classifier and
 *           mask head, modal/modals.py:438, 479 -- then share one set of gradient maps).
 * ------------------------------------------------------------------------- */
int sln_pyramid_crop_fwd_f32(const float *const *maps, const int32_t *map_hw, int B, int C,
                             const float *boxes, const int32_t *box_ind, const int32_t *level,
                             int K, int ch, int cw, float extrapolation_value, float *out,
                             int out_cstride, int out_coffset, sln_stream_t stream);
int sln_pyramid_crop_bwd_f32(const float *grads, int g_cstride, int g_coffset, const float *boxes,
                             const int32_t *box_ind, const int32_t *level, int K, int ch, int cw,
                             int B, int C, float *const *grad_maps, const int32_t *map_hw, int accumulate,
                             sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Top-k selection in visiting order (proposal front end).
 * Replaces: scores.sort(descending=True) over all A anchors followed by [:6000]
 *           (modal/Functions.py:133-147).
 * scores    B rows of A floats; element (b, a) at scores[b*stride_b + a*stride_a]
 *           (strides in elements, so the foreground column of [B,A,2] is read in place).
 * order     [B,k] int64: indices of the k largest scores, score descending, ties by the lower
 *           index (== a stable descending sort; NaN sorts first, like torch).  k <= min(A, 8192).
 * Every 8192-element segment of every image is a block (three 11/11/10-bit radix-select passes through
 * per-image global histograms, a count pass, an ordered scatter of the k candidates), then one block
 * per image sorts its candidates in LDS.  workspace: sln_topk_workspace_bytes(B, A, k) bytes of device
 * scratch (histograms, per-segment counts, candidates).
 * ------------------------------------------------------------------------- */
size_t sln_topk_workspace_bytes(int B, int A, int k);
int sln_topk_order_f32(const float *scores, int B, int A, long stride_b, long stride_a, int k,
                       int64_t *order, void *workspace, size_t workspace_bytes, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Max-pooling of the two stems, NHWC fp32, C % 4 == 0.
 * Replaces: SamePad2d + nn.MaxPool2d(3, 2) (modal/modals.py:316-317) and nn.MaxPool2d(3, 2, 1, ceil_mode=True)
 *           (modal/resnet_deeplab.py stem) and their backward.
 * Window (oh, ow) covers rows oh*S - pad_top ... + K - 1 (columns alike), clipped to the map; OH / OW are the
 * caller's (floor or ceil mode).  argmax [N,OH,OW,C] uint8: winning tap kh*K + kw (torch's rule: first of
 * equal values, NaN wins).  Backward writes every element of gx [N,H,W,C] exactly once (gather; no memset).
 * ------------------------------------------------------------------------- */
int sln_maxpool_fwd_f32(const float *x, int N, int H, int W, int C, int K, int S, int pad_top, int pad_left,
                        int OH, int OW, float *y, uint8_t *argmax, sln_stream_t stream);
int sln_maxpool_bwd_f32(const float *g, const uint8_t *argmax, int N, int H, int W, int C, int K, int S,
                        int pad_top, int pad_left, int OH, int OW, float *gx, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * FPN top-down merge (modal/modals.py:243-246): out = lateral + nearest-2x(top) in one pass, and the
 * coarse input's gradient.  NHWC fp32, C % 4 == 0.
 * Replaces: F.upsample(scale_factor=2) + add (the upsampled map written and read back) and the
 *           upsample backward.
 * sln_upsample2x_add_f32  lateral, out [N,2h,2w,C]; top [N,h,w,C]
 * sln_sumpool2x2_f32      g [N,2h,2w,C] -> gtop [N,h,w,C] = (g00 + g01) + (g10 + g11)
 * ------------------------------------------------------------------------- */
/* Grouped 3x3 convolution, forward only (BASELINE.json configs[4]; reference modal/resnext.py:31-41,
 * GroupBottleneck.conv2: padding 1, stride 1 or 2, no bias): x [N,H,W,C] fp32 NHWC, w [C][C/groups][3][3] (torch
 * layout), C/groups in {4, 8, 16, 32} (SLN_ERR_UNSUPPORTED otherwise), y [N,OH,OW,C] = relu?(conv * scale[c] +
 * shift[c]) with OH = (H - 1) / stride + 1; scale / shift may be NULL.  (The reference differentiates it through
 * autograd's grouped convolution; the two entry points below are that backward.) */
int sln_grouped_conv3x3_f32(const float *x, int N, int H, int W, int C, int groups, const float *w, int stride,
                            const float *scale, const float *shift, int relu, float *y, sln_stream_t stream);
/* Its data gradient: gy [N,OH,OW,C] (w.r.t. the layer's output) -> gx [N,H,W,C].  With y_out (the forward output
 * of conv -> BN -> ReLU) the gradient is first taken through the ReLU and the BN scale: g = gy * scale[c] where
 * y_out > 0, else 0 (scale NULL = 1); y_out NULL: g = gy. */
int sln_grouped_conv3x3_dgrad_f32(const float *gy, const float *y_out, const float *scale, int N, int H, int W,
                                  int C, int groups, const float *w, int stride, float *gx, sln_stream_t stream);
/* Its weight gradient gw [C][C/groups][3][3] (the parameter's order) from x [N,H,W,C] and the same g; `workspace`
 * of sln_grouped_conv3x3_wgrad_workspace_bytes() bytes: per-range partial sums + an ordered reduce, no atomics. */
size_t sln_grouped_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int C, int groups, int stride);
int sln_grouped_conv3x3_wgrad_f32(const float *x, const float *gy, const float *y_out, const float *scale, int N,
                                  int H, int W, int C, int groups, int stride, float *gw, void *workspace,
                                  size_t workspace_bytes, sln_stream_t stream);
/* The same layer on the fp16 matrix cores (ABI 11; BASELINE.json configs[4] "fp16 MFMA"): operands are single scaled
 * fp16 parts (the parts = 1 format of the conv entry points below: h = fp16(v * s), s a device scalar), fp32
 * accumulation in v_mfma_f32_16x16x32_f16, C % 64 == 0, C/groups in {4, 8, 16, 32}.
 * sln_grouped_conv3x3_pack_weights_f16: w fp32 [C][C/groups][3][3] -> the kernels' fragment order
 *   (sln_grouped_conv3x3_packed_weight_elems() 16-bit words), flip 0 for the forward pass, 1 for the data gradient;
 *   out NULL: only the tensor's amax is recorded (first use of its scale slot).
 * sln_grouped_conv3x3_f16: mode 0 forward, x16 [N,H,W,C] -> y fp32 and / or y16 scaled fp16 [N,OH,OW,C] (+ its running
 *   amax / clamp count); mode 1 data gradient, x16 = the PREPARED gradient [N,OH,OW,C] (ReLU mask and BN scale applied:
 *   sln_conv_grad_prep_f32 with parts = 1), w_packed packed with flip = 1 -> y = gx [N,H,W,C] fp32 and / or y16 = the
 *   prepared gradient of the layer BELOW: (mask16 > 0 ? gx : 0) * post_scale[c] as a scaled fp16 part (mask16 = that
 *   layer's output part, post_scale = its BN scale; both NULL in mode 0).
 * sln_grouped_conv3x3_wgrad_f16: gw fp32 [C][C/groups][3][3] from the prepared gradient and the layer's input parts;
 *   workspace as for the fp32 entry point (per-range partial sums + ordered reduce: bit-reproducible). */
int64_t sln_grouped_conv3x3_packed_weight_elems(int C, int groups);
int sln_grouped_conv3x3_pack_weights_f16(const float *w, int C, int groups, int flip, uint16_t *out, const float *q_scale,
                                         float *q_amax, int32_t *q_saturated, sln_stream_t stream);
int sln_grouped_conv3x3_f16(const uint16_t *x16, int N, int H, int W, int C, int groups, const uint16_t *w_packed,
                            int stride, int mode, const float *scale, const float *shift, int relu, float *y,
                            uint16_t *y16, const float *x_scale, const float *w_scale, const float *y_q_scale,
                            float *y_q_amax, int32_t *y_q_saturated, const uint16_t *mask16, const float *post_scale,
                            sln_stream_t stream);
int sln_grouped_conv3x3_wgrad_f16(const uint16_t *x16, const uint16_t *gz16, int N, int H, int W, int C, int groups,
                                  int stride, const float *gz_scale, const float *x_scale, float *gw, void *workspace,
                                  size_t workspace_bytes, sln_stream_t stream);
/* Tail of the global layer module (reference model.py:537-541, modal/msc_deeplab.py:42-48), one pass: the logits
 * of the coarser scales resized bilinearly (align_corners = False) to the scale-1 grid, element-wise maximum over
 * the scales, softmax over the C classes, argmax.  All maps NHWC fp32 with the given pixel strides (floats).
 * probs [B,H,W,C+1]: the C probabilities and argmax / 255; label [B,H,W] int64: the argmax. */
int sln_msc_softmax_tail_f32(const float *logits, int64_t logits_pixel_stride, const float *const *pyramid,
                             const int32_t *pyramid_hw, const int64_t *pyramid_pixel_stride, int n_pyramid,
                             int B, int C, int H, int W, float *probs, int64_t *label, sln_stream_t stream);
int sln_upsample2x_add_f32(const float *lateral, const float *top, int N, int h, int w, int C, float *out,
                           sln_stream_t stream);
int sln_sumpool2x2_f32(const float *g, int N, int h, int w, int C, float *gtop, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Pyramid RoIAlign backward as a write-once gather (no zero fill, no atomics, bit-reproducible).
 * Replaces: crop_and_resize_gpu_backward's memset + atomicAdd scatter
 *           (roialign/roi_align/src/cuda/crop_and_resize_kernel.cu:118-215) for the crops of
 *           pyramid_roi_align (modal/modals.py:66-108), for up to 4 crop sets of the SAME four maps at once
 *           (classifier 7x7 + mask 16x16: modals.py:438, 479).
 * Host arrays of length nsrc describe the sources: grads[i] [K_i][ch_i][cw_i][g_cstride_i] with the C channels
 * at g_coffset_i, boxes / box_ind / level as in sln_pyramid_crop_fwd_f32 (box_ind < 0: padded slot), ch, cw <= 32.
 * Every element of the four grad_maps [B][H_l][W_l][C] is WRITTEN (zeros where no roi reaches): per 8x8 map tile
 * a block finds the rois of its image and level that meet the tile, inverts their bin taps per map row /
 * column and sums w_x * (w_y * g) -- the reference's products, a fixed order of additions.
 * workspace: sln_pyramid_crop_bwd_gather_workspace_bytes(sum K_i, B) bytes (per-image roi lists).
 * ------------------------------------------------------------------------- */
size_t sln_pyramid_crop_bwd_gather_workspace_bytes(int total_rois, int B);
int sln_pyramid_crop_bwd_gather_f32(int nsrc, const float *const *grads, const int32_t *g_cstride,
                                    const int32_t *g_coffset, const float *const *boxes,
                                    const int32_t *const *box_ind, const int32_t *const *level, const int32_t *K,
                                    const int32_t *ch, const int32_t *cw, int B, int C, float *const *grad_maps,
                                    const int32_t *map_hw, void *workspace, size_t workspace_bytes,
                                    sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Optimiser step: global-norm clip + momentum SGD over the whole parameter set.
 * Replaces: torch.nn.utils.clip_grad_norm(params, 5.0) followed by torch.optim.SGD.step()
 *           (model.py:441-444, optimizer built at model.py:352-358): one norm kernel per tensor,
 *           a host-side sum, then 4-5 elementwise kernels per tensor.
 * The n tensors are cut into chunks of chunk_elems elements; every table below lives in DEVICE
 * memory:  params/grads/bufs [n] pointers, numel [n], weight_decay [n] (per tensor: the reference's
 * two parameter groups), chunk_tensor [n_chunks] (tensor index), chunk_offset [n_chunks] (first
 * element).  Momentum buffers start at zero (the first step then stores d, like SGD's clone).
 *
 * sln_grad_sqnorm_f32   partial [n_chunks] float64 scratch; sqnorm [1] float64 = sum of squares of
 *                       every gradient element (float64 accumulation, fixed order: reproducible).
 * sln_sgd_clip_step_f32 c = max_norm / (sqrt(sqnorm) + 1e-6), applied when < 1 (clip_grad_norm);
 *                       d = g*c + wd*p;  buf = momentum*buf + d;  p = p - lr*buf  (dampening 0,
 *                       no nesterov).  Gradients are read, not rewritten.  A non-finite sqnorm (an inf /
 *                       NaN gradient) skips the update of every tensor and adds 1 to *skipped (device
 *                       int32, may be NULL): the reference's loop `continue`s past a batch it cannot use
 *                       (model.py:416-418, 433-434); here no host round trip is needed to decide it.
 * ------------------------------------------------------------------------- */
int sln_grad_sqnorm_f32(const float *const *grads, const int64_t *numel, const int32_t *chunk_tensor,
                        const int64_t *chunk_offset, int n_chunks, int chunk_elems, double *partial,
                        double *sqnorm, sln_stream_t stream);
int sln_sgd_clip_step_f32(float *const *params, const float *const *grads, float *const *bufs,
                          const int64_t *numel, const float *weight_decay, const int32_t *chunk_tensor,
                          const int64_t *chunk_offset, int n_chunks, int chunk_elems, const double *sqnorm,
                          float max_norm, float lr, float momentum, int32_t *skipped, sln_stream_t stream);

/* ---------------------------------------------------------------------------
 * Inference tail: full-size masks and their COCO run-length encoding.
 *
 * sln_unmold_masks_u8
 * Replaces: the per-detection host loop utils.unmold_mask (utils.py:447-465) called from
 *           MaskRCNN.unmold_detections (model.py:796-803): scipy.misc.imresize(mask, (y2-y1, x2-x1),
 *           interp='bilinear') / 255 >= 0.5, pasted into a zero image at the box.
 * masks     [N,C,mh,mw] float32 head outputs (mh, mw <= 64); class_ids [N] int32 picks the plane of
 *           each detection (NULL: plane 0).
 * boxes     [N,4] int32 (y1,x1,y2,x2) in image pixels; an empty or out-of-image box gives a zero
 *           mask (the reference drops such detections beforehand, model.py:786-793).
 * full      [N,W,H] uint8 in {0,1}, COLUMN-major per mask (y fastest): the np.asfortranarray layout
 *           the reference hands to pycocotools (amodal_train.py:397); fully written, no memset needed.
 * Bit-identical to scipy<=1.2 bytescale (float32) + Pillow's 8-bit BILINEAR resample.
 *
 * sln_rle_encode_u8
 * Replaces: rleEncode (cocoapi/common/maskApi.c:33-42) behind pycocotools.mask.encode.
 * masks     [N,a] uint8, a = h*w bytes per mask in column-major order.
 * counts    [N,max_runs] uint32 run lengths, starting with a (possibly empty) run of zeros;
 * num_runs  [N] int32, always the true run count.  If num_runs[n] > max_runs, row n of counts is
 *           unspecified: call again with a larger capacity (a + 1 always suffices).
 *
 * sln_rle_to_string / sln_rle_from_string  (host memory, no device work)
 * Replace:  rleToString / rleFrString (maskApi.c:204-231), the "counts" bytes of a COCO RLE dict.
 *           Return the length written (>= 0), or -SLN_ERR_INVALID_ARG when cap is too small (6 characters per
 *           count, resp. one count per character, always suffice).
 * ------------------------------------------------------------------------- */
int sln_unmold_masks_u8(const float *masks, const int32_t *class_ids, const int32_t *boxes, int N, int C,
                        int mh, int mw, int H, int W, uint8_t *full, sln_stream_t stream);
int sln_rle_encode_u8(const uint8_t *masks, int N, int64_t a, int max_runs, uint32_t *counts,
                      int32_t *num_runs, sln_stream_t stream);
int64_t sln_rle_to_string(const uint32_t *counts, int64_t m, char *out, int64_t cap);
int64_t sln_rle_from_string(const char *s, int64_t len, uint32_t *counts, int64_t cap);
/* sln_rle_to_strings: rleToString (maskApi.c:204-216) for the N masks of a batch in one call (ABI 12): row n of
 * counts (row pitch row_stride words) holds num_runs[n] counts; strings back to back in out, string n =
 * out[offsets[n] .. offsets[n + 1]) (offsets: N + 1 entries).  Returns the total length or -SLN_ERR_INVALID_ARG. */
int64_t sln_rle_to_strings(const uint32_t *counts, int64_t row_stride, const int32_t *num_runs, int N, char *out,
                           int64_t cap, int64_t *offsets);

/* ---------------------------------------------------------------------------
 * Convolution stacks (modal/modals.py:203-499 backbone / FPN / RPN / heads,
 * modal/resnet_deeplab.py + modal/deeplabv2.py GLM): implicit-GEMM on the bf16
 * matrix cores with split-bf16 operands (fp32-class accuracy, see csrc/conv.hip).
 * Replaces the cuDNN calls behind nn.Conv2d (+ the separate BatchNorm / ReLU /
 * residual-add / F.pad passes around them) on the reference's path.
 * All activations are NHWC = [pixels][channels]; `parts` = 3 bf16 or 2 scaled fp16 parts per
 * fp32 value (see "Operand formats" below); "*_pad" channel counts are multiples of 8 (zero filled).
 *
 * sln_conv_split_weights_f32  fp32 weights with element strides (s_o,s_i,s_kh,s_kw)
 *     -> out [parts][O][KH][KW][I_pad] bf16.  flip=1 mirrors the taps (data grad).
 * sln_act_split_f32           x [M][C] fp32 -> out [parts][M][C_pad] bf16.
 * sln_im2col_split_f32        patch matrix of a small-Cin convolution as operand parts: x
 *     [N,H,W,C] fp32 -> rows [row0, row0 + N*OH*OW) of out [parts][out_rows][K_pad], column
 *     k = (kh, kw, c), K = KH*KW*C zero-padded to K_pad (% 8 == 0); taps outside the image are zero.
 *     The 3-channel 7x7/2 stems (modal/modals.py:311, modal/resnet_deeplab.py conv1) then run as a
 *     1x1 convolution over K_pad channels on sln_conv2d_fwd_f32 / sln_conv2d_wgrad_f32 (weights
 *     reordered to [Cout][KH][KW][C]); replaces the cuDNN 7x7 calls.  out == NULL: amax only.
 * sln_col2im_f32              adjoint of the patch matrix: cols [N*OH*OW][K_pad] fp32 (a 1x1 data
 *     gradient over the K_pad channels) -> gx [N,H,W,C]; only needed when the image carries a gradient.
 * sln_conv_grad_prep_f32      gz = gy * (y > 0) * scale[c] (y, scale optional):
 *     writes gu = gy*(y>0) fp32 [M][C] (optional), gz parts [parts][M][C_pad] and
 *     the per-channel sum of gz into gbias [C] (optional, zeroed by the callee).
 * sln_conv2d_fwd_f32          y [N,OH,OW,Cout] fp32 = relu?(conv(x,w)*scale[c] +
 *     shift[c] + residual); x_parts [parts][N*H*W][Cin] (Cin % 8 == 0), w_parts
 *     [parts][Cout][KH][KW][Cin].  Taps outside the image read zero (pad_top/left
 *     may differ from bottom/right: TensorFlow 'SAME' padding).  y_parts (optional):
 *     the output's own parts [parts][N*OH*OW][Cout_pad], written by the epilogue
 *     (fused sln_act_split_f32 for the next layer); pad channels are NOT written.
 * sln_conv2d_fwd_ms_f32       the same convolution over up to 4 image groups of different
 *     sizes in ONE launch: group q = seg_nhw[3q..3q+2] = (N, H, W); inputs and outputs
 *     are the groups' [pixels][channels] rows back to back.  Replaces the per-scale
 *     Python loop of the multi-scale GLM (modal/msc_deeplab.py:29-37 runs the same
 *     network at scales 1, 0.5, 0.75 one after the other): 65^2 / 49^2 / 33^2 maps
 *     alone fill 1.4 / 0.8 / 0.4 rounds of the chip's resident tiles, together 2.5.
 *     Results are bit-identical to per-group sln_conv2d_fwd_f32 calls.
 *     Epilogue extras used by the backward pass, where this kernel computes a data
 *     gradient whose only reader is the previous layer's gradient preparation
 *     (sln_conv_grad_prep_f32): mask [M][Cout] (optional) zeroes every output element
 *     whose mask value is not > 0 (the previous layer's ReLU, after scale/shift/
 *     residual/relu); y may be NULL when only y_parts are wanted; colsum [Cout]
 *     (optional, zeroed by the callee) receives the per-channel sums of the output
 *     (the previous layer's bias gradient; fp32 atomics).  post_scale [Cout]
 *     (optional): y is written as computed, but y_parts and colsum hold y*post_scale[c]
 *     (product rounded to fp32 first) -- the previous layer's frozen-BN scale when the
 *     fp32 gradient itself is still needed (it is that layer's shortcut gradient).
 *     Parts-only operands (parts = 2, Cout % 8 == 0): the outputs of a bottleneck's convolutions
 *     (modal/modals.py:264-301, modal/resnet_deeplab.py:26-71) are read by convolutions, by the next
 *     shortcut and as ReLU masks only, so their fp32 copy is never written (y = NULL, y_parts set:
 *     4 B per element instead of 8) and their other two readers take the parts instead:
 *     residual_parts [2][M][Cout_pad] + residual_scale (device scalar): the shortcut, added as
 *     (h0 + h1) / s -- the value the next convolution reads of the same tensor (exclusive with
 *     `residual`); mask_part0 [M][Cout_pad]: part 0 of the tensor whose ReLU pattern masks the output
 *     (h0 > 0; exclusive with `mask`).  sln_conv_grad_prep_f32 takes the pattern the same way
 *     (y_part0 instead of y).  SLN_ERR_UNSUPPORTED outside the fp16 x 2 fixed-feature epilogue.
 * sln_conv2d_wgrad_f32        gw [Cout][KH][KW][Cin] fp32 (zeroed by the callee) =
 *     sum over output pixels of gz[pix][co] * x[pix @ tap][ci]; split-K over pixel
 *     ranges, summed in range order through a caller-lent workspace (or with fp32
 *     atomics when none is given).
 *
 * Operand formats.  parts = 3: three bf16 parts per fp32 value, six part products per
 * fp32 product.  parts = 2 ("scaled split-fp16"): two fp16 parts of v*s, s a per-tensor
 * power of two, three part products (22-bit operands: the accuracy of fp32 at half the
 * matrix work).  parts = 1 (ABI 11; BASELINE.json configs[4] "fp16 MFMA"): ONE scaled fp16 part h = fp16(v*s) --
 * fp16 storage, one product per multiply-add, fp32 accumulate, fp16-class results; generic 128-wide kernels only,
 * weights in SLN_WEIGHTS_ROWS.  Every function that WRITES parts takes the tensor's scaling record
 *     q_scale      device scalar s (NULL = 1): the parts encode v*s
 *     q_amax       device scalar (optional): atomic running max |v| of what was split --
 *                  the input of sln_scale_update_f32, which derives the scale the same
 *                  tensor will use the next time it is produced (delayed scaling)
 *     q_saturated  device counter (optional): += number of blocks that had to clamp an
 *                  element to +-65504 (never inf: a stale scale costs accuracy, not NaNs)
 * and with a NULL output pointer only takes the amax (first use of a tensor, before it has
 * a scale); every function that READS parts takes its operands' scales (device scalars,
 * NULL = 1) and divides them out of the accumulator -- exactly, powers of two.
 * parts = 3 ignores all of these (pass NULL).
 * sln_scale_update_f32        history [window][history_stride] (optional) is a ring of the last `window` maxima
 *     of every tensor, cursor [n] its write positions: amax[i] is stored, a = max over the ring (a = amax[i]
 *     without a history); scale[i] <- 2^k with a*2^k in [2^(t-1), 2^t), t = target_log2 (11 on the path);
 *     amax[i] == 0 (tensor not produced since the last call) changes nothing; amax[i] <- 0.
 * sln_scale_update_headroom_f32   the same with headroom[i] (0 ... 8, NULL = 0) extra bits of head room per tensor:
 *     t = target_log2 - headroom[i].  The path gives its GRADIENT tensors 3 (2^8 below fp16's 65504 instead of 2^5).
 * ------------------------------------------------------------------------- */
/* Weight-part layouts.  ROWS: [parts][O][KH][KW][I_pad] (the 128x128 forward kernel).  TILED256: the
 * LDS image of the 256x256 forward kernel -- for each 256-row Cout tile, each 16-channel K stage (in
 * the kernel's own K order) and each part one contiguous 8-KB block, so that its 1-KiB DMA pieces
 * read whole consecutive 128-B lines (sln_conv_tiled_weight_elems() elements per part, zero padded).
 * (TILED256H: the same for the fp16 x 2 kernel's 32-channel stages.)  A forward call must pass the
 * layout its kernel uses: sln_conv_fwd_weights_layout(). */
/* Column sums (colsum of sln_conv2d_fwd_ms_f32, gbias of sln_conv_grad_prep_f32) are accumulated with atomics into a
 * buffer the callee zeroes first -- one fill launch per layer and step.  A caller that hands over memory it has
 * already cleared (slices of an arena zeroed once per step) ORs this flag into `relu` / `parts` to skip the fill. */
#define SLN_SUMS_PREZEROED 0x100
#define SLN_WEIGHTS_ROWS 0
#define SLN_WEIGHTS_TILED256 1
#define SLN_WEIGHTS_TILED256H 2 /* parts = 2 only: 32-channel stages, 16-KB blocks */
int64_t sln_conv_tiled_weight_elems(int O, int I, int KH, int KW, int layout);
/* Debug sessions only (SLN_DEBUG_KNOBS + SLN_CONV_STAMP): per-wave cycle sums of the stamped
 * diagnostic build of the fp16 forward kernel, [8 waves][4 phases][4 segments] -> host. */
int sln_debug_read_stamps(uint64_t *out128);
/* The layout a forward call must pass: M output pixels, Cout, Cin (padded, as in x_parts), taps =
 * KH*KW, parts, and x_pixels = the number of input pixels (rows of x_parts). */
int sln_conv_fwd_weights_layout(int64_t M, int Cout, int Cin, int taps, int parts, int64_t x_pixels);
/* One entry of sln_conv_split_weights_batch_f32's device-resident table: the arguments of one
 * sln_conv_split_weights_f32 call with parts = 2 (total = elements of one part in the chosen layout:
 * O*KH*KW*I_pad for ROWS, sln_conv_tiled_weight_elems() otherwise). */
typedef struct {
    const float *w;
    uint16_t *out;
    const float *q_scale;
    float *q_amax;
    int32_t *q_saturated;
    int64_t s_o, s_i, s_kh, s_kw, total;
    int32_t O, I, Ip, KH, KW, flip, layout, reserved;   /* reserved: 1 = ONE scaled fp16 part (ABI 11), else two */
} sln_split_desc_t;
/* Every stale weight tensor of a step in one launch (parts = 2, or 1 per entry): chunk b covers elements
 * [chunk_first[b], chunk_first[b] + chunk_elems) of entry chunk_entry[b]; all three arrays in device memory. */
int sln_conv_split_weights_batch_f32(const sln_split_desc_t *descs, const int32_t *chunk_entry,
                                     const int64_t *chunk_first, int n_chunks, int chunk_elems,
                                     sln_stream_t stream);
int sln_conv_split_weights_f32(const float *w, int O, int I, int I_pad, int KH, int KW, long s_o,
                               long s_i, long s_kh, long s_kw, int flip, int parts, int layout,
                               uint16_t *out, const float *q_scale, float *q_amax,
                               int32_t *q_saturated, sln_stream_t stream);
int sln_im2col_split_f32(const float *x, int N, int H, int W, int C, int KH, int KW, int stride_h,
                         int stride_w, int pad_top, int pad_left, int OH, int OW, int K_pad, int parts,
                         uint16_t *out, int64_t out_rows, int64_t row0, const float *q_scale, float *q_amax,
                         int32_t *q_saturated, sln_stream_t stream);
int sln_col2im_f32(const float *cols, int N, int H, int W, int C, int KH, int KW, int stride_h, int stride_w,
                   int pad_top, int pad_left, int OH, int OW, int K_pad, float *gx, sln_stream_t stream);
int sln_act_split_f32(const float *x, int64_t M, int C, int C_pad, int parts, uint16_t *out,
                      const float *q_scale, float *q_amax, int32_t *q_saturated, sln_stream_t stream);
int sln_conv_grad_prep_f32(const float *gy, const float *y, const uint16_t *y_part0, const float *scale,
                           int64_t M, int C, int C_pad, int parts, float *gu, uint16_t *gz_parts, float *gbias,
                           const float *q_scale, float *q_amax, int32_t *q_saturated,
                           sln_stream_t stream);
/* The same preparation for a layer whose output went through a max-pool (sln_maxpool_fwd_f32's geometry and
 * winning taps): the layer's gradient is gathered from the POOLED gradient g_pool [N,OH,OW,C] on the fly -- no
 * pool-backward launch and no fp32 gradient map of the layer's [N,H,W,C] output (the backbone stem: modals.py:311-317).
 * y: the layer's fp32 output (ReLU pattern) or NULL; C % 8 == 0; gz_parts [parts][N*H*W][C]; parts may carry
 * SLN_SUMS_PREZEROED; gz_parts NULL with parts 2: amax-only pass. */
int sln_conv_grad_prep_pooled_f32(const float *g_pool, const uint8_t *argmax, int N, int H, int W, int K, int S,
                                  int pad_top, int pad_left, int OH, int OW, const float *y, const float *scale,
                                  int C, int parts, uint16_t *gz_parts, float *gbias, const float *q_scale,
                                  float *q_amax, int32_t *q_saturated, sln_stream_t stream);
int sln_scale_update_f32(float *amax, float *scale, float *history, int32_t *cursor, int n,
                         int64_t history_stride, int window, int target_log2, sln_stream_t stream);
int sln_scale_update_headroom_f32(float *amax, float *scale, float *history, int32_t *cursor, const int8_t *headroom,
                                  int n, int64_t history_stride, int window, int target_log2, sln_stream_t stream);
int sln_conv2d_fwd_f32(const uint16_t *x_parts, int N, int H, int W, int Cin,
                       const uint16_t *w_parts, int w_layout, int parts, int Cout, int KH, int KW, int stride_h,
                       int stride_w, int dil_h, int dil_w, int pad_top, int pad_left, int OH, int OW,
                       const float *scale, const float *shift, const float *residual, int relu,
                       float *y, uint16_t *y_parts, const float *x_scale, const float *w_scale,
                       const float *y_q_scale, float *y_q_amax, int32_t *y_q_saturated,
                       sln_stream_t stream);
int sln_conv2d_fwd_ms_f32(const uint16_t *x_parts, int nseg, const int32_t *seg_nhw, int Cin,
                          const uint16_t *w_parts, int w_layout, int parts, int Cout, int KH, int KW, int stride_h,
                          int stride_w, int dil_h, int dil_w, int pad_top, int pad_left, int pad_bottom,
                          int pad_right, const float *scale, const float *shift, const float *residual,
                          int relu, const float *mask, const float *post_scale, float *y,
                          uint16_t *y_parts, float *colsum, const float *x_scale,
                          const float *w_scale, const float *y_q_scale, float *y_q_amax,
                          int32_t *y_q_saturated, const uint16_t *residual_parts, const float *residual_scale,
                          const uint16_t *mask_part0, sln_stream_t stream);
/* Tile edge (128 or 256) of the forward kernel the two functions above use for M output pixels,
 * Cout channels, K = KH*KW*Cin and `parts`: host-side rule, no GPU work (csrc/conv.hip). */
int sln_conv_fwd_tile(int64_t M, int Cout, int64_t K, int parts);
/* Which forward kernel the calling thread's last sln_conv2d_fwd*_f32 call launched: 0 conv_fwd_kernel (128 x 128),
 * 1 conv_fwd256_kernel, 2 conv_fwd256h_kernel, 3 conv_fwd128x256h_kernel (pointwise layers: two blocks per CU),
 * 4 conv_fwd256h_kernel's tap-row instances (3-wide kernels, stride 1, maps of 32 ... 256 columns),
 * 5 conv_fwd_kernel's ROW3 instances (3-wide kernels, stride 1, rows of whole 128-pixel tiles: k-steps per kernel row).
 * Profiling labels only. */
int sln_conv_fwd_last_kernel(void);
/* The same for the PROCESS's last sln_conv2d_wgrad_f32 call (autograd launches weight gradients from its own thread):
 * 0 conv_wgrad_kernel (a block per tap), 1 the 256 x 256 kernels, 2 conv_wgrad_kernel's
 * ROW3 instances (3-wide kernels, stride 1, rows of whole 32-pixel k-steps: a block per kernel row). */
int sln_conv_wgrad_last_kernel(void);
/* Tile edge (128 or 256) of the weight-gradient kernel sln_conv2d_wgrad_f32 uses (host-side rule). */
int sln_conv_wgrad_tile(int64_t M, int Cout, int Cin, int taps, int parts);
/* workspace (optional): sln_conv_wgrad_workspace_bytes() bytes lent by the caller make the split-K
 * sum two-phase -- every pixel range stores its partial gradient, a second kernel adds the ranges in
 * order: bit-reproducible, no atomics.  NULL: fp32 atomics (summation order not deterministic).
 * gw_layout 0: gw [Cout][KH][KW][Cin]; 1: gw [Cout][Cin][KH][KW], the parameter's own order (written by the
 * reduce pass: needs the workspace), so that no layout copy stands between the kernel and the optimiser. */
size_t sln_conv_wgrad_workspace_bytes(int64_t M, int Cout, int Cin, int taps, int parts);
/* gw_layout | SLN_WGRAD_DEFER_REDUCE: only the partial sums are written (workspace required, gw may be NULL);
 * the caller adds them later with sln_wgrad_reduce_batch_f32 -- the reduce passes of up to
 * SLN_WGRAD_REDUCE_BATCH layers in one launch.  An entry: partial = the layer's workspace, gw = its gradient,
 * n = Cout*KH*KW*Cin, ksplit = sln_conv_wgrad_ksplit() of the same problem, taps = KH*KW if the gradient is wanted
 * in the parameter's own order (gw_layout 1) and KH*KW > 1, else 0.  `descs` is HOST memory (copied into the
 * launch).  Same summation tree as the per-layer reduce: the same bits. */
#define SLN_WGRAD_DEFER_REDUCE 2
#define SLN_WGRAD_REDUCE_BATCH 16
typedef struct {
    const float *partial;
    float *gw;
    int64_t n;
    int32_t ksplit, taps, Cin, reserved;
} sln_wgrad_reduce_desc_t;
int sln_conv_wgrad_ksplit(int64_t M, int Cout, int Cin, int taps, int parts);
int sln_wgrad_reduce_batch_f32(const sln_wgrad_reduce_desc_t *descs, int n, sln_stream_t stream);
int sln_conv2d_wgrad_f32(const uint16_t *gz_parts, int Cout, int Cout_pad, const uint16_t *x_parts,
                         int N, int H, int W, int Cin, int Cin_pad, int parts, int KH, int KW,
                         int stride_h, int stride_w, int dil_h, int dil_w, int pad_top, int pad_left,
                         int OH, int OW, float *gw, const float *gz_scale, const float *x_scale,
                         void *workspace, size_t workspace_bytes, int gw_layout, sln_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SLN_AMODAL_H */
