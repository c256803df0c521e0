"""Per-kernel roofline measurements at the BASELINE configuration (bs 16, 1024^2):
HIP-event timing on the launch stream, algorithmic bytes as defined in DESIGN.md
("Measurement").  Imported by bench.py; results go into the JSON line."""
import torch

PEAK_HBM_GBS = 8000.0
# memory-side float atomic adds: ~1.3 TB/s of ADDED bytes chip-wide whatever the footprint, adders per address or L2
# locality (MI355X_MICROARCH.md, "Global float atomics": they execute at the memory side, not in L2)
PEAK_ATOMIC_ADD_GBS = 1300.0
# the committed counter pass that `measured_*` replays (tools/collect_pmc.sh; the headline problem only)
PMC_ROIALIGN = "r6_f_pmc_roialign.json"


def _time(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters  # seconds per launch


def measure(dev, B=16, replay_traffic=True):
    from sln_amodal_amd import ops
    from sln_amodal_amd.modal.modals import _PyramidCrop
    g = torch.Generator(device=dev).manual_seed(7)
    out = {}
    # ---- RoIAlign: mask-head crop, 1600 rois x 256 ch x 16x16 from P2..P5 (NHWC) ----
    C, K, pool = 256, B * 100, 16
    maps = [torch.randn(B, C, s, s, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
            for s in (256, 128, 64, 32)]
    ctr = torch.rand(K, 2, device=dev, generator=g) * 0.6 + 0.2
    size = torch.exp(torch.rand(K, 2, device=dev, generator=g) * 2.5 - 3.0)
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, 1).contiguous()
    ind = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(100)
    from sln_amodal_amd.modal.modals import roi_levels
    lvl = roi_levels(boxes, (1024, 1024))
    elems = K * C * pool * pool
    t = _time(lambda: _PyramidCrop.apply(boxes, ind, lvl, pool, None, *maps))
    out["roialign_fwd"] = {"kernel": "pyr_fwd_kernel<4>", "bound": "hbm", "bytes_per_elem": 20,
                           "elems": elems, "ms": round(t * 1e3, 4),
                           "achieved": round(elems * 20 / t / 1e9, 1), "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": round(elems * 20 / t / 1e9 / PEAK_HBM_GBS, 4)}
    ms = [m.clone().requires_grad_(True) for m in maps]
    o = _PyramidCrop.apply(boxes, ind, lvl, pool, None, *ms)
    up = torch.randn_like(o)
    # the op as the train step runs it: one-launch zero fill of the four maps + atomic scatter
    t = _time(lambda: torch.autograd.grad(o, ms, up, retain_graph=True))
    from sln_amodal_amd.modal import modals as _modals
    was = _modals.GATHER_BACKWARD
    _modals.GATHER_BACKWARD = True
    try:          # the deterministic option (SLN_CROP_GATHER=1): write-once gather, no fill, no atomics
        t_gather = _time(lambda: torch.autograd.grad(o, ms, up, retain_graph=True))
    finally:
        _modals.GATHER_BACKWARD = was
    # its two launches timed alone: the scatter in accumulate mode (how the second crop of a train step runs)
    # and the zero fill (the same entry point with no rois)
    import ctypes as C
    from sln_amodal_amd import _lib
    grads = [torch.empty_like(m) for m in maps]
    ptrs = (C.c_void_p * 4)(*[t_.data_ptr() for t_ in grads])
    hw = (C.c_int * 8)(*[d for m in maps for d in (m.shape[2], m.shape[3])])
    upc = up.contiguous(memory_format=torch.channels_last)
    C_ = maps[0].shape[1]

    def launch(k, accumulate):
        _lib.check(_lib.lib().sln_pyramid_crop_bwd_f32(ops._ptr(upc), C_, 0, ops._ptr(boxes), ops._ptr(ind),
                                                       ops._ptr(lvl), k, pool, pool, B, C_, ptrs, hw, accumulate,
                                                       ops._stream()), "sln_pyramid_crop_bwd_f32")
    tk = _time(lambda: launch(K, 1))
    tz = _time(lambda: launch(0, 0))
    map_bytes = sum(g_.numel() * 4 for g_ in grads)
    # the classifier's 7x7 crops through the same kernel (the committed counter pass averages over both kinds)
    up7 = torch.randn(K, C_, 7, 7, device=dev, generator=g).contiguous(memory_format=torch.channels_last)

    def launch7():
        _lib.check(_lib.lib().sln_pyramid_crop_bwd_f32(ops._ptr(up7), C_, 0, ops._ptr(boxes), ops._ptr(ind),
                                                       ops._ptr(lvl), K, 7, 7, B, C_, ptrs, hw, 1,
                                                       ops._stream()), "sln_pyramid_crop_bwd_f32")
    tk7 = _time(launch7)
    # HBM bytes the scatter kernel really moved (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, committed counter pass; mean
    # over its pool-16 and pool-7 launches) over the mean live duration of the same two launches.  Replayed ONLY for
    # the problem the pass was collected on (replay_traffic): every other line carries the keys as null.
    measured = {"measured_hbm_bytes_per_launch": None, "measured_hbm_gbs": None, "measured_hbm_frac": None,
                "measured_replayed_from": None, "scatter_pool7_ms": round(tk7 * 1e3, 4),
                "atomic_added_bytes_per_launch": None, "atomic_add_gbs": None, "atomic_add_guide_gbs": PEAK_ATOMIC_ADD_GBS,
                "atomic_add_rate_vs_guide": None}
    if replay_traffic:
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        rel = os.path.join("profiles", PMC_ROIALIGN)
        try:
            pk = json.load(open(os.path.join(root, rel)))["pyr_bwd_patch_kernel"]
            by_meas = pk["fetch_bytes_per_launch_x2"] + pk["write_bytes_per_launch"]
            gbs = by_meas / (0.5 * (tk + tk7)) / 1e9
            measured.update({"measured_hbm_bytes_per_launch": by_meas, "measured_hbm_gbs": round(gbs, 1),
                             "measured_hbm_frac": round(gbs / PEAK_HBM_GBS, 4), "measured_replayed_from": rel})
            # what this kernel is really bound by (round 6): its stores ARE its atomic adds, so WRITE_SIZE is the added
            # bytes, and the guide measures ~1.3 TB/s of memory-side float adds chip-wide wherever they land; this
            # kernel sustains that rate (1.1 ... 1.2 x the guide's figure: a measured rate, not a hard roof) -- the HBM
            # fraction above cannot reach 0.6 through an atomic scatter
            agbs = pk["write_bytes_per_launch"] / (0.5 * (tk + tk7)) / 1e9
            measured.update({"atomic_added_bytes_per_launch": pk["write_bytes_per_launch"],
                             "atomic_add_gbs": round(agbs, 1), "atomic_add_guide_gbs": PEAK_ATOMIC_ADD_GBS,
                             "atomic_add_rate_vs_guide": round(agbs / PEAK_ATOMIC_ADD_GBS, 4)})
        except (OSError, KeyError, ValueError):
            pass
    model_frac = round(elems * 36 / t / 1e9 / PEAK_HBM_GBS, 4)
    # `frac` LEADS WITH THE MEASURED BYTES where they exist: the 36 B / element model over-counts --
    # the atomics of neighbouring taps resolve in L2 / Infinity Cache, so the scatter kernel alone reads > 1 against
    # it -- and the honest fraction of the HBM roof is measured bytes / measured time.  The model figures stay beside
    # it under their own names.
    out["roialign_bwd"] = {"kernel": "pyr_zero_kernel + pyr_bwd_patch_kernel", "bound": "hbm",
                           "frac": measured["measured_hbm_frac"] if measured["measured_hbm_frac"] is not None
                           else model_frac,
                           "frac_basis": "measured HBM bytes of the scatter kernel (PMC pass, replayed) / its live "
                                         "duration" if measured["measured_hbm_frac"] is not None
                           else "36 B / element model over the whole op (no counter pass for this problem)",
                           "achieved": measured["measured_hbm_gbs"] if measured["measured_hbm_gbs"] is not None
                           else round(elems * 36 / t / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "bytes_per_elem_model": 36, "elems": elems, "ms": round(t * 1e3, 4),
                           "model_gbs_whole_op": round(elems * 36 / t / 1e9, 1),
                           "model_frac_whole_op": model_frac,
                           "scatter_kernel_only_ms": round(tk * 1e3, 4),
                           "scatter_kernel_only_model_frac": round(elems * 36 / tk / 1e9 / PEAK_HBM_GBS, 4),
                           "zero_fill_ms": round(tz * 1e3, 4), "zero_fill_bytes": map_bytes,
                           "zero_fill_gbs": round(map_bytes / tz / 1e9, 1),
                           "gather_form_ms": round(t_gather * 1e3, 4), **measured,
                           "note": "ms = the whole op (zero fill + scatter, two launches). 36 B / element is the "
                                   "algorithmic model (4 B load + 4 x 8 B atomic RMW); the four maps total %d MB, so "
                                   "most of it is served by L2 / Infinity Cache and a model fraction can exceed 1 "
                                   "(scatter_kernel_only_model_frac): `frac` is the measured one where a counter pass "
                                   "exists.  gather_form_ms: the deterministic write-once option "
                                   "(SLN_CROP_GATHER=1)" % (map_bytes >> 20)}
    # ---- GLM tail: resize + max over three scales + softmax + argmax, 16 x 182 x 65 x 65 ----
    lg = torch.randn(B, 65, 65, 182, device=dev, generator=g).permute(0, 3, 1, 2)
    pyr = [torch.randn(B, s, s, 182, device=dev, generator=g).permute(0, 3, 1, 2) for s in (33, 49)]
    t = _time(lambda: ops.msc_softmax_tail(lg, pyr))
    by = 4 * (lg.numel() + sum(p_.numel() for p_ in pyr) + B * 65 * 65 * 183) + 8 * B * 65 * 65
    out["glm_tail"] = {"kernel": "glm_tail_kernel<3>", "bound": "hbm", "bytes": by, "ms": round(t * 1e3, 4),
                       "achieved": round(by / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": round(by / t / 1e9 / PEAK_HBM_GBS, 4),
                       "note": "every input read once + the [probs | argmax/255] map and the labels written; 135 MB: "
                               "launch- and latency-bound at this size"}
    # ---- NMS: 16 images x 6000 boxes ----
    N = 6000
    tl = torch.rand(B, N, 2, device=dev, generator=g) * 900
    wh = torch.rand(B, N, 2, device=dev, generator=g) * 200 + 8
    sc = torch.sort(torch.rand(B, N, device=dev, generator=g), dim=1, descending=True)[0]
    dets = torch.cat([tl, (tl + wh).clamp(max=1024), sc.unsqueeze(2)], 2).contiguous()
    t = _time(lambda: ops.nms_sorted(dets, 0.7, 1000))
    out["nms"] = {"kernel": "nms_mask_kernel+nms_reduce_kernel", "bound": "latency",
                  "us_per_image": round(t * 1e6 / B, 2), "ms_per_batch": round(t * 1e3, 4),
                  "images": B, "boxes": N, "host_sync": False}
    # ---- label decode: 16 x 1024^2 uint64 -> [16,1,8,1024,1024] u8 ----
    lab = torch.randint(0, 1 << 40, (B, 1024, 1024), device=dev, generator=g, dtype=torch.int64)
    t = _time(lambda: ops.label_decode(lab, 1, 8))
    by = B * 1024 * 1024 * (8 + 8)
    out["label_decode"] = {"kernel": "label_decode_kernel", "bound": "hbm", "bytes": by,
                           "ms": round(t * 1e3, 4), "achieved": round(by / t / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(by / t / 1e9 / PEAK_HBM_GBS, 4)}
    return out
