"""Training input pipeline: what feeds `MaskRCNN.train_epoch` from files at the rate the GPU consumes batches.

The reference feeds its loop from `torch.utils.data.DataLoader(train_set, batch_size=1, shuffle=True, num_workers=4)`
(model.py:340-342) over `Dataset.__getitem__` (model.py:76-116) -> `load_image_gt` (Functions.py:675-736): per image a
worker reads the compressed uint64 label (`np.load(...)['layer']`, amodal_train.py:238), decodes the JPEG, squashes it
to IMAGE_MAX_DIM^2 (scipy.misc.imresize = Pillow BILINEAR), zooms the label planes (scipy.ndimage.zoom, order 0),
counts the objects and builds boxes / RPN targets in numpy.

Here the per-image work is split by where it is cheap:
  * worker PROCESSES (spawned interpreters that never touch the GPU): file read + inflate of the label, JPEG decode,
    Pillow resize -- the three steps that need a host core (~40 ms per 1024^2 image) -- written straight into a ring of
    shared-memory slots;
  * the DEVICE: nearest zoom of the label (csrc/label_decode.hip `label_zoom_kernel`: scipy's index maps, a flip is
    the reversed column map), object count of the original label, molding, tight boxes + jitter, RPN targets
    (`AmodalDataset._assemble`);
  * a FEEDER thread in the training process: collects a batch's slots, packs them into pinned staging buffers,
    issues the host->device copies on a copy stream (`non_blocking`), hands the batch over through a bounded queue
    of `depth` batches.  The training thread only waits on an event.

Order: `EpochSampler` -- one seeded permutation of the file list per epoch, the same on every rank, cut into global
batches of world x B images of which rank r takes images [r B, (r + 1) B): disjoint across ranks, every rank runs the
same number of steps (a tail that does not fill a global batch is dropped, like DistributedSampler(drop_last)).
Flips are drawn by the sampler's own generator (the reference: `random.randint` in the worker, Functions.py:713).
"""
import os
import queue
import threading
import time

import numpy as np

META = 8            # int32 words per slot: H0, W0, zoomed on the host (0 / 1), image id, flip, error flag


class EpochSampler(object):
    """Per-epoch seeded shuffle, sharded by rank."""

    def __init__(self, n, batch, rank=0, world=1, seed=0, shuffle=True, flip=True):
        if n < 1 or batch < 1 or world < 1 or not (0 <= rank < world):
            raise ValueError("EpochSampler: n %d batch %d rank %d world %d" % (n, batch, rank, world))
        self.n, self.batch, self.rank, self.world, self.seed, self.shuffle = n, batch, rank, world, seed, shuffle
        self.flip = flip                 # draw the horizontal-flip augmentation (Functions.py:713); False: never
        g = batch * world
        # fewer files than one global batch: the permutation is repeated until one is filled (tiny datasets, tests)
        self.steps = max(1, n // g)

    def order(self, epoch):
        """The epoch's visiting order of ALL ranks: a permutation of range(n) (tiled if n < one global batch)."""
        rs = np.random.RandomState((self.seed * 1000003 + epoch) % (2 ** 32))
        perm = rs.permutation(self.n) if self.shuffle else np.arange(self.n)
        need = self.steps * self.batch * self.world
        if need > self.n:
            perm = np.concatenate([perm] * (need // self.n + 1))
        return perm[:need]

    def epoch(self, epoch):
        """-> [(image ids [B], flips [B] of 0 / 1)] of this rank, one entry per step."""
        perm = self.order(epoch)
        rs = np.random.RandomState((self.seed * 1000003 + epoch + 7919) % (2 ** 32))
        flips = rs.randint(0, 2, size=perm.shape[0]) if self.flip else np.zeros(perm.shape[0], np.int64)
        g = self.batch * self.world
        out = []
        for s in range(self.steps):
            lo = s * g + self.rank * self.batch
            out.append((perm[lo:lo + self.batch].copy(), flips[lo:lo + self.batch].copy()))
        return out


# ---------------------------------------------------------------------------------------------- worker side
def load_item(info, dim):
    """What a worker does for one image (no torch, no GPU): -> (u8 [dim, dim, 3], label uint64 [H0, W0]).
    The image is squashed to dim x dim like utils.resize_image (utils.py:351-356); a missing image file gives the
    zeros load_image returns."""
    from PIL import Image
    layer = np.load(info["label"])["layer"].astype(np.uint64, copy=False)
    path = info["path"]
    if os.path.exists(path):
        im = Image.open(path).convert("RGB")
        if im.size != (dim, dim):
            im = im.resize((dim, dim), Image.BILINEAR)
        u8 = np.asarray(im)
    else:
        u8 = np.zeros((dim, dim, 3), np.uint8)
    return u8, layer


def _zoom_index(n_in, n_out):
    """utils.zoom_nearest_index without importing torch into a worker (kept identical: tests compare them)."""
    if n_out <= 0:
        return np.zeros(0, dtype=np.int64)
    if n_out == 1 or n_in <= 1:
        return np.zeros(n_out, dtype=np.int64)
    step = np.float64(n_in - 1) / np.float64(n_out - 1)
    cc = np.arange(n_out, dtype=np.float64) * step
    idx = np.floor(cc + 0.5).astype(np.int64)
    return np.where(cc > np.float64(n_in - 1), -1, np.clip(idx, 0, n_in - 1))


def _attach(names, nslots, dim, cap):
    from multiprocessing import shared_memory
    shms = [shared_memory.SharedMemory(name=n) for n in names]
    imgs = np.ndarray((nslots, dim, dim, 3), np.uint8, buffer=shms[0].buf)
    labs = np.ndarray((nslots, cap), np.uint64, buffer=shms[1].buf)
    meta = np.ndarray((nslots, META), np.int32, buffer=shms[2].buf)
    return shms, imgs, labs, meta


def _worker_main(names, nslots, dim, cap, infos, task_q, done_q):
    """Worker process: (slot, image id, flip) -> the slot's image / label / meta filled -> (slot, error or None)."""
    try:
        os.environ["OMP_NUM_THREADS"] = "1"
        shms, imgs, labs, meta = _attach(names, nslots, dim, cap)
    except Exception as e:          # pragma: no cover
        done_q.put((-1, "worker start: %r" % (e,)))
        return
    while True:
        task = task_q.get()
        if task is None:
            break
        slot, iid, flip = task
        try:
            u8, layer = load_item(infos[iid], dim)
            h0, w0 = layer.shape
            hz = 0
            if h0 * w0 > cap:       # a label larger than a slot: zoomed here (the slow way), identity maps on the device
                ys, xs = _zoom_index(h0, dim), _zoom_index(w0, dim)
                z = layer[np.maximum(ys, 0)][:, np.maximum(xs, 0)]
                z[ys < 0] = 0
                z[:, xs < 0] = 0
                layer, h0, w0, hz = z, dim, dim, 1
            imgs[slot] = u8[:, ::-1] if flip else u8
            labs[slot, :h0 * w0] = layer.reshape(-1)
            meta[slot, :6] = (h0, w0, hz, iid, flip, 0)
            done_q.put((slot, None))
        except Exception as e:
            meta[slot, 5] = 1
            done_q.put((slot, "image %d (%s): %r" % (iid, infos[iid].get("label"), e)))


# ---------------------------------------------------------------------------------------------- training-process side
def capped_workers(requested, cores=None):
    """Worker processes a rank may start: at most its core share minus two (the training thread and the feeder
    thread keep a core each); at least one.  cores=None: the mask this process is pinned to."""
    if cores is None:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(int(requested), cores - 2))


class PrefetchLoader(object):
    """Ring of shared-memory slots filled by worker processes, drained by a feeder thread into pinned staging
    buffers and copied to the device on a copy stream.  Iterating yields, per step, a dict of DEVICE tensors
    {"u8" [B,dim,dim,3] uint8 (already flipped), "labels" [B, stride] int64 (raw, un-zoomed), "src_hw" [B,2] int32,
    "flips" (host list), "host_zoomed" (host list), "ids" (host list), "ready" (a CUDA event on the copy stream)}.

    start() spawns the workers; call it BEFORE the process touches the GPU when the launcher allows (the workers are
    fresh `spawn` interpreters either way -- never a fork or an exec of a process that has initialised the GPU)."""

    def __init__(self, infos, dim, batch, sampler, workers=8, depth=3, cap_pixels=None, device=None, start_epoch=0):
        self.infos, self.dim, self.batch, self.sampler = list(infos), int(dim), int(batch), sampler
        self.workers_requested = max(1, int(workers))
        self.workers, self.depth = self.workers_requested, max(1, int(depth))
        self.cap = int(cap_pixels or max(dim * dim, 640 * 640))     # label pixels a slot holds (larger ones: zoomed by the worker)
        self.device = device
        self.nslots = (self.depth + 1) * self.batch
        self.epoch0 = start_epoch
        self._procs, self._shms, self._thread = [], [], None
        self._out = queue.Queue(maxsize=self.depth)
        self._stop = threading.Event()
        self._err = None
        self._device_index = 0
        self.stats = {"batches": 0, "wait_s": 0.0, "depth_sum": 0, "depth_min": None, "host_zoomed": 0}

    # -------------------------------------------------------------- lifecycle
    def start(self):
        if self._procs:
            return self
        # capped to this rank's share of the host cores (parallel.set_cpu_affinity pinned the process before this
        # call; the workers inherit the mask): eight ranks x eight workers on one host must not queue behind each
        # other's training threads
        self.workers = capped_workers(self.workers_requested)
        import multiprocessing as mp
        from multiprocessing import shared_memory
        ctx = mp.get_context("spawn")
        sizes = (self.nslots * self.dim * self.dim * 3, self.nslots * self.cap * 8, self.nslots * META * 4)
        self._shms = [shared_memory.SharedMemory(create=True, size=max(s, 8)) for s in sizes]
        names = [s.name for s in self._shms]
        self._imgs = np.ndarray((self.nslots, self.dim, self.dim, 3), np.uint8, buffer=self._shms[0].buf)
        self._labs = np.ndarray((self.nslots, self.cap), np.uint64, buffer=self._shms[1].buf)
        self._meta = np.ndarray((self.nslots, META), np.int32, buffer=self._shms[2].buf)
        self._task_q, self._done_q = ctx.Queue(), ctx.Queue()
        slim = [{"path": i["path"], "label": i["label"]} for i in self.infos]
        for _ in range(self.workers):
            p = ctx.Process(target=_worker_main, args=(names, self.nslots, self.dim, self.cap, slim, self._task_q,
                                                       self._done_q), daemon=True)
            p.start()
            self._procs.append(p)
        import atexit
        atexit.register(self.close)          # (the shared-memory segments live in /dev/shm: never left behind)
        return self

    def close(self):
        self._stop.set()
        for _ in self._procs:
            try:
                self._task_q.put(None)
            except Exception:
                pass
        if self._thread is not None:
            try:                         # unblock a feeder waiting on a full queue
                while True:
                    self._out.get_nowait()
            except queue.Empty:
                pass
            self._thread.join(timeout=10)
            self._thread = None
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._procs = []
        for name in ("_imgs", "_labs", "_meta"):
            if hasattr(self, name):
                delattr(self, name)
        for s in self._shms:
            try:
                s.close()
                s.unlink()
            except Exception:
                pass
        self._shms = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -------------------------------------------------------------- feeder thread
    def _schedule(self):
        """Generator of (ids, flips) per step, epoch after epoch."""
        e = self.epoch0
        while True:
            for ids, flips in self.sampler.epoch(e):
                yield ids, flips
            e += 1

    def _feed(self):
        import torch
        try:
            dev = self.device
            cuda = dev is not None and torch.device(dev).type == "cuda"
            if cuda:
                dev = torch.device(dev)
                if dev.index is None:            # a bare "cuda": the device the training thread made current
                    dev = torch.device("cuda", self._device_index)
                torch.cuda.set_device(dev)
                stream = torch.cuda.Stream(device=dev)
            B, dim = self.batch, self.dim
            nstage = self.depth + 2
            pin = dict(pin_memory=True) if cuda else {}
            st_img = [torch.empty((B, dim, dim, 3), dtype=torch.uint8, **pin) for _ in range(nstage)]
            st_lab = [torch.empty((B * self.cap,), dtype=torch.int64, **pin) for _ in range(nstage)]
            st_evt = [None] * nstage
            sched = self._schedule()
            free = list(range(self.nslots))
            pending = []                 # batches in flight: [slots, ids, flips, outstanding set]
            done = {}
            k = 0
            while not self._stop.is_set():
                # keep the workers busy: issue whole batches while slots are free
                while len(free) >= B and len(pending) < self.depth + 1:
                    ids, flips = next(sched)
                    slots = [free.pop() for _ in range(B)]
                    for s, i, f in zip(slots, ids, flips):
                        self._task_q.put((s, int(i), int(f)))
                    pending.append([slots, ids, flips, set(slots)])
                head = pending[0]
                while head[3]:
                    try:
                        slot, err = self._done_q.get(timeout=0.5)
                    except queue.Empty:
                        if self._stop.is_set():
                            return
                        if not all(p.is_alive() for p in self._procs):
                            raise RuntimeError("a loader worker died")
                        continue
                    if err is not None:
                        raise RuntimeError("loader worker: " + err)
                    done[slot] = True
                    for b in pending:
                        b[3].discard(slot)
                slots, ids, flips, _ = pending.pop(0)
                j = k % nstage
                if st_evt[j] is not None:
                    st_evt[j].synchronize()          # the copy that last read this staging buffer has finished
                meta = self._meta[slots].copy()
                npix = meta[:, 0].astype(np.int64) * meta[:, 1]
                if npix.min() < 0 or npix.max() > self.cap:
                    # (a worker's metadata that disagrees with what a slot can hold: fail here, on the host, instead
                    # of handing the device kernels a row pitch that walks out of the packed buffer)
                    raise RuntimeError("loader: a label of %d pixels in a slot of %d" % (int(npix.max()), self.cap))
                stride = int((npix.max() + 7) // 8 * 8)
                img_t, lab_t = st_img[j], st_lab[j][:B * stride].view(B, stride)
                img_np, lab_np = img_t.numpy(), lab_t.numpy()
                for r, s in enumerate(slots):
                    img_np[r] = self._imgs[s]
                    n = int(npix[r])
                    lab_np[r, :n] = self._labs[s, :n].view(np.int64)
                free.extend(slots)
                item = {"flips": [int(f) for f in flips], "ids": [int(i) for i in ids],
                        "host_zoomed": [int(v) for v in meta[:, 2]], "src_hw_host": meta[:, :2].copy()}
                if cuda:
                    with torch.cuda.stream(stream):
                        item["u8"] = img_t.to(dev, non_blocking=True)
                        item["labels"] = lab_t.to(dev, non_blocking=True)
                        item["src_hw"] = torch.from_numpy(meta[:, :2].copy()).pin_memory().to(dev, non_blocking=True)
                        evt = torch.cuda.Event()
                        evt.record(stream)
                    st_evt[j] = evt
                    item["ready"] = evt
                else:                                # (CPU tests of the plumbing: host tensors, copies)
                    item["u8"], item["labels"] = img_t.clone(), lab_t.clone()
                    item["src_hw"] = torch.from_numpy(meta[:, :2].copy())
                    item["ready"] = None
                self.stats["host_zoomed"] += int(meta[:, 2].sum())
                k += 1
                while not self._stop.is_set():
                    try:
                        self._out.put(item, timeout=0.5)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:       # handed to the consumer, which re-raises it
            self._err = e
            try:
                self._out.put_nowait(None)
            except queue.Full:
                pass

    # -------------------------------------------------------------- consumer
    def queue_depth(self):
        """Batches ready and waiting for the training thread right now (0 = the GPU is waiting for the loader)."""
        return self._out.qsize()

    def __iter__(self):
        self.start()
        if self._thread is None:
            if self.device is not None:
                import torch
                if torch.device(self.device).type == "cuda":
                    self._device_index = torch.cuda.current_device()     # (read on the consumer's thread)
            self._thread = threading.Thread(target=self._feed, name="sln-loader-feeder", daemon=True)
            self._thread.start()
        while True:
            d = self._out.qsize()
            st = self.stats
            st["depth_sum"] += d
            st["depth_min"] = d if st["depth_min"] is None else min(st["depth_min"], d)
            t0 = time.perf_counter()
            while True:
                if self._err is not None:
                    raise RuntimeError("input pipeline failed") from self._err
                try:
                    item = self._out.get(timeout=1.0)
                    break
                except queue.Empty:
                    continue
            if item is None:
                raise RuntimeError("input pipeline failed") from self._err
            st["wait_s"] += time.perf_counter() - t0
            st["batches"] += 1
            yield item

    def report(self):
        """-> {batches, mean / min queue depth seen by the consumer, seconds it waited, labels zoomed on the host}."""
        st = self.stats
        n = max(st["batches"], 1)
        return {"batches": st["batches"], "queue_depth_mean": round(st["depth_sum"] / n, 2),
                "queue_depth_min": st["depth_min"], "consumer_wait_ms_per_batch": round(1e3 * st["wait_s"] / n, 3),
                "labels_zoomed_on_host": st["host_zoomed"], "workers": self.workers, "workers_requested": self.workers_requested,
                "prefetch_batches": self.depth}


# ---------------------------------------------------------------------------------------------- synthetic file sets
def write_synthetic_pair(root, k, dim, n_obj=8, seed=1234):
    """One `<name>.jpg` + `<name>.npz` pair in the reference's on-disk format (amodal_train.py:238: the uint64
    occlusion label under 'layer'), COCOA-shape as SURVEY.md section 8(d) defines the synthetic scenes: n_obj
    axis-aligned ellipses, painter's order with object 0 on top -- low word = the visible object's bit, high word =
    the bits of the objects it covers.  numpy + Pillow only (runs in pool workers)."""
    from PIL import Image
    rs = np.random.RandomState((seed * 9973 + k) % (2 ** 32))
    yy, xx = np.mgrid[0:dim, 0:dim]
    base = rs.randint(0, 256, (dim // 8 + 1, dim // 8 + 1, 3)).astype(np.uint8)
    img = np.asarray(Image.fromarray(base).resize((dim, dim), Image.BILINEAR)).copy()     # smooth background
    lab = np.zeros((dim, dim), np.uint64)
    covered = np.zeros((dim, dim), bool)
    for i in range(n_obj):
        cy, cx = rs.uniform(0.125, 0.875, 2) * dim
        ry, rx = rs.uniform(0.047, 0.25, 2) * dim
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        lab[m & ~covered] |= np.uint64(1) << np.uint64(i)
        lab[m & covered] |= np.uint64(1) << np.uint64(32 + i)
        img[m & ~covered] = rs.randint(0, 256, 3).astype(np.uint8)
        covered |= m
    img = np.clip(img.astype(np.int16) + rs.randint(-12, 13, img.shape), 0, 255).astype(np.uint8)   # sensor-like noise
    stem = os.path.join(root, "syn%05d" % k)
    Image.fromarray(img).save(stem + ".jpg", quality=90)
    np.savez_compressed(stem + ".npz", layer=lab)
    return stem


def _write_range(args):
    root, ks, dim, n_obj, seed = args
    for k in ks:
        write_synthetic_pair(root, k, dim, n_obj, seed)
    return len(ks)


def write_synthetic_dataset(root, n, dim, n_obj=8, seed=1234, procs=8):
    """n pairs under `root` (created), written by `procs` spawned processes; idempotent: a finished set (marker file
    with the same parameters) is reused, concurrent callers (the ranks of one job) wait for the one that writes."""
    os.makedirs(root, exist_ok=True)
    marker = os.path.join(root, ".complete")
    tag = "%d %d %d %d" % (n, dim, n_obj, seed)
    lock = os.path.join(root, ".writing")
    t0 = time.time()
    while True:
        if os.path.exists(marker) and open(marker).read().strip() == tag:
            return root
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
            os.close(fd)
            break
        except FileExistsError:
            if time.time() - t0 > 1800:
                raise RuntimeError("write_synthetic_dataset: %s is held by another writer" % lock)
            time.sleep(0.5)
    try:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        procs = max(1, min(procs, n))
        chunks = [(root, list(range(p, n, procs)), dim, n_obj, seed) for p in range(procs)]
        with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as pool:
            assert sum(pool.map(_write_range, chunks)) == n
        with open(marker, "w") as f:
            f.write(tag)
    finally:
        os.unlink(lock)
    return root
