"""MXNet on-disk formats on the edges of the hot path, pure host code:

* `.params` — NDArray-list files written by `net.save_parameters` / `mx.callback.do_checkpoint` (ref: train_efm.py:289-290;
  mutli_gpu_v3.py:160; read back at final_efm.py:213 and extract_feacture_v2.py:51);
* RecordIO `.rec` (+ `.lst`) image datasets read by `mx.io.ImageRecordIter` (ref: train_efm.py:179-181).

PARITY UNPINNED: no MXNet-written file exists in the reference tree or this environment, and MXNet is not installable
here; both formats are restated from MXNet 1.x's published layouts (src/ndarray/ndarray.cc `NDArray::Save`,
dmlc-core recordio.h, python/mxnet/recordio.py `IRHeader`) and are tested by round trips only.
"""
import io
import struct

import numpy as np

_LIST_MAGIC = 0x112
_ND_V1, _ND_V2, _ND_V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_DTYPE_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------ .params
def save_params(path, params):
    """params: {name: array}.  Dense NDArray V2 records, cpu(0) context — what mx.nd.save writes."""
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", _LIST_MAGIC, 0, len(params)))
        for arr in params.values():
            a = np.ascontiguousarray(np.asarray(arr))
            if a.dtype not in _DTYPE_FLAGS:
                a = a.astype(np.float32)
            f.write(struct.pack("<Ii", _ND_V2, 0))                       # magic, stype = kDefaultStorage
            f.write(struct.pack("<I", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, _DTYPE_FLAGS[a.dtype]))     # Context{cpu, 0}, type flag
            f.write(a.tobytes())
        f.write(struct.pack("<Q", len(params)))
        for name in params:
            b = name.encode()
            f.write(struct.pack("<Q", len(b)) + b)


def load_params(path, strip_prefix=True):
    """-> {name: numpy array}.  Accepts V1/V2/V3 dense records; 'arg:' / 'aux:' prefixes of Module checkpoints are
    stripped when `strip_prefix`."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, buf, off)
        off += struct.calcsize(fmt)
        return v

    magic, _, count = take("<QQQ")
    if magic != _LIST_MAGIC:
        raise ValueError("%s: not an MXNet NDArray list file (magic %#x)" % (path, magic))
    arrays = []
    for _ in range(count):
        (m,) = take("<I")
        if m in (_ND_V2, _ND_V3):
            (stype,) = take("<i")
            if stype != 0:
                raise NotImplementedError("sparse NDArray (stype %d)" % stype)
            (ndim,) = take("<I")
            shape = take("<%dq" % ndim) if ndim else ()
        elif m == _ND_V1:
            (ndim,) = take("<I")
            shape = take("<%dq" % ndim) if ndim else ()
        else:  # legacy: the word just read is ndim, dims are uint32
            ndim = m
            shape = take("<%dI" % ndim) if ndim else ()
        if ndim == 0:
            arrays.append(np.zeros((), np.float32))
            continue
        take("<ii")  # context
        (flag,) = take("<i")
        dt = np.dtype(_DTYPES[flag])
        n = int(np.prod(shape))
        arrays.append(np.frombuffer(buf, dtype=dt, count=n, offset=off).reshape(shape).copy())
        off += n * dt.itemsize
    (ncount,) = take("<Q")
    names = []
    for _ in range(ncount):
        (ln,) = take("<Q")
        names.append(buf[off:off + ln].decode())
        off += ln
    if ncount == 0:
        names = [str(i) for i in range(count)]
    out = {}
    for n, a in zip(names, arrays):
        if strip_prefix and (n.startswith("arg:") or n.startswith("aux:")):
            n = n[4:]
        out[n] = a
    return out


# ----------------------------------------------------------------------------------------- RecordIO
_REC_MAGIC = 0xCED7230A


def read_records(path):
    """Yields the payload bytes of every record of a .rec file (multi-part records are re-joined)."""
    with open(path, "rb") as f:
        pending = b""
        while True:
            head = f.read(8)
            if len(head) < 8:
                return
            magic, lrec = struct.unpack("<II", head)
            if magic != _REC_MAGIC:
                raise ValueError("%s: bad RecordIO magic %#x" % (path, magic))
            cflag, length = lrec >> 29, lrec & ((1 << 29) - 1)
            data = f.read(length)
            f.read((4 - length % 4) % 4)
            if cflag == 0:
                yield data
            elif cflag == 1:
                pending = data
            elif cflag == 2:
                pending += struct.pack("<I", _REC_MAGIC) + data
            else:
                yield pending + struct.pack("<I", _REC_MAGIC) + data
                pending = b""


def write_records(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            f.write(struct.pack("<II", _REC_MAGIC, len(p)) + p + b"\0" * ((4 - len(p) % 4) % 4))


def pack_img(label, index, img, fmt="PNG"):
    """IRHeader(flag, label, id, id2) + encoded image (mx.recordio.pack_img)."""
    from PIL import Image
    bio = io.BytesIO()
    Image.fromarray(img).save(bio, format=fmt)
    labels = np.atleast_1d(np.asarray(label, dtype=np.float32))
    if labels.size == 1:
        head = struct.pack("<IfQQ", 0, float(labels[0]), index, 0)
    else:
        head = struct.pack("<IfQQ", labels.size, 0.0, index, 0) + labels.tobytes()
    return head + bio.getvalue()


def unpack_img(payload, gray=False):
    """-> (label (float or float array), id, HxW or HxWx3 uint8 array)."""
    from PIL import Image
    flag, label, idx, _ = struct.unpack_from("<IfQQ", payload, 0)
    off = struct.calcsize("<IfQQ")
    if flag > 0:
        label = np.frombuffer(payload, dtype=np.float32, count=flag, offset=off).copy()
        off += 4 * flag
    img = Image.open(io.BytesIO(payload[off:]))
    img = img.convert("L" if gray else "RGB")
    return label, idx, np.asarray(img)


def read_lst(path):
    """`.lst` lines: index <tab> label(s) <tab> relative path (ref: train_efm.py:142-148 only counts them)."""
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) >= 3:
                rows.append((int(parts[0]), [float(v) for v in parts[1:-1]], parts[-1]))
    return rows


def index_records(path):
    """[(payload offset, payload length)] of every whole record of a .rec file, read from the 8-byte record headers only
    (no payload is read or decoded).  Multi-part records (cflag 1/2/3) are reported as one entry starting at their first part
    with length -1: `read_record_at` re-joins them."""
    out = []
    with open(path, "rb") as f:
        f.seek(0, 2)
        end = f.tell()
        off, start = 0, None
        while off + 8 <= end:
            f.seek(off)
            magic, lrec = struct.unpack("<II", f.read(8))
            if magic != _REC_MAGIC:
                raise ValueError("%s: bad RecordIO magic %#x at byte %d" % (path, magic, off))
            cflag, length = lrec >> 29, lrec & ((1 << 29) - 1)
            if cflag == 0:
                out.append((off, length))
            elif cflag == 1:
                start = off
            elif cflag == 3:
                out.append((start, -1))
            off += 8 + length + (4 - length % 4) % 4
    return out


def read_record_at(f, off, length):
    """Payload of the record whose header starts at byte `off` of the open file `f`."""
    if length >= 0:
        f.seek(off + 8)
        return f.read(length)
    f.seek(off)
    parts = []
    while True:
        magic, lrec = struct.unpack("<II", f.read(8))
        cflag, ln = lrec >> 29, lrec & ((1 << 29) - 1)
        parts.append(f.read(ln))
        f.read((4 - ln % 4) % 4)
        if cflag == 3:
            return struct.pack("<I", _REC_MAGIC).join(parts)


def pread_record_at(fd, off, length):
    """`read_record_at` on a file DESCRIPTOR with positional reads (os.pread): no shared file position, so the decode workers of
    `ImageRecordIter` read concurrently from one descriptor."""
    import os
    if length >= 0:
        return os.pread(fd, length, off + 8)
    parts = []
    while True:
        magic, lrec = struct.unpack("<II", os.pread(fd, 8, off))
        cflag, ln = lrec >> 29, lrec & ((1 << 29) - 1)
        parts.append(os.pread(fd, ln, off + 8))
        off += 8 + ln + (4 - ln % 4) % 4
        if cflag == 3:
            return struct.pack("<I", _REC_MAGIC).join(parts)


_WORKER_FD = {}


def _decode_slice(task):
    """Body of a decode-pool worker (decode_worker.py, a separate PROCESS: PIL holds the GIL for about half of a small image's decode, so threads convoy):
    task = (path, [(offset, length)...], gray) -> ([label...], uint8 array (n, H, W[, C]) when all images share one size, else a list)."""
    import os
    path, entries, gray = task
    fd = _WORKER_FD.get(path)
    if fd is None:
        fd = _WORKER_FD[path] = os.open(path, os.O_RDONLY)
    labels, imgs = [], []
    for off, length in entries:
        label, _, img = unpack_img(pread_record_at(fd, off, length), gray=gray)
        labels.append(float(np.atleast_1d(label)[0]))
        imgs.append(img)
    if len({im.shape for im in imgs}) == 1:
        imgs = np.stack(imgs)
    return labels, imgs


class _DecodePool:
    """N decode_worker child processes (plain subprocesses over pipes: no multiprocessing start-method pitfalls, no re-import of the
    caller's script, nothing inherited from a GPU-initialised parent).  map() hands task i to worker i and collects in order."""

    def __init__(self, n):
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        self.procs = [subprocess.Popen([sys.executable, "-m", __package__ + ".decode_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                       env=env, cwd=root) for _ in range(n)]

    def map(self, tasks):
        import pickle
        assert len(tasks) <= len(self.procs)
        for p, t in zip(self.procs, tasks):
            pickle.dump(t, p.stdin, protocol=pickle.HIGHEST_PROTOCOL)
            p.stdin.flush()
        out = []
        for p, _ in zip(self.procs, tasks):
            try:
                r = pickle.load(p.stdout)
            except EOFError:
                raise RuntimeError("decode worker died (exit code %s)" % p.poll())
            if isinstance(r, BaseException):
                raise r
            out.append(r)
        return out

    def close(self):
        import pickle
        for p in self.procs:
            try:
                pickle.dump(None, p.stdin)
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.procs = []


class ImageRecordIter:
    """mx.io.ImageRecordIter(path_imgrec, data_shape=(C,H,W), batch_size, scale, rand_crop, rand_mirror, shuffle, part_index,
    num_parts, preprocess_threads, prefetch_buffer) stand-in (ref: train_efm.py:179-181): emits NCHW float32 batches scaled by
    `scale`; images larger than (H, W) are randomly (or centre-) cropped, smaller ones are an error; the last partial batch is dropped.

    Streaming: __init__ only indexes the record headers (16 bytes of host memory per image — the reference's 4.6 M-image set is
    74 MB of index, not 80 GB of decoded pixels); a batch is decoded, cropped and mirrored when it is asked for, with a fresh
    crop / mirror / order every epoch, as MXNet's iterator does.

    Parallel like MXNet's C++ iterator (its `preprocess_threads` / `prefetch_buffer` parameters): a producer thread assembles
    batches AHEAD of the consumer — the records of a batch are read (positional reads) and decoded by a pool of
    `preprocess_threads` worker PROCESSES, each taking a contiguous slice of the batch (processes, not threads: for face-sized
    images PIL holds the GIL for about half of a decode and a thread pool runs slower than one thread — measured); 1 = decode on
    the producer thread itself.  The crop / mirror draws are then taken sequentially in record order, exactly the draws of the
    synchronous path, so the batches are bit-identical to it whatever the worker count.  With
    `device=` the decoded uint8 images go into a pinned staging buffer and cross PCIe on a copy stream while the previous step is
    still computing; `efm_crop_mirror_u8` (crop, mirror, scale, uint8 -> fp32 NCHW) then runs on the consumer's stream.
    `preprocess_threads=0` is the synchronous path (decode on the calling thread when the batch is asked for).

    `part_index` / `num_parts` (MXNet's own parameters for distributed reading) give each data-parallel rank a disjoint share of
    the records.  Every share holds EXACTLY n // num_parts records (the n % num_parts records at the end are not read this
    epoch by anyone): ranks that all-reduce once per batch must see the same number of batches, or the rank with one more blocks
    forever in its collective."""

    def __init__(self, path_imgrec, data_shape, batch_size, scale=1.0, rand_crop=False, rand_mirror=False, shuffle=False, seed=0,
                 part_index=0, num_parts=1, device=None, preprocess_threads=None, prefetch_buffer=2, **_):
        import os
        if not 0 <= part_index < num_parts:
            raise ValueError("part_index %d outside [0, %d)" % (part_index, num_parts))
        self.path, self.data_shape, self.batch_size, self.scale = path_imgrec, tuple(data_shape), batch_size, scale
        self.rand_crop, self.rand_mirror, self.shuffle = rand_crop, rand_mirror, shuffle
        index = index_records(path_imgrec)
        n = len(index)
        share = n // num_parts                                       # equal shares: equal batch counts on every rank
        self.index = index[part_index * share: (part_index + 1) * share]   # contiguous chunk, like MXNet's partition
        self.num_total = n
        self._rng = np.random.default_rng(seed)
        # device: crop / mirror / scale / uint8 -> fp32 run on the GPU (efm_crop_mirror_u8) and the batch is born there: the host only
        # decodes, and 4x fewer bytes cross PCIe.  Same random draws in the same order as the host path: bit-identical batches.
        self.device = device
        if preprocess_threads is None:
            preprocess_threads = int(os.environ.get("EFM_DECODE_THREADS", min(16, os.cpu_count() or 1)))
        self.threads = max(0, int(preprocess_threads))
        self.prefetch = max(1, int(prefetch_buffer))
        self._fd = None
        self._pool = None
        self._producer = None
        self._queue = None
        self._stop = None
        self._copy_stream = None
        self._staging = {}
        self._order = np.arange(len(self.index))
        self.epoch = -1
        self.reset()

    def __len__(self):
        return len(self.index)

    def __iter__(self):
        self.reset()
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        import os
        self._stop_producer()
        if self._pool is not None:
            self._pool.close()
            self._pool = None
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None

    # ---- one record: read + decode (any thread), then its crop / mirror draws (sequential, in record order)
    def _load(self, k):
        c, h, w = self.data_shape
        label, _, img = unpack_img(pread_record_at(self._fd, *self.index[k]), gray=(c == 1))
        ih, iw = img.shape[:2]
        if ih < h or iw < w:
            raise ValueError("record image %dx%d smaller than data_shape %dx%d" % (ih, iw, h, w))
        return img, float(np.atleast_1d(label)[0])

    def _draw(self, img):
        c, h, w = self.data_shape
        ih, iw = img.shape[:2]
        y0 = int(self._rng.integers(0, ih - h + 1)) if self.rand_crop else (ih - h) // 2
        x0 = int(self._rng.integers(0, iw - w + 1)) if self.rand_crop else (iw - w) // 2
        flip = bool(self.rand_mirror and self._rng.random() < 0.5)
        return y0, x0, flip

    def _assemble(self, pos):
        """Batch starting at `pos` of the epoch order -> a host-side item: ("dev", pinned uint8 (B, IH, IW, C), crops, labels, event)
        when the GPU does the cropping, else ("host", uint8 (B, C, H, W), labels)."""
        import os

        import torch
        if self._fd is None:
            self._fd = os.open(self.path, os.O_RDONLY)
        c, h, w = self.data_shape
        ks = [int(self._order[pos + j]) for j in range(self.batch_size)]
        if self.threads > 1:
            if self._pool is None:
                self._pool = _DecodePool(self.threads)
            per = -(-self.batch_size // self.threads)
            tasks = [(self.path, [self.index[k] for k in ks[i:i + per]], c == 1) for i in range(0, self.batch_size, per)]
            loaded = []
            for labels_s, imgs_s in self._pool.map(tasks):
                loaded += [(im, lb) for im, lb in zip(imgs_s, labels_s)]
            for im, _ in loaded:
                if im.shape[0] < h or im.shape[1] < w:
                    raise ValueError("record image %dx%d smaller than data_shape %dx%d" % (im.shape[0], im.shape[1], h, w))
        else:
            loaded = [self._load(k) for k in ks]
        labels = np.array([lb for _, lb in loaded], dtype=np.float32)
        crops = np.empty((self.batch_size, 3), dtype=np.int32)
        for j, (img, _) in enumerate(loaded):
            crops[j] = self._draw(img)
        imgs = [img if img.ndim == 3 else img[:, :, None] for img, _ in loaded]
        if self.device is not None and len({im.shape for im in imgs}) == 1:
            # one source size per batch (the usual pre-resized .rec): the GPU path.  Pinned staging buffers rotate; a buffer is
            # rewritten only after the copy that read it has finished (its event).
            shape = (self.batch_size,) + imgs[0].shape
            slot = self._slot = (getattr(self, "_slot", -1) + 1) % (self.prefetch + 2)
            st = self._staging.get(slot)
            if st is None or tuple(st[0].shape) != shape:
                st = self._staging[slot] = [torch.empty(shape, dtype=torch.uint8).pin_memory(), torch.empty((self.batch_size, 3), dtype=torch.int32).pin_memory(), None]
            if st[2] is not None:
                st[2].synchronize()
            np.stack(imgs, out=st[0].numpy())
            st[1].numpy()[:] = crops
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._copy_stream):
                src = st[0].to(self.device, non_blocking=True)
                crop_d = st[1].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            st[2] = ev
            return ("dev", src, crop_d, labels, ev)
        data = np.empty((self.batch_size, c, h, w), dtype=np.uint8)   # host crop (no device, or mixed source sizes): same draws
        for j, (im, (y0, x0, flip)) in enumerate(zip(imgs, crops)):
            im = im[y0:y0 + h, x0:x0 + w]
            data[j] = (im[:, ::-1] if flip else im).transpose(2, 0, 1)
        return ("host", data, labels)

    # ---- producer thread: batches of the current epoch, at most `prefetch` ahead of the consumer
    def _produce(self, first_pos, q, stop):
        try:
            pos = first_pos
            while pos + self.batch_size <= len(self.index) and not stop.is_set():
                state = self._rng.bit_generator.state      # lets a mid-epoch reset() hand back the draws of unconsumed batches
                item = self._assemble(pos)
                pos += self.batch_size
                while not stop.is_set():
                    try:
                        q.put((item, state), timeout=0.05)
                        break
                    except Exception:
                        continue
            q_put_forever(q, (None, None), stop)
        except BaseException as e:  # surfaces in the consumer
            q_put_forever(q, (e, None), stop)

    def _start_producer(self):
        import queue
        import threading
        self._queue = queue.Queue(maxsize=self.prefetch)
        self._stop = threading.Event()
        self._producer = threading.Thread(target=self._produce, args=(self.pos, self._queue, self._stop), name="efm-record-iter", daemon=True)
        self._producer.start()

    def _stop_producer(self):
        """Stop the producer and give the random stream back the draws of every batch that was assembled but never consumed."""
        if self._producer is None:
            return
        self._stop.set()
        first_unconsumed = None
        while self._producer.is_alive() or not self._queue.empty():
            try:
                item, state = self._queue.get(timeout=0.05)
            except Exception:
                continue
            if first_unconsumed is None and state is not None:
                first_unconsumed = state
        self._producer.join()
        self._producer = None
        if first_unconsumed is not None:
            self._rng.bit_generator.state = first_unconsumed

    def __next__(self):
        import torch

        from .data import Batch
        if self.pos + self.batch_size > len(self.index):
            raise StopIteration
        c, h, w = self.data_shape
        if self.threads > 0:
            if self._producer is None:
                self._start_producer()
            item, _ = self._queue.get()
            if isinstance(item, BaseException):
                self._producer.join()
                self._producer = None
                raise item
            if item is None:
                raise StopIteration
        else:
            item = self._assemble(self.pos)
        self.pos += self.batch_size
        if item[0] == "dev":
            from . import ops
            _, src, crop_d, labels, ev = item
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            src.record_stream(cur)
            crop_d.record_stream(cur)
            x = ops.crop_mirror_u8(src, crop_d, h, w, self.scale)
            return Batch(["data"], [x], ["softmax_label"], [torch.from_numpy(labels)])
        _, data, labels = item
        x = torch.from_numpy(data).to(torch.float32)
        if self.scale != 1.0:
            x *= self.scale
        if self.device is not None:
            x = x.to(self.device)
        return Batch(["data"], [x], ["softmax_label"], [torch.from_numpy(labels)])

    next = __next__

    def reset(self):
        self._stop_producer()
        self.pos = 0
        self.epoch += 1
        if self.shuffle:
            self._order = self._rng.permutation(len(self.index))


def q_put_forever(q, item, stop):
    while not stop.is_set():
        try:
            q.put(item, timeout=0.05)
            return
        except Exception:
            continue


# ---------------------------------------------------------------------------------------------------------------------
# MXNet `-symbol.json` (the graph half of a checkpoint: mx.callback.do_checkpoint / mx.sym.load, ref: mutli_gpu_v3.py:160,
# final_efm.py:205-211, extract_feacture_v2.py:47-51).  Export expands the one-node MFM of graph.py into the reference's
# SliceChannel / maximum / minimum / Concat idiom (efm_symbol.py:25-30, 68-77) with its operand order; import folds that idiom
# back, so a file written by the reference's own builder loads onto the fused kernels.  Parity UNPINNED: the reference holds no
# .json file (only the builder code and two PDFs of the graph); the schema below is MXNet 1.x's (nodes / arg_nodes /
# node_row_ptr / heads, attrs as strings) and a round trip through it is what the tests check.
def save_symbol(path, outputs):
    """Write the network whose output nodes are `outputs` (graph.Sym) as an MXNet symbol JSON file."""
    import json

    from . import graph as G
    nodes, out_of = [], {}          # out_of[sym id] = (node index, output index)

    def add(op, name, inputs=(), attrs=None):
        n = {"op": op, "name": name, "inputs": [[i, o, 0] for i, o in inputs]}
        if attrs:
            n["attrs"] = {k: str(v) for k, v in attrs.items()}
        nodes.append(n)
        return len(nodes) - 1

    def tup(v):
        return "(%d, %d)" % (v[0], v[1])

    for s in G.topo_sort(list(outputs)):
        ins = [out_of[i.id] for i in s.inputs]
        if s.op == "var":
            out_of[s.id] = (add("null", s.name), 0)
        elif s.op in ("conv", "fc"):
            w = add("null", s.name + "_weight")
            extra = [(w, 0)]
            if not s.attrs.get("no_bias"):
                extra.append((add("null", s.name + "_bias"), 0))
            if s.op == "conv":
                a = {"kernel": tup(s.attrs["kernel"]), "num_filter": s.attrs["num_filter"], "pad": tup(s.attrs["pad"]),
                     "stride": tup(s.attrs.get("stride", (1, 1)))}
                if s.attrs.get("no_bias"):
                    a["no_bias"] = "True"
                out_of[s.id] = (add("Convolution", s.name, ins + extra, a), 0)
            else:
                a = {"num_hidden": s.attrs["num_hidden"]}
                if s.attrs.get("no_bias"):
                    a["no_bias"] = "True"
                out_of[s.id] = (add("FullyConnected", s.name, ins + extra, a), 0)
        elif s.op == "mfm":
            ways, order = s.attrs["ways"], s.attrs.get("order", G.ORDER_GROUP)
            sl = add("SliceChannel", "slice_" + s.name, ins, {"axis": 1, "num_outputs": ways})
            mx1 = add("_maximum", s.name if ways == 2 else s.name + "_max1", [(sl, 0), (sl, 1)])
            if ways == 2:
                out_of[s.id] = (mx1, 0)
                continue
            mn1 = add("_minimum", s.name + "_min1", [(sl, 0), (sl, 1)])
            if order == G.ORDER_GROUP:      # maximum(max1, s2)        ref: efm_symbol.py:70-73
                mx2 = add("_maximum", s.name + "_max2", [(mx1, 0), (sl, 2)])
                mn2 = add("_minimum", s.name + "_min2", [(mn1, 0), (sl, 2)])
            else:                           # maximum(s2, max1)        ref: efm_symbol.py:26-29
                mx2 = add("_maximum", s.name + "_max2", [(sl, 2), (mx1, 0)])
                mn2 = add("_minimum", s.name + "_min2", [(sl, 2), (mn1, 0)])
            out_of[s.id] = (add("Concat", s.name, [(mx2, 0), (mn2, 0)], {"dim": 1, "num_args": 2}), 0)
        elif s.op == "pool":
            out_of[s.id] = (add("Pooling", s.name, ins, {"kernel": "(2, 2)", "pool_type": "max", "stride": "(2, 2)"}), 0)
        elif s.op == "add":
            out_of[s.id] = (add("elemwise_add", s.name, ins), 0)
        elif s.op == "l2norm":
            out_of[s.id] = (add("L2Normalization", s.name, ins, {"mode": "instance"}), 0)
        else:
            raise ValueError("save_symbol: no MXNet operator for '%s'" % s.op)
    row, ptr = [0], 0
    for n in nodes:
        ptr += int(n.get("attrs", {}).get("num_outputs", 1)) if n["op"] == "SliceChannel" else 1
        row.append(ptr)
    doc = {"nodes": nodes, "arg_nodes": [i for i, n in enumerate(nodes) if n["op"] == "null"], "node_row_ptr": row,
           "heads": [[out_of[o.id][0], out_of[o.id][1], 0] for o in outputs], "attrs": {"mxnet_version": ["int", 10301]}}
    with open(path, "w") as f:
        json.dump(doc, f, indent=2)


_MAX_OPS = ("_maximum", "_Maximum", "maximum", "broadcast_maximum")
_MIN_OPS = ("_minimum", "_Minimum", "minimum", "broadcast_minimum")
_ADD_OPS = ("elemwise_add", "_plus", "_Plus", "broadcast_add", "_add")


def load_symbol(path, outputs=None):
    """Read an MXNet symbol JSON file -> list of graph.Sym output nodes (the file's heads, or the internals named in `outputs`, e.g.
    ["fc2_output", "concat29_output"] as final_efm.py:207-210 picks them).  Handles the operators the reference's builders emit;
    Flatten / Dropout / SoftmaxOutput pass through (the plan flattens in FullyConnected, dropout and softmax belong to the id head)."""
    import ast
    import json

    from . import graph as G
    doc = json.load(open(path))
    nodes = doc["nodes"]

    def attrs(n):
        return n.get("attrs") or n.get("attr") or n.get("param") or {}

    built = {}

    def entry(e):
        return build(e[0], e[1])

    def is_slice_out(e, sl, k):
        return e[0] == sl and e[1] == k

    def fold_mfm(i):
        """nodes[i] is a Concat or a maximum: recognise the MFM idiom, return the Sym or None."""
        n = nodes[i]
        if n["op"] in _MAX_OPS:                                   # MFM2: maximum(slice[0], slice[1])
            a, b = n["inputs"]
            sl = a[0]
            if nodes[sl]["op"] == "SliceChannel" and int(attrs(nodes[sl]).get("num_outputs", 0)) == 2 and b[0] == sl and (a[1], b[1]) == (0, 1):
                return G.MFM(entry(nodes[sl]["inputs"][0]), 2, G.ORDER_GROUP, name=n["name"])
            return None
        if n["op"] != "Concat" or len(n["inputs"]) != 2:
            return None
        mx2, mn2 = nodes[n["inputs"][0][0]], nodes[n["inputs"][1][0]]
        if mx2["op"] not in _MAX_OPS or mn2["op"] not in _MIN_OPS:
            return None

        def tree(top, ops):
            """-> (slice node, order) if top = op(op(s0, s1), s2) [GROUP] or op(s2, op(s0, s1)) [RES]."""
            a, b = top["inputs"]
            for inner, other, order in ((a, b, G.ORDER_GROUP), (b, a, G.ORDER_RES)):
                m1 = nodes[inner[0]]
                if m1["op"] in ops and nodes[other[0]]["op"] == "SliceChannel" and other[1] == 2:
                    sl = other[0]
                    x, y = m1["inputs"]
                    if is_slice_out(x, sl, 0) and is_slice_out(y, sl, 1) and int(attrs(nodes[sl]).get("num_outputs", 0)) == 3:
                        return sl, order
            return None
        t1, t2 = tree(mx2, _MAX_OPS), tree(mn2, _MIN_OPS)
        if t1 is None or t2 is None or t1 != t2:
            return None
        return G.MFM(entry(nodes[t1[0]]["inputs"][0]), 3, t1[1], name=n["name"])

    def build(i, out=0):
        if (i, out) in built:
            return built[(i, out)]
        n = nodes[i]
        op, a = n["op"], attrs(n)
        data_in = [e for e in n["inputs"] if nodes[e[0]]["op"] != "null" or not nodes[e[0]]["name"].endswith(("_weight", "_bias", "_label"))]
        if op == "null":
            s = G.Variable(n["name"])
        elif op == "Convolution":
            k, p = ast.literal_eval(a["kernel"]), ast.literal_eval(a.get("pad", "(0, 0)"))
            st = ast.literal_eval(a.get("stride", "(1, 1)"))
            s = G.Convolution(entry(data_in[0]), int(a["num_filter"]), tuple(k), n["name"], pad=tuple(p), stride=tuple(st),
                              no_bias=str(a.get("no_bias", "False")) == "True")
        elif op == "FullyConnected":
            s = G.FullyConnected(entry(data_in[0]), int(a["num_hidden"]), n["name"], no_bias=str(a.get("no_bias", "False")) == "True")
        elif op == "Pooling":
            if a.get("pool_type", "max") != "max" or ast.literal_eval(a.get("kernel", "(2, 2)")) != (2, 2):
                raise ValueError("load_symbol: only 2x2 max pooling is on the path (node %s)" % n["name"])
            s = G.Pooling(entry(data_in[0]), name=n["name"])
        elif op in _ADD_OPS:
            s = entry(data_in[0]) + entry(data_in[1])
            s.name = n["name"]
        elif op in ("Flatten", "Dropout", "SoftmaxOutput", "identity", "_copy"):
            s = entry(data_in[0])
        elif op == "L2Normalization":
            s = G.L2Normalization(entry(data_in[0]), name=n["name"])
        elif op in _MAX_OPS or op == "Concat":
            s = fold_mfm(i)
            if s is None:
                raise ValueError("load_symbol: '%s' (%s) is not part of a max / min feature-map idiom" % (n["name"], op))
        else:
            raise ValueError("load_symbol: operator '%s' (node %s) is not on the path" % (op, n["name"]))
        built[(i, out)] = s
        return s

    if outputs is None:
        return [build(h[0], h[1]) for h in doc["heads"]]
    by_name = {n["name"] + "_output": i for i, n in enumerate(nodes)}
    return [build(by_name[o]) for o in outputs]
