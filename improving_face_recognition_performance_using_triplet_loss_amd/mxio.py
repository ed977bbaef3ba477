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


class ImageRecordIter:
    """mx.io.ImageRecordIter(path_imgrec, data_shape=(C,H,W), batch_size, scale, rand_crop, rand_mirror, shuffle)
    stand-in (ref: train_efm.py:179-181): decodes on the host, emits NCHW float32 batches scaled by `scale`; images
    larger than (H, W) are randomly (or centre-) cropped, smaller ones are an error; the last partial batch is dropped."""

    def __init__(self, path_imgrec, data_shape, batch_size, scale=1.0, rand_crop=False, rand_mirror=False, shuffle=False, seed=0, **_):
        import torch
        self._torch = torch
        c, h, w = data_shape
        self.batch_size, self.scale = batch_size, scale
        rng = np.random.default_rng(seed)
        data, labels = [], []
        for payload in read_records(path_imgrec):
            label, _, img = unpack_img(payload, gray=(c == 1))
            ih, iw = img.shape[:2]
            if ih < h or iw < w:
                raise ValueError("record image %dx%d smaller than data_shape %dx%d" % (ih, iw, h, w))
            y0 = int(rng.integers(0, ih - h + 1)) if rand_crop else (ih - h) // 2
            x0 = int(rng.integers(0, iw - w + 1)) if rand_crop else (iw - w) // 2
            img = img[y0:y0 + h, x0:x0 + w]
            if rand_mirror and rng.random() < 0.5:
                img = img[:, ::-1]
            img = img[None] if c == 1 else img.transpose(2, 0, 1)
            data.append(np.ascontiguousarray(img))
            labels.append(float(np.atleast_1d(label)[0]))
        order = rng.permutation(len(data)) if shuffle else np.arange(len(data))
        self.data_arr = torch.from_numpy(np.stack([data[i] for i in order]).astype(np.float32) * scale)
        self.label_arr = torch.tensor([labels[i] for i in order], dtype=torch.float32)
        self.pos = 0

    def __iter__(self):
        self.reset()
        return self

    def __next__(self):
        from .data import Batch
        if self.pos + self.batch_size > len(self.data_arr):
            raise StopIteration
        s = slice(self.pos, self.pos + self.batch_size)
        self.pos += self.batch_size
        return Batch(["data"], [self.data_arr[s]], ["softmax_label"], [self.label_arr[s]])

    next = __next__

    def reset(self):
        self.pos = 0
