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


class ImageRecordIter:
    """mx.io.ImageRecordIter(path_imgrec, data_shape=(C,H,W), batch_size, scale, rand_crop, rand_mirror, shuffle, part_index,
    num_parts) stand-in (ref: train_efm.py:179-181): emits NCHW float32 batches scaled by `scale`; images larger than (H, W)
    are randomly (or centre-) cropped, smaller ones are an error; the last partial batch is dropped.

    Streaming: __init__ only indexes the record headers (16 bytes of host memory per image — the reference's 4.6 M-image set is
    74 MB of index, not 80 GB of decoded pixels); a batch is decoded, cropped and mirrored when it is asked for, with a fresh
    crop / mirror / order every epoch, as MXNet's iterator does.  `part_index` / `num_parts` (MXNet's own parameters for
    distributed reading) give each data-parallel rank a disjoint 1/num_parts of the records, so that one epoch covers the data
    set once however many ranks read it."""

    def __init__(self, path_imgrec, data_shape, batch_size, scale=1.0, rand_crop=False, rand_mirror=False, shuffle=False, seed=0,
                 part_index=0, num_parts=1, device=None, **_):
        if not 0 <= part_index < num_parts:
            raise ValueError("part_index %d outside [0, %d)" % (part_index, num_parts))
        self.path, self.data_shape, self.batch_size, self.scale = path_imgrec, tuple(data_shape), batch_size, scale
        self.rand_crop, self.rand_mirror, self.shuffle = rand_crop, rand_mirror, shuffle
        index = index_records(path_imgrec)
        n = len(index)
        self.index = index[part_index * n // num_parts: (part_index + 1) * n // num_parts]   # contiguous chunk, like MXNet's partition
        self.num_total = n
        self._rng = np.random.default_rng(seed)
        # device: crop / mirror / scale / uint8 -> fp32 run on the GPU (efm_crop_mirror_u8) and the batch is born there: the host only
        # decodes, and 4x fewer bytes cross PCIe.  Same random draws in the same order as the host path: bit-identical batches.
        self.device = device
        self._file = None
        self._order = np.arange(len(self.index))
        self.epoch = -1
        self.reset()

    def __len__(self):
        return len(self.index)

    def __iter__(self):
        self.reset()
        return self

    def _read(self, k):
        c, h, w = self.data_shape
        label, _, img = unpack_img(read_record_at(self._file, *self.index[k]), gray=(c == 1))
        ih, iw = img.shape[:2]
        if ih < h or iw < w:
            raise ValueError("record image %dx%d smaller than data_shape %dx%d" % (ih, iw, h, w))
        y0 = int(self._rng.integers(0, ih - h + 1)) if self.rand_crop else (ih - h) // 2
        x0 = int(self._rng.integers(0, iw - w + 1)) if self.rand_crop else (iw - w) // 2
        flip = bool(self.rand_mirror and self._rng.random() < 0.5)
        return img, (y0, x0, flip), float(np.atleast_1d(label)[0])

    def _decode(self, k):
        c, h, w = self.data_shape
        img, (y0, x0, flip), label = self._read(k)
        img = img[y0:y0 + h, x0:x0 + w]
        if flip:
            img = img[:, ::-1]
        return (img[None] if c == 1 else img.transpose(2, 0, 1)), label

    def __next__(self):
        import torch

        from .data import Batch
        if self.pos + self.batch_size > len(self.index):
            raise StopIteration
        if self._file is None:
            self._file = open(self.path, "rb")
        c, h, w = self.data_shape
        labels = np.empty((self.batch_size,), dtype=np.float32)
        if self.device is not None:
            imgs, crops = [], np.empty((self.batch_size, 3), dtype=np.int32)
            for j in range(self.batch_size):
                img, crops[j], labels[j] = self._read(int(self._order[self.pos + j]))
                imgs.append(img if img.ndim == 3 else img[:, :, None])
            if len({im.shape for im in imgs}) == 1:      # one source size per batch (the usual pre-resized .rec): the GPU path
                from . import ops
                self.pos += self.batch_size
                src = torch.from_numpy(np.ascontiguousarray(np.stack(imgs))).to(self.device, non_blocking=True)
                x = ops.crop_mirror_u8(src, torch.from_numpy(crops).to(self.device, non_blocking=True), h, w, self.scale)
                return Batch(["data"], [x], ["softmax_label"], [torch.from_numpy(labels)])
            data = np.empty((self.batch_size, c, h, w), dtype=np.uint8)   # mixed sizes: crop on the host, same draws
            for j, (im, (y0, x0, flip)) in enumerate(zip(imgs, crops)):
                im = im[y0:y0 + h, x0:x0 + w]
                data[j] = (im[:, ::-1] if flip else im).transpose(2, 0, 1)
        else:
            data = np.empty((self.batch_size, c, h, w), dtype=np.uint8)
            for j in range(self.batch_size):
                data[j], labels[j] = self._decode(int(self._order[self.pos + j]))
        self.pos += self.batch_size
        x = torch.from_numpy(data).to(torch.float32)
        if self.scale != 1.0:
            x *= self.scale
        if self.device is not None:
            x = x.to(self.device)
        return Batch(["data"], [x], ["softmax_label"], [torch.from_numpy(labels)])

    next = __next__

    def reset(self):
        self.pos = 0
        self.epoch += 1
        if self.shuffle:
            self._order = self._rng.permutation(len(self.index))


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
