"""Round trips of the MXNet on-disk formats (parity unpinned: no MXNet-written file exists here — see mxio.py)."""
import struct

import numpy as np
import torch

from improving_face_recognition_performance_using_triplet_loss_amd import mxio


def test_params_round_trip_and_layout(tmp_path):
    p = {"conv1_weight": np.random.default_rng(0).normal(size=(99, 3, 5, 5)).astype(np.float32),
         "conv1_bias": np.zeros(99, np.float32), "fc1_weight": np.arange(12, dtype=np.float32).reshape(3, 4)}
    path = str(tmp_path / "efm_res-0000.params")
    mxio.save_params(path, p)
    raw = open(path, "rb").read()
    assert struct.unpack_from("<QQQ", raw, 0) == (0x112, 0, 3)
    assert struct.unpack_from("<Ii", raw, 24) == (0xF993FAC9, 0)                  # NDArray V2, dense
    assert struct.unpack_from("<I4q", raw, 32) == (4, 99, 3, 5, 5)
    q = mxio.load_params(path)
    assert list(q) == list(p) and all(np.array_equal(q[k], p[k]) and q[k].dtype == np.float32 for k in p)
    # Module-style names
    mxio.save_params(path, {"arg:fc1_weight": p["fc1_weight"], "aux:bn_moving_mean": np.ones(4, np.float32)})
    assert sorted(mxio.load_params(path)) == ["bn_moving_mean", "fc1_weight"]
    assert sorted(mxio.load_params(path, strip_prefix=False)) == ["arg:fc1_weight", "aux:bn_moving_mean"]


def test_recordio_round_trip_and_iterator(tmp_path):
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, size=(20, 18, 3), dtype=np.uint8) for _ in range(5)]
    payloads = [mxio.pack_img(float(i % 2), i, im) for i, im in enumerate(imgs)]
    path = str(tmp_path / "train.rec")
    mxio.write_records(path, payloads)
    got = list(mxio.read_records(path))
    assert got == payloads
    label, idx, im = mxio.unpack_img(got[3])
    assert label == 1.0 and idx == 3 and np.array_equal(im, imgs[3])                 # PNG is lossless
    it = mxio.ImageRecordIter(path, (3, 16, 16), batch_size=2, scale=1.0 / 255)
    batches = list(it)
    assert len(batches) == 2 and tuple(batches[0].data[0].shape) == (2, 3, 16, 16)
    want = imgs[0][2:18, 1:17].transpose(2, 0, 1).astype(np.float32) / 255           # centre crop
    assert np.allclose(batches[0].data[0][0].numpy(), want)
    assert batches[1].label[0].tolist() == [0.0, 1.0]
    gray = mxio.ImageRecordIter(path, (1, 16, 16), batch_size=5, rand_crop=True, rand_mirror=True, shuffle=True, seed=3)
    b = next(iter(gray))
    assert tuple(b.data[0].shape) == (5, 1, 16, 16) and float(b.data[0].max()) <= 255.0


def test_lst_reader(tmp_path):
    p = tmp_path / "train.lst"
    p.write_text("0\t3.000000\tid3/a.jpg\n1\t7.000000\tid7/b.jpg\n")
    rows = mxio.read_lst(str(p))
    assert rows == [(0, [3.0], "id3/a.jpg"), (1, [7.0], "id7/b.jpg")]


def test_symbol_json_round_trip_efm29_and_mfm2(tmp_path):
    """EFM-29 (both MFM operand orders, residual adds, the id head) and LightCNN-9 (MFM2) survive save_symbol -> load_symbol:
    same parameter table, same lowered plan, the MFM idiom folds back into one node per occurrence."""
    import json

    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, mxio
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    for name, outs in (("efm", efm_symbol.get_net(10)), ("l9", efm_symbol.lightcnn9_embedding_net())):
        path = str(tmp_path / (name + "-symbol.json"))
        mxio.save_symbol(path, outs)
        doc = json.load(open(path))
        ops_used = {n["op"] for n in doc["nodes"]}
        assert {"Convolution", "SliceChannel", "_maximum", "Pooling", "FullyConnected"} <= ops_used
        assert len(doc["node_row_ptr"]) == len(doc["nodes"]) + 1 and all(doc["nodes"][i]["op"] == "null" for i in doc["arg_nodes"])
        back = mxio.load_symbol(path)
        p0 = Plan(outs, (2, 3, 112, 112), device="cpu")
        p1 = Plan(back, (2, 3, 112, 112), device="cpu")
        assert list(p0.params) == list(p1.params)
        assert [(s.op, s.shape, bool(s.epi), s.residual is not None) for s in p0.steps] == \
               [(s.op, s.shape, bool(s.epi), s.residual is not None) for s in p1.steps]
        assert [s.epi for s in p0.steps] == [s.epi for s in p1.steps]           # ways / order / pool of every fused epilogue
        assert p0.flops_fwd == p1.flops_fwd
    # internals by name, as final_efm.py:207-210 / extract_feacture_v2.py:49-50 pick them
    path = str(tmp_path / "efm-symbol.json")
    heads = mxio.load_symbol(path, outputs=["fc2_output"])
    assert heads[0].op == "fc" and heads[0].name == "fc2"


def test_params_file_byte_level_known_answer(tmp_path):
    """A .params file assembled BY HAND from MXNet 1.x's published layout (src/ndarray/ndarray.cc: NDArray::Save list header
    kMXAPINDArrayListMagic = 0x112, reserved 0, count; per array NDARRAY_V2_MAGIC 0xF993FAC9, storage type 0, TShape (uint32 ndim +
    int64 dims), Context {dev_type int32 = 1 cpu, dev_id int32}, type flag int32 (0 = float32), raw data; then the name list with
    uint64 lengths) — not produced by this package's writer — must load, and the writer must reproduce it byte for byte.
    Spec-pinned, not MXNet-pinned: no MXNet-written file exists in the reference or this image."""
    blob = bytes.fromhex(
        "1201000000000000" "0000000000000000" "0200000000000000"                 # list magic 0x112, reserved, 2 arrays
        "c9fa93f9" "00000000" "02000000" "0200000000000000" "0300000000000000"   # V2 magic, stype 0, ndim 2, shape (2, 3)
        "01000000" "00000000" "00000000"                                         # Context cpu(0), type flag 0 = float32
        "0000803f" "00000040" "00004040" "00008040" "0000a040" "0000c040"        # 1 2 3 4 5 6
        "c9fa93f9" "00000000" "01000000" "0200000000000000"                      # V2 magic, stype 0, ndim 1, shape (2,)
        "01000000" "00000000" "00000000"
        "000000bf" "00002041"                                                    # -0.5, 10
        "0200000000000000"                                                       # 2 names
        "0d00000000000000") + b"arg:fc_weight" + bytes.fromhex("0b00000000000000") + b"aux:bn_mean"
    path = tmp_path / "hand-0001.params"
    path.write_bytes(blob)
    got = mxio.load_params(str(path))
    assert list(got) == ["fc_weight", "bn_mean"]
    assert got["fc_weight"].dtype == np.float32 and np.array_equal(got["fc_weight"], [[1, 2, 3], [4, 5, 6]])
    assert np.array_equal(got["bn_mean"], [-0.5, 10.0])
    raw = mxio.load_params(str(path), strip_prefix=False)
    assert list(raw) == ["arg:fc_weight", "aux:bn_mean"]
    out = tmp_path / "rewritten.params"
    mxio.save_params(str(out), raw)
    assert out.read_bytes() == blob


def test_recordio_byte_level_known_answer(tmp_path):
    """One RecordIO record assembled by hand from dmlc-core's recordio.h (uint32 kMagic 0xced7230a, uint32 lrecord = cflag << 29 |
    length, payload, zero padding to 4 bytes) holding an mx.recordio IRHeader ('IfQQ': flag 0, label 7.0, id 42, id2 0) and a
    payload tail; plus a record split into two parts at an embedded magic word (cflag 1 then 3), which readers must re-join with
    the magic re-inserted."""
    head = bytes.fromhex("00000000" "0000e040" "2a00000000000000" "0000000000000000")          # IRHeader(0, 7.0, 42, 0)
    rec1 = bytes.fromhex("0a23d7ce" "1b000000") + head + b"abc" + b"\x00"                       # length 27 -> 1 pad byte
    magic = bytes.fromhex("0a23d7ce")
    part_a, part_b = b"1234", b"5678xy"
    rec2 = magic + (1 << 29 | len(part_a)).to_bytes(4, "little") + part_a + magic + (3 << 29 | len(part_b)).to_bytes(4, "little") + part_b + b"\x00\x00"
    path = tmp_path / "hand.rec"
    path.write_bytes(rec1 + rec2)
    recs = list(mxio.read_records(str(path)))
    assert recs == [head + b"abc", part_a + magic + part_b]
    import struct
    assert struct.unpack_from("<IfQQ", recs[0]) == (0, 7.0, 42, 0)
    idx = mxio.index_records(str(path))
    assert idx == [(0, 27), (36, -1)]
    with open(path, "rb") as f:
        assert [mxio.read_record_at(f, *e) for e in idx] == recs
    # the writer emits the single-part form of the first record byte for byte
    out = tmp_path / "w.rec"
    mxio.write_records(str(out), [head + b"abc"])
    assert out.read_bytes() == rec1


def test_image_record_iter_streams_and_shards(tmp_path):
    """ImageRecordIter keeps an index, not pixels; `part_index` / `num_parts` (MXNet's parameters) give data-parallel ranks disjoint
    shares of EQUAL size (n // num_parts records each: ranks that all-reduce per batch must step the same number of times);
    crops / mirrors / order are redrawn every epoch."""
    rng = np.random.default_rng(2)
    imgs = [rng.integers(0, 256, size=(20, 20), dtype=np.uint8) for _ in range(10)]
    path = str(tmp_path / "d.rec")
    mxio.write_records(path, [mxio.pack_img(float(i), i, im) for i, im in enumerate(imgs)])
    parts = [mxio.ImageRecordIter(path, (1, 20, 20), batch_size=1, part_index=k, num_parts=3) for k in range(3)]
    seen = [[int(b.label[0][0]) for b in p] for p in parts]
    assert sum(seen, []) == list(range(9)) and [len(s) for s in seen] == [3, 3, 3]
    assert all(p.num_total == 10 for p in parts) and not hasattr(parts[0], "data_arr")
    it = mxio.ImageRecordIter(path, (1, 16, 16), batch_size=5, rand_crop=True, rand_mirror=True, shuffle=True, seed=1)
    e1 = [b.label[0].tolist() for b in it]
    e2 = [b.label[0].tolist() for b in it]
    assert sorted(sum(e1, [])) == sorted(sum(e2, [])) == [float(i) for i in range(10)] and e1 != e2
    full = next(iter(mxio.ImageRecordIter(path, (1, 20, 20), batch_size=10, scale=1.0 / 255)))
    assert np.allclose(full.data[0][3, 0].numpy() * 255, imgs[3])


def test_image_record_iter_equal_batches_per_rank(tmp_path):
    """1025 records over 2 ranks at local batch 171 (the advisor's case: 512 vs 513 records gave 2 vs 3 batches, and the rank with
    the extra batch would block forever in its all-reduce): every rank yields the same number of batches."""
    img = np.zeros((4, 4), dtype=np.uint8)
    path = str(tmp_path / "n.rec")
    mxio.write_records(path, [mxio.pack_img(float(i), i, img) for i in range(1025 + 2)])
    for n_parts, bs in ((2, 171), (3, 114), (8, 64)):
        its = [mxio.ImageRecordIter(path, (1, 4, 4), batch_size=bs, part_index=k, num_parts=n_parts, preprocess_threads=0) for k in range(n_parts)]
        counts = [sum(1 for _ in it) for it in its]
        assert len(set(counts)) == 1 and counts[0] == (1027 // n_parts) // bs, (n_parts, bs, counts)
        assert len({len(it) for it in its}) == 1


def test_image_record_iter_threaded_prefetch_is_bit_identical_to_synchronous(tmp_path):
    """The decode pool + one-batch-ahead producer (MXNet's `preprocess_threads` / `prefetch_buffer`, ref: train_efm.py:179-181 is a
    threaded C++ iterator) must emit exactly the batches of the synchronous path: decoding is parallel, the crop / mirror draws
    stay sequential in record order.  Also across epochs (reshuffle) and across a reset() in the middle of an epoch, where the
    producer has already drawn for batches nobody consumed."""
    rng = np.random.default_rng(5)
    path = str(tmp_path / "t.rec")
    mxio.write_records(path, [mxio.pack_img(float(i), i, rng.integers(0, 256, size=(24, 28), dtype=np.uint8), fmt="PNG") for i in range(70)])
    kw = dict(batch_size=8, scale=1.0 / 255, rand_crop=True, rand_mirror=True, shuffle=True, seed=9)
    sync = mxio.ImageRecordIter(path, (1, 16, 16), preprocess_threads=0, **kw)
    thr = mxio.ImageRecordIter(path, (1, 16, 16), preprocess_threads=4, prefetch_buffer=3, **kw)

    def take(it, n):
        out = []
        for b in it:
            out.append((b.data[0].clone(), b.label[0].clone()))
            if len(out) == n:
                break
        return out

    for n in (100, 3, 100):      # a whole epoch, an epoch abandoned after 3 batches, another whole epoch
        a, b = take(sync, n), take(thr, n)
        assert len(a) == len(b) == min(n, 8)
        for (xa, la), (xb, lb) in zip(a, b):
            assert torch.equal(xa, xb) and torch.equal(la, lb)
    thr.close()
