"""Round trips of the MXNet on-disk formats (parity unpinned: no MXNet-written file exists here — see mxio.py)."""
import struct

import numpy as np

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
