"""The native predictor (c_predict_api call shape, ref: Feature.hpp:163-205) returns the same 342-d feature as the training
plan, from the bytes of an MXNet-format .params file."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_predictor_matches_plan_forward(tmp_path):
    from improving_face_recognition_performance_using_triplet_loss_amd import _lib, efm_symbol, mxio, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    batch, image = 3, 112
    plan = Plan(efm_symbol.embedding_net(), (batch, 3, image, image))
    flat = plan.new_flat()
    plan.init_xavier(flat, 11)
    x = synth.images(batch, 3, image, 5)
    _, feat = plan.forward(x, flat, train=False)
    params = {("arg:" + k): v.cpu().numpy() for k, v in plan.export_params(flat).items() if k != "head_weight"}
    params["arg:fc1_weight"] = params["arg:fc1_weight"].reshape(513, -1)  # MXNet stores FullyConnected weights 2-D
    path = str(tmp_path / "EFM_RES.params")
    mxio.save_params(path, params)
    blob = open(path, "rb").read()

    lib = _lib.load()
    keys = (ctypes.c_char_p * 1)(b"data")
    indptr = (ctypes.c_uint32 * 2)(0, 4)
    shape = (ctypes.c_uint32 * 4)(batch, 3, image, image)
    h = ctypes.c_void_p()
    _lib.check(lib.efm_pred_create(None, blob, len(blob), 0, 1, keys, indptr, shape, ctypes.byref(h)), "efm_pred_create")
    xin = np.ascontiguousarray(x.cpu().numpy())
    _lib.check(lib.efm_pred_set_input(h, b"data", xin.ctypes.data_as(ctypes.c_void_p), xin.size), "efm_pred_set_input")
    _lib.check(lib.efm_pred_forward(h), "efm_pred_forward")
    sd, nd = ctypes.POINTER(ctypes.c_uint32)(), ctypes.c_uint32()
    _lib.check(lib.efm_pred_get_output_shape(h, 0, ctypes.byref(sd), ctypes.byref(nd)), "efm_pred_get_output_shape")
    assert nd.value == 2 and (sd[0], sd[1]) == (batch, 342)
    out = np.empty((batch, 342), np.float32)
    _lib.check(lib.efm_pred_get_output(h, 0, out.ctypes.data_as(ctypes.c_void_p), out.size), "efm_pred_get_output")
    assert np.array_equal(out, feat[:, :342].cpu().numpy())  # same kernels, same bits
    # error convention: wrong size is refused, not a crash
    assert lib.efm_pred_set_input(h, b"data", xin.ctypes.data_as(ctypes.c_void_p), 7) == -1
    _lib.check(lib.efm_pred_free(h), "efm_pred_free")
    # a blob that is not an NDArray list
    assert lib.efm_pred_create(None, b"nonsense" * 8, 64, 0, 1, keys, indptr, shape, ctypes.byref(h)) == -1
