"""The C ABI used from plain C++ (no Python / torch in the consumer): builds tests/c_abi/efm_abi_example.cpp against include/efm_hip.h
and libefm_hip.so with hipcc and runs it on the GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "improving_face_recognition_performance_using_triplet_loss_amd")


def test_cpp_consumer_of_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(os.path.join(PKG, "libefm_hip.so")), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    exe = str(tmp_path / "efm_abi_example")
    r = subprocess.run([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "efm_abi_example.cpp"),
                        "-L", PKG, "-lefm_hip", "-Wl,-rpath," + PKG, "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "efm C ABI example: OK" in r.stdout


def test_feature_hpp_shaped_consumer_through_mxnet_symbols(tmp_path):
    """tests/c_abi/feature_consumer.cpp — the MXNet call sequence of Feature.hpp:163-205, built with plain g++ against
    include/c_predict_api.h only — extracts the 342-d feature of one 1x128x128 face through MXPredCreatePartialOut(..., "concat29") /
    MXPredSetInput / MXPredForward / MXPredGetOutputShape / MXPredGetOutput; the result equals the training plan's forward bit for bit."""
    import numpy as np
    import torch
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, mxio, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    S = 128
    data = efm_symbol.G.Variable("data")
    feat_sym, _ = efm_symbol.efm_feature(data)
    plan = Plan([feat_sym], (1, 1, S, S))
    flat = plan.new_flat()
    plan.init_xavier(flat, 3)
    x = synth.images(1, 1, S, 9)
    (feat,) = plan.forward(x, flat, train=False)
    params = {("arg:" + k): v.cpu().numpy() for k, v in plan.export_params(flat).items()}
    params["arg:fc1_weight"] = params["arg:fc1_weight"].reshape(513, -1)
    mxio.save_params(str(tmp_path / "EFM_RES.params"), params)
    x.cpu().numpy().astype(np.float32).tofile(str(tmp_path / "in.f32"))
    exe = str(tmp_path / "feature_consumer")
    gxx = shutil.which("g++")
    r = subprocess.run([gxx, "-std=c++11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "feature_consumer.cpp"),
                        "-L", PKG, "-lefm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    for layer in ("concat29", "concat29_output"):
        r = subprocess.run([exe, str(tmp_path / "EFM_RES.params"), str(S), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), layer],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "feature_consumer: OK" in r.stdout, r.stdout + r.stderr
        got = np.fromfile(str(tmp_path / "out.f32"), dtype=np.float32)
        assert np.array_equal(got, feat[0, :342].cpu().numpy())
    r = subprocess.run([exe, str(tmp_path / "EFM_RES.params"), str(S), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "fc2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "unknown output node" in r.stdout
