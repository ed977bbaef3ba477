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
