"""Build driver: compiles csrc/*.hip for gfx950 into one in-tree shared library.

`python -m improving_face_recognition_performance_using_triplet_loss_amd.build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libefm_hip.so")
SOURCES = ["efm_api.hip", "efm_conv.hip", "efm_winograd.hip", "efm_wino_wgrad.hip", "efm_convb_wgrad.hip", "efm_elementwise.hip", "efm_head.hip", "efm_predict.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-source extras: the SLP vectoriser packs the Winograd weight gradient's operand transforms into v_pk_*_f32, which is slower beside MFMAs
EXTRA = {"efm_wino_wgrad.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA.items()))).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [_hipcc(), *FLAGS, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, r.stderr


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, "efm_common.h"),
        os.path.join(HERE, "..", "include", "efm_hip.h"),
    ]
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest(deps)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print("[efm build] up to date:", LIB)
        return LIB
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        results = list(ex.map(_compile, SOURCES))
    for _, warn in results:
        if warn.strip() and verbose:
            print(warn, file=sys.stderr)
    objs = [o for o, _ in results]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("[efm build] built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
