"""The data-parallel path with more than one rank actually running (ref: mutli_gpu_v3.py:117,153-162)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    return dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")


def test_two_rank_step_equals_single_process_double_batch(tmp_path):
    """Two ranks x 8 images (own anchors, positives, LOCAL negatives) for two SGD steps == one process stepping the same 16 images
    with the same triplets: the gradient exchange is a plain SUM launched bucket by bucket from backward (late layers first), the
    mean is the optimiser's rescale = 1/(global anchors).  Both ranks end with bit-identical parameters; the 2-rank and the
    1-process all-reduced GRADIENTS agree to 1e-4 (the updates to that plus their fp32 representation floor)."""
    batch, image, steps = 8, 32, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), str(batch), str(image), str(steps)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ranks = [torch.load(str(tmp_path / ("rank%d.pt" % k)), weights_only=True) for k in range(2)]
    assert torch.equal(ranks[0]["flat"], ranks[1]["flat"]) and torch.equal(ranks[0]["grad"], ranks[1]["grad"])
    for k in range(2):
        assert ranks[k]["order"] == sorted(ranks[k]["order"], reverse=True) and len(ranks[k]["order"]) == ranks[k]["nbuckets"] > 1

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dp_worker import shard_inputs
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    (x0, n0), (x1, n1) = shard_inputs(0, batch, image), shard_inputs(1, batch, image)
    h = batch // 2
    x = torch.cat([x0[:h], x1[:h], x0[h:], x1[h:]])                     # [anchors r0, anchors r1 ; positives r0, positives r1]
    neg = torch.cat([n0, n1 + h]).to(torch.int32)
    big = TripletTrainer(2 * batch, image=image, seed=3, optimizer="sgd", lr=0.05, wd=1e-5)
    init = big.flat.clone()
    losses, errs, gerrs = [], [], []
    for k in range(steps):
        losses.append(big.step(x, neg).clone())
        upd_big = (big.flat - init).cpu()
        upd_dp = ranks[0]["flats"][k] - init.cpu()
        errs.append(float((upd_dp - upd_big).abs().max() / upd_big.abs().max()))
        g_big, g_dp = big.grad.cpu().double(), ranks[0]["grads"][k].double()
        gerrs.append(float((g_dp - g_big).abs().max() / g_big.abs().max()))
    assert torch.equal(losses[0].cpu(), torch.cat([ranks[0]["loss"][0], ranks[1]["loss"][0]]))   # step 1 forward: identical weights
    # what fp32 parameters can resolve of an update: `flat - init` is a difference of weights of magnitude |w| whose update is
    # lr * g / B ~ 1e-5 — half an ulp of the largest weight, relative to the largest update, is the floor of the update comparison
    floor = float(torch.finfo(torch.float32).eps * init.abs().max() / (big.flat - init).abs().max())
    print("2-rank vs single-process: all-reduced gradient rel err step 1 %.2e, step 2 %.2e; update rel err %.2e / %.2e (fp32 "
          "representation floor of an update %.1e)" % (gerrs[0], gerrs[1], errs[0], errs[1], floor))
    # the exchanged quantity itself — the SUM over ranks of the flat gradient — against the single process's gradient of the doubled
    # batch: summation order only (a + b across ranks vs one split-K sweep).  A bucket skipped, summed twice or left unscaled is O(1).
    assert gerrs[0] < 1e-4, gerrs
    # the update adds the cancellation of `flat - init` to that (measured 4e-4 where the gradient agrees to ~1e-6)
    assert errs[0] < 1e-4 + 4 * floor, (errs, floor)
    # two steps: last-bit weight differences also move a few max/min/pool routes of the second forward
    assert gerrs[1] < 2e-2 and errs[1] < 2e-3 + 4 * floor, (gerrs, errs)


def test_bench_gpus2_self_starts_and_reports_one_line(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment starts its own ranks as a child process (both on the one card of this box,
    gloo instead of RCCL) and rank 0 prints the single JSON line with n_gpus = 2, whole-job throughput, scaling 'weak'."""
    env = dict(_env(), EFM_BENCH_ONE_DEVICE="1", EFM_DIST_BACKEND="gloo", EFM_AUTOTUNE="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--image", "32"], env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "triplets/s" and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2"
