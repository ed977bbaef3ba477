"""The RCCL (`nccl` backend) code path under test on a one-GPU box (ref: mutli_gpu_v3.py:117,153-162 — the reference's only multi-device
contract: gradient SUM over devices, rescale_grad = 1/batch).

N > 1 needs hardware the test box does not have; N = 1 does not: with EFM_FORCE_ALLREDUCE=1 a one-rank process group makes every
bucket of the flat gradient go through `dist.all_reduce(async_op=True)` on ProcessGroupNCCL — RCCL's communicator, its internal
streams, the event hand-over between the plan's reduce stream (which finishes a gradient slice) and the collective, and the
`wait()` that orders the optimiser after it.  A one-rank sum is the identity, so the run must leave parameters BIT-IDENTICAL to
the same run with no process group at all; any mis-ordered stream (a bucket reduced before its last slab reduction finished, the
optimiser started before a collective) shows up as a difference."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH, IMAGE, STEPS = 256, 112, 3


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, mode):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "EFM_FORCE_ALLREDUCE", "EFM_TWO_STREAMS", "EFM_REDUCE_STREAM"):
        env.pop(k, None)
    if mode == "rccl":
        env.update(EFM_FORCE_ALLREDUCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = str(tmp_path / ("%s.pt" % mode))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out, str(BATCH), str(IMAGE), str(STEPS), mode],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s run failed (rc %d)\n%s\n%s" % (mode, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return torch.load(out, weights_only=True)


def test_one_rank_rccl_step_is_bit_identical_to_the_step_without_collectives(tmp_path):
    """BASELINE configs[1] (256 images of 3x112x112, the committed tuning table, three-stream backward), 3 SGD steps: with the nccl
    process group every bucket is all-reduced (6 collectives per step, launched late-layers-first from the reduce stream) and the
    parameters, last gradient and every loss equal the collective-free run bit for bit."""
    plain = _run(tmp_path, "plain")
    rccl = _run(tmp_path, "rccl")
    assert rccl["tuned"] and plain["tuned"] and rccl["wino"] >= 10          # the benchmarked kernel selection, not the heuristics
    nb = rccl["nbuckets"]
    assert nb > 1 and rccl["collectives"] == nb * STEPS and plain["collectives"] == 0
    for order in rccl["orders"] + plain["orders"]:
        assert len(order) == nb and order == sorted(order, reverse=True), order   # buckets become final late layers first
    assert torch.equal(rccl["loss"], plain["loss"])
    assert torch.equal(rccl["grad"], plain["grad"])
    assert torch.equal(rccl["flat"], plain["flat"])
    assert torch.isfinite(rccl["flat"]).all() and float(rccl["grad"].abs().max()) > 0


def test_bench_one_rank_over_rccl_reports_the_headline_line(tmp_path):
    """`bench.py` itself with the nccl process group initialised the way its N > 1 branch does (`init_process_group("nccl",
    device_id=...)`, EFM_FORCE_ALLREDUCE): rc 0 and one JSON line — the launch the driver's scaling run makes, minus the other ranks."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", EFM_FORCE_ALLREDUCE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary", "--no-host-loops"], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["parallelism"] == "dp1"
    assert out["collectives_per_step"] >= 2
