"""Committed kernel-selection tables: what `Plan.autotune()` chose on an MI355X for the BASELINE configurations, frozen as data.

Autotune is timing-based, so two runs can pick different (equally valid) kernels for a layer whose candidates are within noise;
a benchmark that re-tunes on every start therefore times a selection no test has seen.  `tools/make_tuning.py` runs autotune once
on the GPU box and writes `<workload>_b<batch>_<image>_<dtype>.json` here; `bench.py` and the full-size parity tests
(tests/test_tuned_gpu.py) both load that file, so the benched selection IS the tested selection.  `EFM_AUTOTUNE=live` re-times."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def path(workload, batch, image, dtype="f32"):
    return os.path.join(HERE, "%s_b%d_%d_%s.json" % (workload, batch, image, dtype))


def load(workload, batch, image, dtype="f32", file=None):
    """-> (table, source) or (None, None) when no table is committed for this configuration."""
    f = file or path(workload, batch, image, dtype)
    if not os.path.exists(f):
        return None, None
    doc = json.load(open(f))
    return doc["table"], os.path.relpath(f, os.path.dirname(os.path.dirname(HERE)))


def save(table, workload, batch, image, dtype="f32", note="", file=None):
    f = file or path(workload, batch, image, dtype)
    with open(f, "w") as fh:
        json.dump({"workload": workload, "batch": batch, "image": image, "dtype": dtype, "note": note, "table": table}, fh, indent=1)
    return f
