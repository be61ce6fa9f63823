"""bench.py contract checks on the GPU box: the N = 1 line carries the roofline / reps blocks, and `--gpus 2` starts its own ranks
(two ranks share cuda:0 with ADER_DIST_BACKEND=gloo: the 8-GPU run is the driver's) in both data-parallel modes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run(["--steps", "3", "--warmup", "3", "--reps", "3", "--items", "30000", "--no-cpu-baseline", "--no-herding",
              "--sustained-steps", "40"])
    assert d["n_gpus"] == 1 and d["dtype"].startswith("bf16x3") and d["unit"] == "sessions/s" and d["scaling"] == "weak"
    assert d["reps"] == 3 and d["reps_ms"]["min"] <= d["reps_ms"]["median"] <= d["reps_ms"]["max"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["logit_gemm"]["flops_executed"] == 3 * r["logit_gemm"]["flops_credited"]
    assert d["value_bf16"] is not None and d["companion"]["logits"] == "bf16"
    assert "REDUCED SIZE" in d["config"]["workload"]
    su = d["sustained"]
    assert su["steps"] == 40 and abs(su["value"] - 512 / (su["ms_per_step"] * 1e-3)) < 1e-6 * su["value"]
    assert r["frac_of_achievable"] is None or r["frac_of_achievable"] > r["frac"]


@pytest.mark.parametrize("mode", ["catalog", "replicated"])
def test_gpus_2_starts_its_own_ranks(mode):
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "2", "--reps", "1", "--items", "20000", "--no-cpu-baseline", "--dp-mode", mode,
              "--sustained-steps", "0"],
             env={"ADER_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["global_batch"] == 1024
    assert d["comm"]["exchange_bytes_per_step"] > 0 and d["comm"]["comm_ms"] >= 0
    assert (d["comm"]["dp_mode"] == "catalog") == (mode == "catalog")
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
