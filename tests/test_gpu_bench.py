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
    out = json.loads(lines[0])
    out["_stderr"] = r.stderr
    return out


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
    # the workloads the reference trains, and the ADER-mode step at the headline catalog, each with its own roofline block
    rs = d["real_shapes"]
    assert sorted(rs) == ["ader128", "cfgD", "cfgY"]
    for nm, row in rs.items():
        assert "failed" not in row, row
        rf = row["roofline"]
        assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["floor_ms"] / row["ms_per_step"]) < 2e-3
        assert abs(sum(rf["floor_parts_ms"].values()) - rf["floor_ms"]) < 1e-4 and rf["logit_flops_executed"] > rf["logit_flops_credited"]
        assert row["host_enqueue_ms"] > 0 and row["driver"].startswith("native launch plan")
    assert rs["cfgY"]["session_tiles"] == "packed" and rs["ader128"]["session_tiles"] != "packed"


@pytest.mark.parametrize("mode", ["catalog", "replicated"])
def test_gpus_2_starts_its_own_ranks(mode):
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "2", "--reps", "1", "--items", "20000", "--no-cpu-baseline", "--dp-mode", mode,
              "--sustained-steps", "0"],
             env={"ADER_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["global_batch"] == 1024
    assert d["comm"]["exchange_bytes_per_step"] > 0 and d["comm"]["comm_ms"] >= 0
    assert (d["comm"]["dp_mode"] == "catalog") == (mode == "catalog")
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # first-contact insurance: the guarded first step of BOTH schemes (the value's and the other leg's) announced its collectives,
    # rank 0's lists are on the line, every rank printed its own, and the line says which scheme `value` is
    other = "replicated" if mode == "catalog" else "catalog"
    c = d["comm"]
    assert d["config"]["dp_mode_of_value"] == mode and mode in c["value_is"]
    assert c["other_leg"]["dp_mode"] == other and c["%s_ms_per_step" % other] > 0
    assert len(c["collectives"][mode]) >= 3 and len(c["collectives"][other]) >= 3
    assert any(x["kind"].startswith("all_reduce") for x in c["collectives"]["replicated"])
    assert any(x["kind"] == "all_gather" for x in c["collectives"]["catalog"])
    for r_ in (0, 1):
        assert "[rank %d] dp_mode=%s world=2" % (r_, mode) in d["_stderr"] and "[rank %d] collective  0:" % r_ in d["_stderr"]
    # BASELINE configs[3] on the multi-GPU line: the YOOCHOOSE ADER step shape at global batch 512 x ranks, both schemes
    y = d["real_shapes_dp"]["cfgY"]
    assert y["global_batch"] == 1024 and y["global_exemplar_rows"] == 204 and sorted(y["schemes"]) == ["catalog", "replicated"]
    for nm, sc in y["schemes"].items():
        assert sc["ms_per_step"] > 0 and abs(sc["sessions_per_s"] - 1024 / (sc["ms_per_step"] * 1e-3)) < 1e-6 * sc["sessions_per_s"]
        assert sc["collectives_per_step"] == len(sc["collectives"]) >= 2 and sc["exchange_bytes_per_step"] > 0
        assert "[rank 1] real_shapes_dp cfgY dp_mode=%s world=2" % nm in d["_stderr"]
    assert y["schemes"]["replicated"]["collectives_per_step"] < y["schemes"]["catalog"]["collectives_per_step"]
    assert abs(y["schemes"]["replicated"]["final_loss"] - y["schemes"]["catalog"]["final_loss"]) < 2e-3
