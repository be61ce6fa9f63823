"""Accuracy parity of the whole continual-learning loop on the shipped splits (reference main.py:158-335; the metric of
BASELINE.json is "...; Recall@20 parity").  These are the only pins of the floating-point path that a reference artefact
holds: TensorFlow cannot run here and the reference has no tests, so the model math is checked end to end against the
accuracy table of the reference's poster (ADER Recall@20 49.92 / 50.09 %, MRR@20 17.23 / 17.29 %).

  * BASELINE configs[1]: DIGINETICA, ADER default flags (30k exemplars, herding, lambda 0.8, batch 256, 2 blocks), bf16 logit
    operands: all 16 periods, the `Average:` line must fall inside the poster band +- 0.5 point.
  * BASELINE configs[2]: YOOCHOOSE `--lambda_=1.0 --batch_size=512 --test_batch=64` (reference README.md:77).  The reference
    publishes no number for it: the expected value is a REGRESSION PIN measured with this build (profiles/e2e_r1/
    YOOCHOOSE-ADER_bf16_r1j.txt: Recall@20 72.31 %, MRR@20 36.72 %), +- 0.3 point.

Data order follows the reference's RNG streams only until early stopping first differs, so the parity is
statistical, not bitwise."""
import os
import tempfile

import pytest

pytestmark = pytest.mark.gpu


def _run(argv):
    from ader_amd import main as M
    with tempfile.TemporaryDirectory() as d:
        args = M.build_parser().parse_args(argv + ["--results_root", d])
        lines = []
        out = M.run(args, log=lambda s="": lines.append(str(s)))
        name = args.dataset + "-" + args.save_dir
        text = open(os.path.join(d, name, "Training_logs.txt")).read()
    assert "Done." in lines[-1] and "Average: (MRR@20:" in text
    return out


def test_diginetica_ader_bf16_inside_the_poster_band():
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "bf16"])
    avg, per = out["average"], out["periods"]
    assert len(per) == 16
    assert per[0]["max_item"] == 18569 and per[-1]["max_item"] == 43105
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    # poster: ADER-equal 49.92 / 17.23, ADER-fix 50.09 / 17.29 (the default run is between the two variants); +- 0.5 point
    assert 49.4 <= r20 <= 50.6, ("Recall@20 outside the poster band", r20)
    assert 16.9 <= m20 <= 17.7, ("MRR@20 outside the poster band", m20)
    assert 36.7 <= 100.0 * avg["recall10"] <= 37.9 and 15.85 <= 100.0 * avg["mrr10"] <= 16.9   # poster 37.21-37.41 / 16.35-16.41


def test_yoochoose_ader_regression_pin():
    out = _run(["--dataset", "YOOCHOOSE", "--lambda_", "1.0", "--batch_size", "512", "--test_batch", "64", "--logits_dtype",
                "bf16"])
    avg, per = out["average"], out["periods"]
    assert len(per) == 16
    assert per[0]["max_item"] == 12885 and per[-1]["max_item"] == 25750
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    assert abs(r20 - 72.31) <= 0.3, ("Recall@20 moved from the committed pin 72.31", r20)
    assert abs(m20 - 36.72) <= 0.3, ("MRR@20 moved from the committed pin 36.72", m20)


def test_diginetica_ader_float32_grade_inside_the_poster_band():
    """The same 16-period run with the float32-grade logit kernels (logits_dtype = x3: every product as three bf16 MFMAs on hi/lo
    splits, the reference's fp32 arithmetic of ADER.py:91-93): the band of the bf16 run must hold as well."""
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "x3"])
    avg = out["average"]
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    assert 49.4 <= r20 <= 50.6, ("Recall@20 outside the poster band", r20)
    assert 16.9 <= m20 <= 17.7, ("MRR@20 outside the poster band", m20)
