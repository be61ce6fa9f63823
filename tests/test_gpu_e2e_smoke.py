"""Two continual-learning periods of the driver on the shipped DIGINETICA split (one epoch each): period 1 trains from
scratch, selects exemplars by herding; period 2 restores, trains with the distilled loss on train + exemplar rows.  A
plumbing test of ader_amd.main (reference main.py:68-336): flags, log lines, exemplar hand-over, metrics in a sane range."""
import os
import tempfile

import pytest

pytestmark = pytest.mark.gpu


def test_two_periods_of_the_driver():
    from ader_amd import main as M
    with tempfile.TemporaryDirectory() as d:
        args = M.build_parser().parse_args(["--dataset", "DIGINETICA", "--max_periods", "2", "--num_epochs", "1",
                                            "--results_root", d])
        lines = []
        out = M.run(args, log=lambda s="": lines.append(str(s)))
        log_path = os.path.join(d, "DIGINETICA-ADER", "Training_logs.txt")
        assert os.path.isfile(log_path)
        text = open(log_path).read()
    assert "Continue Learning: number of periods is 2." in lines[0]
    assert any(l.startswith("Total saved exemplar:") for l in lines) and "Done." in lines[-1]
    assert "Period 1:" in text and "Period 2:" in text and "Average: (MRR@20:" in text
    per = out["periods"]
    assert len(per) == 2 and per[0]["max_item"] < per[1]["max_item"]
    # one epoch only: far from converged (the full run reaches ~0.50), but well above chance (20 / 18,569 items)
    assert all(0.05 < p["recall20"] < 0.6 and 0.0 < p["mrr20"] < p["recall20"] for p in per)
