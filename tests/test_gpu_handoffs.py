"""Every in-launch hand-off between workgroups, hammered on a cold process (tests/stress_handoffs.py in a fresh subprocess: its first
launches are the launches under test): 2,000 pack plans against the numpy restatement of the packing rule, 200 herding launches
against oracle/herding_ref, and a four-engine cold / warm bitwise comparison of a packed distilled step of the YOOCHOOSE shape.
Background: a role-split table update (k_tabp, removed in round 5) wrote wrong vectors "on a cold process in one run out of several"
-- a hand-off that was correct by cache behaviour, not by the memory model."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_handoffs_survive_a_cold_process():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_handoffs.py")
    r = subprocess.run([sys.executable, script, "2000", "200"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and "handoffs ok: 2000 pack plans, 200 herding launches, 4 engines bit-identical" in out, out[-3000:]
