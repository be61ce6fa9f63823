"""Ewc.compute_fisher(dry_run=True) must consume the `random` stream exactly like the real pass (and like the reference: the
Sampler's construction shuffle plus the wrap-around reshuffle of its batch_num()-th draw, util.py:149,232-237; EWC.py:139-141):
main.py's per-epoch dry run keeps every later train / exemplar shuffle where the reference has it.  CPU only."""
import random
import types

from ader_amd import model as M


class _FakeEngine:
    def compute_fisher(self, seq, pos, max_item):
        self.n = len(seq)


def _ewc():
    e = M.Ewc.__new__(M.Ewc)
    e.args = types.SimpleNamespace(maxlen=10)
    e.engine = _FakeEngine()
    return e


def test_dry_run_consumes_the_random_stream_like_the_real_pass():
    data = [[1 + (i * 7 + j) % 23 for j in range(2 + i % 5)] for i in range(37)]
    states = []
    for dry in (True, False):
        random.seed(123)
        _ewc().compute_fisher(None, data, 8, 23, dry_run=dry)
        states.append(random.getstate())
    assert states[0] == states[1]
