"""Host logic of the native step driver (no GPU): a recorded launch sequence is lowered to the C ABI's plan, and the plan patches
exactly the input pointers, the per-step scalar and the dropout keys -- checked through ader_step_plan_peek, which launches nothing.
Reference: main.py:220-256 (one sess.run per step)."""
import ctypes
import struct

import pytest


@pytest.fixture(scope="module")
def lib():
    from ader_amd import build, _lib
    build.build()
    return _lib.load()


def _f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_dispatch_table_matches_the_ctypes_signatures(lib):
    from ader_amd import _lib
    n = 0
    for name, argt in _lib._SIGS.items():
        i = _lib.step_fn_index(name)
        if i >= 0:
            n += 1
            assert lib.ader_step_fn_args(i) == len(argt), name
            assert argt[-1] is _lib.P, name          # the trailing stream argument
    assert n >= 30
    assert _lib.step_fn_index("ader_host_shuffle") == -1 and _lib.step_fn_index("nope") == -1


def test_plan_patches_inputs_scalars_and_dropout_keys(lib):
    from ader_amd import _lib
    from ader_amd.engine import dropout_key
    from ader_amd.engine.common import IN_LR, IN_POS, IN_SEQ, StepF
    from ader_amd.engine.plan import StepPlan
    seed, step, L = 7, 41, 2
    MAIN, SIDE = 0x1000, 0x2000
    seq0, pos0, ws = 0x7F0000001000, 0x7F0000100000, 0x7F0000200000
    drop = _lib.AderDrop()
    drop.key, drop.thr, drop.scale, drop.base, drop.split, drop.base2 = dropout_key(seed, step, 0), 5033165, 1.0 / 0.7, 0, 0xFFFFFFFF, 0
    d = _lib.AderSeqFwd()
    d.seq, d.emb, d.x0 = seq0, ws, ws + 64
    d.B, d.T, d.H, d.V, d.L = 8, 50, 150, 100, L
    d.d_emb = drop
    d.blk[1].d_ffn2.key, d.blk[1].d_ffn2.thr = dropout_key(seed, step, 6), 5033165
    off = _lib.AderDrop()          # a disabled site: never patched
    off.key, off.thr = 0, 0
    items = [
        ("ader_seq_fwd", (ctypes.byref(d), MAIN)),
        (None, (SIDE, MAIN)),
        ("ader_embed_bwd_rows", (seq0 + 400, ws, ws + 8, 8, 50, 150, 100, ctypes.byref(drop), SIDE)),
        ("ader_embed_bwd_rows", (None, ws, ws + 8, -3, 50, 150, 100, ctypes.byref(off), SIDE)),
        ("ader_adam_step", (ws, ws, ws, ws, 1 << 33, StepF(0.25, IN_LR), 0.9, 0.999, 1e-8, None, 0, 150, MAIN)),
        ("ader_build_rowinfo", (pos0, 8, None, None, 0, 90, 0, 0.125, 0.0, 128, ws, ws, ws, ws, MAIN)),
        (None, (MAIN, SIDE)),
    ]
    ranges = [(IN_SEQ, seq0, 8 * 50 * 4), (IN_POS, pos0, 8 * 4)]
    plan = StepPlan(items, {MAIN: 0, SIDE: 1}, ranges, seed, step, 1 + 3 * L)
    assert plan.n_ops == 7 and plan.n_keys == 3 and plan.n_patches == 4
    # a later step with other inputs
    seq1, pos1, step1 = 0x7F1111110000, 0x7F2222220000, 1234567
    plan.inputs[IN_SEQ], plan.inputs[IN_POS], plan.inputs[IN_LR] = seq1, pos1, _f32bits(0.001953125)
    ops, blobs = plan.peek(step1)
    assert [o.kind for o in ops] == [0, 1, 0, 0, 0, 0, 1]
    assert (ops[1].stream, ops[1].other) == (1, 0) and (ops[6].stream, ops[6].other) == (0, 1)
    assert [o.stream for o in ops if o.kind == 0] == [0, 1, 1, 0, 0]
    # descriptor copy: seq pointer patched, everything else as recorded, keys of (seed, step1, site)
    d2 = _lib.AderSeqFwd.from_buffer_copy(blobs[0])
    assert d2.seq == seq1 and d2.emb == ws and d2.x0 == ws + 64 and (d2.B, d2.T, d2.H, d2.V, d2.L) == (8, 50, 150, 100, L)
    assert d2.d_emb.key == dropout_key(seed, step1, 0) and d2.d_emb.thr == 5033165
    assert d2.blk[1].d_ffn2.key == dropout_key(seed, step1, 6)
    assert d2.blk[0].d_attn.key == 0                                        # disabled sites stay untouched
    assert _lib.AderDrop.from_buffer_copy(blobs[1]).key == dropout_key(seed, step1, 0)
    assert _lib.AderDrop.from_buffer_copy(blobs[2]).key == 0
    # argument slots: pointer inside the input range keeps its offset; ints sign-extended; floats as IEEE bits
    assert ops[2].args[0] == seq1 + 400 and ops[2].args[1] == ws
    assert ops[3].args[0] == 0 and ops[3].args[3] == (-3) & 0xFFFFFFFFFFFFFFFF
    assert ops[4].args[4] == 1 << 33 and ops[4].args[5] == _f32bits(0.001953125) and ops[4].args[6] == _f32bits(0.9)
    assert ops[5].args[0] == pos1 and ops[5].args[7] == _f32bits(0.125)
    # the original descriptors are not touched by the plan
    assert d.seq == seq0 and d.d_emb.key == dropout_key(seed, step, 0)
    plan.destroy()


def test_unplannable_sequences_are_refused(lib):
    from ader_amd.engine.plan import PlanError, StepPlan
    with pytest.raises(PlanError):
        StepPlan([("ader_row_lse", (1, 2, 3, 4, 5, 6, 0x10))], {0x10: 0}, [], 0, 0, 7)
    with pytest.raises(PlanError):          # a launch on a stream that is not one of the two lanes
        StepPlan([("ader_fill", (0x7F0000000000, 4, 0.0, 0x30))], {0x10: 0}, [], 0, 0, 7)
