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
import json
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


def _curves():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_svg_curves.json")))["curves"]


def _against_figure(out, dataset, method, tol_avg, tol_period):
    """Average and per-period test metrics of a 16-period run against the curve of the reference's published figure (results.svg,
    recovered by tests/golden/make_results_curves.py).  Per-period numbers carry the run's own noise (early stopping, RNG streams
    that diverge from the reference's as soon as an epoch count differs: SURVEY appendix), so they are held to a mean absolute
    deviation; the 16-period averages to tol_avg."""
    ref = _curves()[dataset][method]
    per = out["periods"]
    assert len(per) == 16
    res = {}
    for key in ("recall20", "mrr20"):
        mine = [100.0 * p_[key] for p_ in per]
        avg, avg_ref = sum(mine) / 16, sum(ref[key]) / 16
        mad = sum(abs(a - b) for a, b in zip(mine, ref[key])) / 16
        res[key] = (avg, avg_ref, mad)
        print("%s %s %s: average %.2f (figure %.2f, delta %+.2f), per-period mean |delta| %.2f"
              % (dataset, method, key, avg, avg_ref, avg - avg_ref, mad))
    for key, (avg, avg_ref, mad) in res.items():
        assert abs(avg - avg_ref) <= tol_avg, (dataset, method, key, avg, avg_ref)
        assert mad <= tol_period, (dataset, method, key, "per-period deviation", mad)
    return res


def test_yoochoose_ader_matches_the_published_curve():
    """BASELINE configs[2]: YOOCHOOSE `--lambda_=1.0 --batch_size=512 --test_batch=64` (reference README.md:77) on the CREDITED
    arithmetic -- float32 grade (`--logits_dtype x3`, the default of Engine / main.py / bench.py; rounds 1-4 asserted this run with
    bf16 logit operands) -- and on the default session path (packed tiles: the feeder announces ~10 % real positions).
    The reference's figure gives ADER on YOOCHOOSE 72.38 % Recall@20 / 36.71 % MRR@20 over the 16 periods: this run must be within 0.3
    point of both (measured: 72.34 / 36.71 unpacked in round 4) and follow the per-period curve."""
    out = _run(["--dataset", "YOOCHOOSE", "--lambda_", "1.0", "--batch_size", "512", "--test_batch", "64", "--logits_dtype", "x3"])
    per = out["periods"]
    assert per[0]["max_item"] == 12885 and per[-1]["max_item"] == 25750
    _against_figure(out, "YOOCHOOSE", "ADER", 0.3, 0.35)


# ---- the baselines of the reference's figure (README.md:83-86: --finetune / --dropout / --ewc), float32 grade, both datasets.
# Measured (profiles/e2e_r3/baselines_scan.txt): YOOCHOOSE Finetune 71.83 / 36.50, Dropout 72.21 / 36.61, EWC 71.90 / 36.54,
# ADER 72.34 / 36.71 against the figure's 71.86 / 36.49, 72.20 / 36.60, 71.91 / 36.53, 72.38 / 36.71 -- every average within 0.04 point.
# DIGINETICA (a third of the data: a single run's 16-period average carries +-0.3 of trajectory noise) -- three seeds each, round 4
# (profiles/e2e_r4/seed_scan_baselines.txt): Finetune 47.40 / 46.93 / 47.52 (mean 47.28 = the figure's 47.28), Dropout 48.91 / 48.88 /
# 48.80 (mean 48.86, figure 49.07), EWC 47.59 / 46.99 / 47.26 (mean 47.28, figure 47.66); MRR@20 means 16.12 / 16.82 / 16.10 against
# 16.01 / 16.86 / 16.28.  Round 3's single runs (47.04, 48.72, 47.18) were low draws, not a bias of the Finetune path; Dropout and EWC
# sit 0.2 / 0.4 under the figure on average.  The EWC baseline runs the unfused step, which is bitwise reproducible since round 4
# (seed 0 twice: identical), so it is in the suite now.  Not in the suite (time budget): YOOCHOOSE Joint (measured: profiles/e2e_r3/).
BASELINES = [
    ("YOOCHOOSE", "Finetune", ["--finetune", "True"], 0.3, 0.4),          # (in the suite since round 6: a 16-period run is ~20 s now)
    ("YOOCHOOSE", "Dropout", ["--dropout", "True"], 0.3, 0.4),
    ("YOOCHOOSE", "EWC", ["--ewc", "True", "--lambda_", "1.0"], 0.3, 0.4),
    ("DIGINETICA", "Finetune", ["--finetune", "True"], 0.6, 0.8),
    ("DIGINETICA", "Dropout", ["--dropout", "True"], 0.6, 0.8),
    ("DIGINETICA", "EWC", ["--ewc", "True"], 0.6, 0.8),
    # the upper-bound column of the figure: every period trains on all data so far (main.py:138-142: 218,499 steps over the 16 periods,
    # 85 s since the native fed steps of round 6; measured 49.96 / 17.34 against the figure's 50.03 / 17.31)
    ("DIGINETICA", "Joint", ["--joint", "True"], 0.6, 0.8),
]


@pytest.mark.parametrize("dataset,method,flags,tol_avg,tol_period", BASELINES, ids=["%s-%s" % (b[0][:4], b[1]) for b in BASELINES])
def test_baselines_match_the_published_curves(dataset, method, flags, tol_avg, tol_period):
    extra = ["--batch_size", "512", "--test_batch", "64"] if dataset == "YOOCHOOSE" else []
    out = _run(["--dataset", dataset, "--logits_dtype", "x3", "--save_dir", method] + extra + flags)
    _against_figure(out, dataset, method, tol_avg, tol_period)
    if dataset == "DIGINETICA" and method == "Finetune":
        # ... and against the ORACLE's own 16-period run of the same configuration (tests/golden/make_oracle_finetune16.py: the CPU
        # restatement through the whole continual loop, 1.2 CPU-hours; itself asserted against the figure by tests/test_oracle_model.py):
        # same data, same seeds, same flags -- the two trajectories differ by arithmetic only (and by what early stopping makes of it)
        orc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_finetune16.json")))
        assert len(orc["periods"]) == 16
        for key, tol_a, tol_p in (("recall20", 0.5, 0.7), ("mrr20", 0.35, 0.45)):
            mine = [100.0 * p_[key] for p_ in out["periods"]]
            theirs = [100.0 * p_[key] for p_ in orc["periods"]]
            d_avg = sum(mine) / 16 - sum(theirs) / 16
            mad = sum(abs(a - b) for a, b in zip(mine, theirs)) / 16
            print("DIGINETICA Finetune %s: HIP average %.2f, oracle %.2f (delta %+.2f), per-period mean |delta| %.2f"
                  % (key, sum(mine) / 16, sum(theirs) / 16, d_avg, mad))
            assert abs(d_avg) <= tol_a and mad <= tol_p, (key, d_avg, mad)


def test_diginetica_ader_float32_grade_inside_the_poster_band():
    """The same 16-period run with the float32-grade logit kernels (logits_dtype = x3: every product as three bf16 MFMAs on hi/lo
    splits, the reference's fp32 arithmetic of ADER.py:91-93): the band of the bf16 run must hold as well."""
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "x3"])
    avg = out["average"]
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    assert 49.4 <= r20 <= 50.6, ("Recall@20 outside the poster band", r20)
    assert 16.9 <= m20 <= 17.7, ("MRR@20 outside the poster band", m20)
    _against_figure(out, "DIGINETICA", "ADER", 0.5, 0.8)      # the figure: 50.21 / 17.32
    # ... and against the ORACLE's own 16-period ADER run (tests/golden/make_oracle_ader16.py: oracle/ader_ref_cpu.py + oracle/herding_ref.py
    # through the same host loop, 239 CPU-minutes; itself asserted against the figure by tests/test_oracle_model.py): same data, seeds and flags
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_ader16.json")
    if os.path.exists(path):
        orc = json.load(open(path))
        assert len(orc["periods"]) == 16
        # (measured at the end of round 4: averages 50.19 / 17.40 against the oracle's 50.28 / 17.42, per-period mean |delta| 0.22 / 0.08)
        for key, tol_a, tol_p in (("recall20", 0.45, 0.6), ("mrr20", 0.3, 0.35)):
            mine = [100.0 * p_[key] for p_ in out["periods"]]
            theirs = [100.0 * p_[key] for p_ in orc["periods"]]
            d_avg = sum(mine) / 16 - sum(theirs) / 16
            mad = sum(abs(a - b) for a, b in zip(mine, theirs)) / 16
            print("DIGINETICA ADER %s: HIP average %.2f, oracle %.2f (delta %+.2f), per-period mean |delta| %.2f"
                  % (key, sum(mine) / 16, sum(theirs) / 16, d_avg, mad))
            assert abs(d_avg) <= tol_a and mad <= tol_p, (key, d_avg, mad)


# ---- the other columns of the poster's DIGINETICA table (reference flags main.py:83-91, command lines
# README.md:88-93), float32 grade.  Every column is one flag set of the reference's command line.
#   * ADER-equal / ADER-fix (distillation on): must land within +- 0.5 point of the published Recall@20 and MRR@20 (measured round 3:
#     50.11 / 17.35 and 50.10 / 17.37 against 49.92 / 17.23 and 50.09 / 17.29).
#   * ER-herding / ER-random / ER-loss (--disable_distillation: one-hot replay, ADER.py:126-131) are asserted AT THE REFERENCE'S DOCUMENTED
#     COMMAND LINES (README.md:88-90: no --lambda_, i.e. main.py's default base weight 0.8) as a CHARACTERISED DEVIATION: this build lands
#     1.1-1.5 points of Recall@20 BELOW the poster there (round 4, float32 grade: ER-herding 47.94 / 16.28, ER-random 48.04 / 16.33,
#     ER-loss 47.93 / 16.24 against 49.44 / 16.95, 49.14 / 16.79, 49.31 / 16.90; the exact-f32 kernels give the same: 48.29) while the
#     one-hot exemplar loss and its gradients match the CPU restatement of ADER.py:108-131 at the op level (test_gpu_parity: mode
#     "onehot_ex") and the distilled columns of the same table are met within 0.2.  NOT reproduced at the documented flags -- the test
#     pins this build's own values (+- 0.5) and bounds the gap, so that neither a regression nor a silent "fix by tuning" goes unnoticed.
#     For information only (not asserted; profiles/e2e_r3/er_lambda_scan.txt): a scan of the base weight gives lambda_ 0.1 / 0.2 / 0.3 /
#     0.4 / 0.6 / 0.8 -> Recall@20 49.16 / 49.23 / 49.01 / 48.94 / 48.70 / 48.33, i.e. at --lambda_ 0.2 the poster's ER numbers would be
#     met within 0.25 point -- a fitted value, which the README's command lines do not carry.
#   * ER-loss: the reference's `loss` selector ranks a 0-d scalar (util.py:482-488: `model.loss` is the batch mean), so what its code
#     EXECUTES is "keep the first candidate of every label with a quota": `--selection loss_ref` reproduces that exemplar set (this
#     build's `--selection loss` ranks by the per-row loss the method documents).
POSTER = [
    ("ER-herding", ["--disable_distillation", "True"], 49.44, 16.95, (47.94, 16.28)),
    ("ER-random", ["--disable_distillation", "True", "--selection", "random"], 49.14, 16.79, (48.04, 16.33)),
    ("ER-loss", ["--disable_distillation", "True", "--selection", "loss_ref"], 49.31, 16.90, (47.93, 16.24)),
    ("ADER-equal", ["--equal_exemplar", "True"], 49.92, 17.23, None),
    ("ADER-fix", ["--fix_lambda", "True"], 50.09, 17.29, None),
]


@pytest.mark.parametrize("name,flags,r20_ref,m20_ref,own", POSTER, ids=[p[0] for p in POSTER])
def test_diginetica_poster_columns_float32_grade(name, flags, r20_ref, m20_ref, own):
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "x3", "--save_dir", name] + flags)      # (a later --logits_dtype wins)
    avg = out["average"]
    assert len(out["periods"]) == 16
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    print("%s: Recall@20 %.2f (poster %.2f, delta %+.2f)  MRR@20 %.2f (poster %.2f, delta %+.2f)"
          % (name, r20, r20_ref, r20 - r20_ref, m20, m20_ref, m20 - m20_ref))
    if own is None:
        assert abs(r20 - r20_ref) <= 0.5, (name, "Recall@20", r20, r20_ref)
        assert abs(m20 - m20_ref) <= 0.5, (name, "MRR@20", m20, m20_ref)
    else:       # characterised deviation from the poster (see above and test_er_herding_follows_the_oracle): this build's level
        # (+- 0.6: a 16-period average of this configuration moves by +-0.4 between builds whose arithmetic differs in the last bit --
        # early stopping is chaotic: round 3 48.33, round 4 47.94, packed / unpacked session kernels 0.3 apart over four periods)
        assert abs(r20 - own[0]) <= 0.6 and abs(m20 - own[1]) <= 0.5, (name, r20, m20, own)
        assert abs(r20 - r20_ref) <= 2.0 and abs(m20 - m20_ref) <= 1.1, (name, "gap to the poster grew", r20, m20)


def test_er_herding_follows_the_oracle():
    """Where the ER columns' gap to the poster comes from: NOT from the kernels.  tests/golden/oracle_er4.json is
    the CPU oracle -- oracle/ader_ref_cpu.py with the one-hot replay loss of ADER.py:126-131, oracle/herding_ref.py, the product's host
    loop, no HIP kernel -- through the first four DIGINETICA periods at the reference's documented ER-herding command line
    (--disable_distillation True, base weight 0.8; tests/golden/make_oracle_ader16.py --disable_distillation True --max_periods 4
    --out oracle_er4.json, 56 CPU-minutes): Recall@20 49.02 / 49.26 / 49.39 / 49.47.  The HIP engine at the same flags must follow it
    (measured: float32 grade packed 49.24 / 49.58 / 49.61 / 49.60, unpacked 49.25 / 49.53 / 49.10 / 48.86, exact-f32 logits 49.34 /
    49.30 / 49.54 / 48.78: averages within 0.25 of the oracle's 49.28, single periods within the early-stopping noise).  Both sit
    ~1.6 points under the distilled run of the same periods (figure: 50.80 / 51.23 / 51.19) where the poster's 16-period averages
    differ by 0.77: with the loss (op-level test "onehot_ex"), the feeders (golden fixtures) and lambda's inputs (item_num_prev /
    max_item, exemplar_size / train_size: main.py:181-203 line by line) matching the reference, what is left is the poster's
    base weight (its README: hyper-parameters may be tuned per model; at --lambda_ 0.2 this build gives the poster's numbers)."""
    orc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_er4.json")))
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "x3", "--disable_distillation", "True", "--max_periods", "4",
                "--save_dir", "ER-herding-4"])
    assert len(out["periods"]) == 4 and len(orc["periods"]) == 4
    for key, tol_a, tol_p in (("recall20", 0.5, 0.6), ("mrr20", 0.35, 0.45)):
        mine = [100.0 * p_[key] for p_ in out["periods"]]
        theirs = [100.0 * p_[key] for p_ in orc["periods"]]
        d_avg = sum(mine) / 4 - sum(theirs) / 4
        mad = sum(abs(a - b) for a, b in zip(mine, theirs)) / 4
        print("ER-herding %s: HIP %s, oracle %s (average delta %+.2f, per-period mean |delta| %.2f)"
              % (key, ["%.2f" % x for x in mine], ["%.2f" % x for x in theirs], d_avg, mad))
        assert abs(d_avg) <= tol_a and mad <= tol_p, (key, d_avg, mad)


def test_diginetica_one_attention_block_named_variant():
    """BASELINE.json configs[1] names "1 attn block" (the reference default is num_blocks = 2, main.py:99: SURVEY 8d "run L = 2 for
    parity and report L = 1 as the named variant").  No published number exists for L = 1: the run must complete all 16 periods
    and stay within 2 points of the two-block result (a sanity bound, not a parity claim); the value is logged."""
    out = _run(["--dataset", "DIGINETICA", "--logits_dtype", "x3", "--num_blocks", "1", "--save_dir", "ADER-L1"])
    avg = out["average"]
    r20, m20 = 100.0 * avg["recall20"], 100.0 * avg["mrr20"]
    print("L = 1: Recall@20 %.2f  MRR@20 %.2f" % (r20, m20))
    assert len(out["periods"]) == 16 and 48.0 <= r20 <= 52.0 and 15.5 <= m20 <= 19.0
