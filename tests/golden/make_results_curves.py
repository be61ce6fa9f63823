"""Recovers the per-period test curves of the reference's published figure (results.svg, embedded at README.md:101-106: Recall@20 and
MRR@20 per period, both datasets, Finetune / Dropout / EWC / ADER / Joint) into tests/golden/results_svg_curves.json.

The figure carries no machine-readable numbers: text is drawn as glyph outlines and the curves as paths.  But the geometry is exact:
  * each curve is ONE 16-point path (periods 1..16) in its legend colour; the legend labels are glyph sequences whose letter pattern
    identifies them (8 glyphs F-i-n-e-t-u-n-e with the n / e repeats, 7 glyphs D-r-o-p-o-u-t, 3, 5 (shares o-i-n-t with the others),
    4 (shares D and E): Finetune, Dropout, EWC, Joint, ADER);
  * the digit glyphs are identified by the x axis, whose tick labels read 1..16 left to right; the y tick labels then read 46..52 /
    15..18 (DIGINETICA Recall@20 / MRR@20, percent) and 70..74 / 35..38 (YOOCHOOSE), and the tick MARKS give the pixel positions.
Runs only where /root/reference exists (the build container); the fixture is data -- 320 numbers -- not reference source text."""
import json
import os
import re

SRC = "/root/reference/results.svg"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "results_svg_curves.json")
H = 331.985625      # the paths carry transform="matrix(1,0,0,-1,0,H)"
NAMES = {"75%,0%,75%": "Finetune", "0%,50%,0%": "Dropout", "100%,64.704895%,0%": "EWC", "0%,0%,100%": "Joint", "100%,0%,0%": "ADER"}


def main():
    s = open(SRC).read()
    body = s[s.index('<g id="surface1"'):]
    uses = re.findall(r'<use xlink:href="#(glyph\d+-\d+)" x="([\d.]+)" y="([\d.]+)"/>', body)
    # digits from the x axis (y = 292.2): labels 1..16 left to right
    xl = sorted((float(x), g) for g, x, y in uses if abs(float(y) - 292.2) < 0.3)
    digit = {}
    for k, (x, g) in enumerate(xl[:9]):
        digit[g] = str(k + 1)
    digit[xl[10][1]] = "0"              # "10" = glyphs 9, 10 of the row
    assert digit[xl[9][1]] == "1" and len(digit) == 10
    paths = re.findall(r'<path style="([^"]*)" d="([^"]*)"( transform="([^"]*)")?', body)

    def pts(d, flip):
        return [(float(a), H - float(b) if flip else float(b)) for a, b in re.findall(r'[ML] ([\d.\-]+) ([\d.\-]+)', d)]
    ticks = []
    for st, d, _, tr in paths:
        p = pts(d, bool(tr))
        if "stroke:rgb(0%,0%,0%)" in st and len(p) == 2 and abs(p[0][1] - p[1][1]) < 1e-6 and abs(p[0][0] - p[1][0]) < 6:
            ticks.append((min(p[0][0], p[1][0]), p[0][1]))
    # y tick labels: two digit glyphs per label, left (x ~ 23, 29) and right (x ~ 471, 477) columns; a label belongs to the nearest tick
    labels = {}
    for g, x, y in uses:
        x, y = float(x), float(y)
        if g in digit and (x < 35 or 465 < x < 485) and y < 285:
            labels.setdefault((x < 100, round(y, 1)), []).append((x, digit[g]))
    cal = {}
    for (left, y), gl in labels.items():
        val = float("".join(dg for _, dg in sorted(gl)))
        ty = min((t for t in ticks if (t[0] < 100) == left), key=lambda t: abs(t[1] - (y - 3.7)))[1]   # label baseline sits 3.7 below
        cal.setdefault((left, ty < 150), []).append((ty, val))
    out = {}
    for st, d, _, tr in paths:
        col = re.search(r'stroke:rgb\(([^)]*)\)', st)
        p = pts(d, bool(tr))
        if not (col and len(p) == 16 and col.group(1) in NAMES):
            continue
        left, top = p[0][0] < 400, p[0][1] < 150
        (y1, v1), (y2, v2) = min(cal[(left, top)]), max(cal[(left, top)])
        vals = [round(v1 + (y - y1) * (v2 - v1) / (y2 - y1), 3) for _, y in p]
        out.setdefault("DIGINETICA" if left else "YOOCHOOSE", {}).setdefault(NAMES[col.group(1)], {})["recall20" if top else "mrr20"] = vals
    json.dump({"source": "reference results.svg (README.md:101-106): per-period TEST Recall@20 / MRR@20 in percent, periods 1..16, recovered from "
                         "the curve geometry (tests/golden/make_results_curves.py)", "curves": out}, open(OUT, "w"), indent=1)
    for ds in out:
        for m, r in out[ds].items():
            print("%-10s %-8s Recall@20 %.2f  MRR@20 %.2f" % (ds, m, sum(r["recall20"]) / 16, sum(r["mrr20"]) / 16))


if __name__ == "__main__":
    main()
