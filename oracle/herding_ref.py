"""CPU ORACLE (test infrastructure, NOT product code) for herding exemplar selection.

Restates ExemplarGenerator.herding (reference util.py:401-434) under ONE canonical float32
spec, which is also the spec of the HIP kernel (ader_amd/csrc/herding.hip) so the two can be
compared bit-exactly:

  * every product and every sum is individually rounded to float32 (no FMA contraction);
  * norm2[j] = sum_c rep[j,c]^2 accumulated sequentially c = 0..H-1;  norm = sqrt(norm2);
  * D[c,j]  = rep[j,c] / norm[j]                               (util.py:419)
  * mu[c]   = (sum_j D[c,j], sequential j = 0..n-1) / float32(n) (util.py:420)
  * t[j]    = sum_c w[c]*D[c,j] accumulated sequentially c = 0..H-1 (util.py:426)
  * argmax takes the FIRST maximum (np.argmax, util.py:427)
  * w      <- (w + mu) - D[:, i]                                (util.py:428)
  * loop while len(selected) != m and step < 1.1*m  (float64 compare, util.py:425)

PARITY STATUS: pinned on the duplicate-free cases of tests/golden/herding.json, produced by the
reference's own herding() in this container (tests/golden/make_golden.py).  The reference's
np.dot goes through OpenBLAS sgemv whose blocking/FMA order differs from any fixed spec, so
candidates that are exact duplicates (exact ties) may resolve differently: those cases are a
characterised deviation, reported, not asserted.
"""
import numpy as np


def max_steps(m):
    """Number of loop iterations allowed by `step_t < 1.1 * m` (float64), util.py:425."""
    lim = 1.1 * float(m)
    k = int(lim)
    while k < lim:
        k += 1
    return k


def normalise(rep):
    rep = np.ascontiguousarray(rep, dtype=np.float32)
    n, H = rep.shape
    norm2 = np.zeros(n, dtype=np.float32)
    for c in range(H):
        norm2 = norm2 + rep[:, c] * rep[:, c]
    norm = np.sqrt(norm2)
    with np.errstate(divide="ignore", invalid="ignore"):
        D = (rep.T / norm[None, :]).astype(np.float32)          # [H, n]
    mu = np.zeros(H, dtype=np.float32)
    for j in range(n):
        mu = mu + D[:, j]
    mu = mu / np.float32(n)
    return D, mu


def herding_select(rep, m):
    """Returns (selected indices in selection order, steps executed)."""
    rep = np.asarray(rep, dtype=np.float32)
    n, H = rep.shape
    m = int(min(m, n))
    if m == 0:
        return [], 0
    D, mu = normalise(rep)
    w = mu.copy()
    selected, chosen = [], np.zeros(n, dtype=bool)
    step, lim = 0, max_steps(m)
    while len(selected) != m and step < lim:
        t = np.zeros(n, dtype=np.float32)
        for c in range(H):
            t = t + w[c] * D[c, :]
        i = int(np.argmax(t))
        w = (w + mu) - D[:, i]
        step += 1
        if not chosen[i]:
            chosen[i] = True
            selected.append(i)
    return selected, step
