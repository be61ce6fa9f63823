/* CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of
 * ExemplarGenerator.herding (reference util.py:401-434) under the canonical float32 spec
 * documented in oracle/herding_ref.py (sequential sums, every op individually rounded,
 * first-max argmax, loop bound `step < 1.1*m` evaluated in double).
 *
 * Build (see oracle/Makefile):  gcc -O2 -ffp-contract=off -fPIC -shared -o _build/libherding_ref.so herding_ref.c -lm
 * -ffp-contract=off is REQUIRED: the spec has no fused multiply-add.
 *
 * Pinned against tests/golden/herding.json (outputs of the reference's own herding()).
 * Used only by tests/ and by bench.py's cpu_baseline leg for the herding pass.
 */
#include <math.h>
#include <stdlib.h>

/* rep: [n][H] row-major float32.  sel_out: room for min(m,n) ints.  Returns number selected;
 * *steps_out (optional) receives the number of loop iterations executed. */
int herding_ref(const float *rep, int n, int H, int m, int *sel_out, int *steps_out) {
    if (m > n) m = n;
    if (steps_out) *steps_out = 0;
    if (m <= 0 || n <= 0) return 0;
    float *D = (float *)malloc(sizeof(float) * (size_t)H * (size_t)n);   /* [H][n] */
    float *mu = (float *)malloc(sizeof(float) * (size_t)H);
    float *w = (float *)malloc(sizeof(float) * (size_t)H);
    float *t = (float *)malloc(sizeof(float) * (size_t)n);
    char *chosen = (char *)calloc((size_t)n, 1);
    for (int j = 0; j < n; ++j) {
        float s = 0.0f;
        for (int c = 0; c < H; ++c) { float x = rep[(size_t)j * H + c]; float p = x * x; s = s + p; }
        float nrm = sqrtf(s);
        for (int c = 0; c < H; ++c) D[(size_t)c * n + j] = rep[(size_t)j * H + c] / nrm;
    }
    for (int c = 0; c < H; ++c) {
        float s = 0.0f;
        for (int j = 0; j < n; ++j) s = s + D[(size_t)c * n + j];
        mu[c] = s / (float)n;
        w[c] = mu[c];
    }
    const double lim = 1.1 * (double)m;
    int nsel = 0, step = 0;
    while (nsel != m && (double)step < lim) {
        for (int j = 0; j < n; ++j) t[j] = 0.0f;
        for (int c = 0; c < H; ++c) {
            const float wc = w[c];
            const float *Dc = D + (size_t)c * n;
            for (int j = 0; j < n; ++j) { float p = wc * Dc[j]; t[j] = t[j] + p; }
        }
        int best = 0;
        float bv = t[0];
        for (int j = 1; j < n; ++j) {
            /* np.argmax: first maximum; a NaN is treated as larger than everything (first NaN wins) */
            if ((t[j] > bv) || (t[j] != t[j] && bv == bv)) { bv = t[j]; best = j; }
        }
        for (int c = 0; c < H; ++c) { float a = w[c] + mu[c]; w[c] = a - D[(size_t)c * n + best]; }
        ++step;
        if (!chosen[best]) { chosen[best] = 1; sel_out[nsel++] = best; }
    }
    if (steps_out) *steps_out = step;
    free(D); free(mu); free(w); free(t); free(chosen);
    return nsel;
}
