// Segmented herding exemplar selection: one workgroup per label group, all groups in ONE launch.
// Reference: ExemplarGenerator.herding, util.py:401-434 (called once per label from util.py:447-457):
//   D = rep^T / ||rep^T||_2 ; mu = mean_j D ; w = mu ; repeat: i = argmax_j (w . D[:,j]) ; w += mu - D[:,i]
//   until m distinct indices were picked or 1.1*m steps were made.
// Canonical float32 spec shared with oracle/herding_ref.{py,c}: every product/sum individually rounded
// (this file is compiled with -ffp-contract=off and correctly rounded divide/sqrt), dot products and means
// accumulated sequentially in index order, first-maximum argmax.  Bit-exact against the oracle.
#include "common.h"
#include "../../include/ader_hip.h"

#define HMAX 256

struct Best { float v; int i; };

// np.argmax order: NaN beats everything (first NaN wins), otherwise larger value, ties -> lower index
__device__ __forceinline__ bool beats(float xv, int xi, float yv, int yi) {
    const bool xn = xv != xv, yn = yv != yv;
    if (xn != yn) return xn;
    if (!xn && xv != yv) return xv > yv;
    return xi < yi;
}

__global__ __launch_bounds__(256) void k_herding(const float* __restrict__ rep, const long* __restrict__ seg, const int* __restrict__ quota,
                                                 const int* __restrict__ max_steps, float* __restrict__ D, unsigned char* __restrict__ chosen,
                                                 int* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ steps_out, int H) {
    __shared__ float mu[HMAX];
    __shared__ float w[HMAX];
    __shared__ Best red[256];
    __shared__ int s_best;
    const int g = blockIdx.x, tid = threadIdx.x;
    const long off = seg[g];
    const int n = (int)(seg[g + 1] - off);
    const int m = min(quota[g], n);
    const int lim = max_steps[g];
    if (m <= 0 || n <= 0) {
        if (tid == 0) { sel_cnt[g] = 0; if (steps_out) steps_out[g] = 0; }
        return;
    }
    const float* R = rep + (size_t)off * H;
    float* Dg = D + (size_t)off * H;                  // [H][n]
    for (int j = tid; j < n; j += 256) {
        float s = 0.0f;
        for (int c = 0; c < H; ++c) { const float x = R[(size_t)j * H + c]; const float p = x * x; s = s + p; }
        const float nrm = sqrtf(s);
        for (int c = 0; c < H; ++c) Dg[(size_t)c * n + j] = R[(size_t)j * H + c] / nrm;
    }
    __syncthreads();
    for (int c = tid; c < H; c += 256) {
        float s = 0.0f;
        for (int j = 0; j < n; ++j) s = s + Dg[(size_t)c * n + j];
        const float v = s / (float)n;
        mu[c] = v; w[c] = v;
    }
    __syncthreads();
    int nsel = 0;
    int step = 0;
    while (nsel != m && step < lim) {
        Best b; b.v = 0.0f; b.i = 0x7fffffff;
        bool have = false;
        for (int j = tid; j < n; j += 256) {
            float t = 0.0f;
            for (int c = 0; c < H; ++c) { const float p = w[c] * Dg[(size_t)c * n + j]; t = t + p; }
            if (!have || beats(t, j, b.v, b.i)) { b.v = t; b.i = j; have = true; }
        }
        if (!have) { b.v = -INFINITY; b.i = 0x7fffffff; }
        red[tid] = b;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const Best o = red[tid + s];
                const Best c = red[tid];
                // an empty slot (index 0x7fffffff) never wins against a real candidate
                if (o.i != 0x7fffffff && (c.i == 0x7fffffff || beats(o.v, o.i, c.v, c.i))) red[tid] = o;
            }
            __syncthreads();
        }
        if (tid == 0) s_best = red[0].i;
        __syncthreads();
        const int best = s_best;
        for (int c = tid; c < H; c += 256) { const float a = w[c] + mu[c]; w[c] = a - Dg[(size_t)c * n + best]; }
        ++step;
        const bool fresh = chosen[off + best] == 0;     // uniform: every thread reads the same byte
        __syncthreads();
        if (fresh) {
            if (tid == 0) { chosen[off + best] = 1; sel[off + nsel] = best; }
            ++nsel;
        }
        __syncthreads();
    }
    if (tid == 0) { sel_cnt[g] = nsel; if (steps_out) steps_out[g] = step; }
}

extern "C" {

// rep [n_total,H] candidate representations in group order; seg [G+1] group offsets (int64); quota [G]; max_steps [G]
// (= number of integers k with k < 1.1*min(quota,n) in float64, computed by the host as the reference's loop does).
// Scratch: D n_total*H floats, chosen n_total bytes (zeroed here).  sel [n_total]: first sel_cnt[g] entries of each
// group's span are the selected LOCAL indices in selection order.
int ader_herding_select(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total, int H,
                        float* D, unsigned char* chosen, int* sel, int* sel_cnt, int* steps_out, void* stream) {
    if (G <= 0) return 0;
    if (H > HMAX) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(chosen, 0, (size_t)n_total, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_herding, dim3(G), dim3(256), 0, st, rep, seg, quota, max_steps, D, chosen, sel, sel_cnt, steps_out, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
