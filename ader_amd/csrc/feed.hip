// Device-side feeder of the train step: the batch is cut from the GPU-resident packed rows by ONE launch.
// Reference: Sampler.sampler / exemplar_sampler gather the rows of a batch by the shuffled index list (util.py:218-262) and
// main.py:229 appends the exemplar rows to the train rows; main.py then feeds the lists to sess.run.  With the rows resident on the
// device (ader_amd/data.py: Sampler.to_device) the same batch is two index_selects per Sampler, a concatenation and -- when the
// batch is padded to its nominal row count -- two more concatenations: ~8 torch launches and ~0.1 ms of host time per step.  This
// kernel writes the step's input tensors [train rows | padding | exemplar rows | padding] directly, from the index slices of the
// epoch plan.  Index work only (bit-exact); HBM traffic is the batch itself (~0.1 MB).
#include "common.h"
#include "../../include/ader_hip.h"

namespace {

// one wave per output row: lanes 0..T-1 copy the inputs, lane 0 the label
__global__ __launch_bounds__(256) void k_feed_step(const int* __restrict__ rows_t, const long* __restrict__ idx_t, int n_t, int Bt,
                                                   const int* __restrict__ rows_e, const long* __restrict__ idx_e, int n_e, int Be, int T,
                                                   int* __restrict__ seq, int* __restrict__ pos, int* __restrict__ ex_pos,
                                                   int* __restrict__ ex_trow) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= Bt + Be) return;
    const bool ex = r >= Bt;
    const int k = ex ? r - Bt : r;
    const bool real = k < (ex ? n_e : n_t);
    long src = -1;
    if (real) src = ex ? idx_e[k] : idx_t[k];
    const int* row = real ? (ex ? rows_e : rows_t) + (size_t)src * (T + 1) : nullptr;
    for (int c = lane; c < T; c += 64) seq[(size_t)r * T + c] = real ? row[c] : 0;
    if (lane == 0) {
        const int lab = real ? row[T] : 0;
        if (ex) {
            if (ex_pos) ex_pos[k] = lab;
            if (ex_trow) ex_trow[k] = real ? (int)src : -1;
        } else {
            pos[k] = lab;
        }
    }
}

__global__ __launch_bounds__(256) void k_concat_i32(const int* __restrict__ a, int na, const int* __restrict__ b, int nb, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < na) out[i] = a[i];
    else if (i < na + nb) out[i] = b[i - na];
}

}  // namespace

extern "C" {

int ader_feed_step(const int* rows_t, const long* idx_t, int n_t, int Bt, const int* rows_e, const long* idx_e, int n_e, int Be, int T,
                   int* seq, int* pos, int* ex_pos, int* ex_trow, void* stream) {
    if (n_t < 0 || n_e < 0 || n_t > Bt || n_e > Be || T <= 0 || Bt < 0 || Be < 0) return -2;
    if ((n_t > 0 && (!rows_t || !idx_t)) || (n_e > 0 && (!rows_e || !idx_e)) || !seq || (Bt > 0 && !pos)) return -2;
    if (Bt + Be == 0) return 0;
    hipLaunchKernelGGL(k_feed_step, dim3((Bt + Be + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows_t, idx_t, n_t, Bt, rows_e, idx_e, n_e,
                       Be, T, seq, pos, ex_pos, ex_trow);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_concat_i32(const int* a, int na, const int* b, int nb, int* out, void* stream) {
    if (na < 0 || nb < 0 || (na + nb > 0 && !out)) return -2;
    if (na + nb == 0) return 0;
    hipLaunchKernelGGL(k_concat_i32, dim3((na + nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
