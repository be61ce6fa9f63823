// Segmented herding exemplar selection over all label groups of a period in ONE launch.
// Reference: ExemplarGenerator.herding, util.py:401-434 (called once per label from util.py:447-457):
//   D = rep^T / ||rep^T||_2 ; mu = mean_j D ; w = mu ; repeat: i = argmax_j (w . D[:,j]) ; w += mu - D[:,i]
//   until m distinct indices were picked or 1.1*m steps were made.
// Canonical float32 spec shared with oracle/herding_ref.{py,c}: every product/sum individually rounded
// (this file is compiled with -ffp-contract=off and correctly rounded divide/sqrt), dot products and means
// accumulated sequentially in index order, first-maximum argmax.  Bit-exact against the oracle.
//
// Two kernels:
//   k_herding      any H <= 256: one 256-thread workgroup per group, D streamed from L2 every iteration (round 1-3 kernel; the
//                  generic-H path and the kernel-vs-kernel reference of the tests: ader_herding_select_generic).
//   k_herding_reg  H = 150 (the reference's hidden_units): a candidate's normalised column D[:,j] (150 floats) lives in the
//                  REGISTERS of the lane that owns it, so an iteration's n dot products read nothing but the 600-byte w from
//                  LDS.  The per-candidate arithmetic (sequential mul / add chain over c) is the canonical one, so the
//                  selections are bit-identical.  Persistent 512-thread workgroups (one per CU) pull work items from a ticket
//                  counter in size order -- the period's few huge groups first (longest-processing-time-first: the launch is as
//                  long as its largest group), then the mid-sized ones, then the thousands of tiny ones eight to a workgroup:
//                    role C  n > 512 : one workgroup; candidates 0..511 in registers, the next 240 in LDS, the rest streamed
//                                      from the L2-resident scratch copy (a 1,710-row group: 0.57 MB instead of 1.03 MB per step)
//                    role B  65..512 : one workgroup, every candidate in registers
//                    role A  1..64   : one WAVE per group (no workgroup barrier in the loop), eight groups per item
//                  (YOOCHOOSE period 1 has 12,113 groups, median 3 rows, largest 1,710.)
#include "common.h"
#include "../../include/ader_hip.h"

#define HMAX 256

struct Best { float v; int i; };

// np.argmax order: NaN beats everything (first NaN wins), otherwise larger value, ties -> lower index
__device__ __forceinline__ bool beats(float xv, int xi, float yv, int yi) {
    // branch-free (selects): the wave reductions call this six times per iteration
    const bool xn = xv != xv, yn = yv != yv;
    const bool by_value = !xn & !yn & (xv != yv);        // two different numbers: the larger one
    const bool r_same = by_value ? (xv > yv) : (xi < yi);
    return (xn != yn) ? xn : r_same;
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


// ================================================================================================ register-resident kernel
#define HR_H 150
#define HR_HP 152            // w / dbest rows padded to whole float4s
#define HR_T 512             // threads per workgroup = register-resident candidates of a role B / C group
#define HR_LC 240            // role C: candidates HR_T .. HR_T+HR_LC-1 live in LDS
#define HR_BITS 65536        // role B / C: chosen[] as an LDS bitmap up to this many candidates (beyond: the global byte array)
#define HR_EMPTY 0x7fffffff

// work list built on the device by k_herd_prep (ints, at the tail of the D scratch):
//   [0] nC  [1] nB  [2] nA  [3] ticket  [4..15] reserved   [16 ..) group ids: C groups, B groups (largest size class first), A groups
#define HR_HDR 16

__global__ __launch_bounds__(1024) void k_herd_prep(const long* __restrict__ seg, const int* __restrict__ quota, int G,
                                                    int* __restrict__ work, int* __restrict__ sel_cnt, int* __restrict__ steps_out) {
    __shared__ int cnt[5], base[5], cur[5];
    const int tid = threadIdx.x;
    if (tid < 5) { cnt[tid] = 0; cur[tid] = 0; }
    __syncthreads();
    auto cls = [&](int g) -> int {
        const int n = (int)(seg[g + 1] - seg[g]);
        const int m = min(quota[g], n);
        if (m <= 0 || n <= 0) return -1;
        return n > HR_T ? 0 : n > 256 ? 1 : n > 128 ? 2 : n > 64 ? 3 : 4;
    };
    for (int g = tid; g < G; g += 1024) {
        const int k = cls(g);
        if (k >= 0) atomicAdd(&cnt[k], 1);
        else { sel_cnt[g] = 0; if (steps_out) steps_out[g] = 0; }
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k < 5; ++k) { base[k] = o; o += cnt[k]; }
        work[0] = cnt[0]; work[1] = cnt[1] + cnt[2] + cnt[3]; work[2] = cnt[4]; work[3] = 0;
    }
    __syncthreads();
    for (int g = tid; g < G; g += 1024) {
        const int k = cls(g);
        if (k >= 0) work[HR_HDR + base[k] + atomicAdd(&cur[k], 1)] = g;
    }
}

__device__ __forceinline__ Best wave_argmax(Best b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(b.v, o, 64);
        const int oi = __shfl_xor(b.i, o, 64);
        if (oi != HR_EMPTY && (b.i == HR_EMPTY || beats(ov, oi, b.v, b.i))) { b.v = ov; b.i = oi; }
    }
    return b;       // beats() is a total order (ties by index): every lane ends with the same winner
}

// row j of the group -> normalised column in registers, held as 38 aligned 4-register tuples (what ds_read_b128 / ds_write_b128
// move: no copies between the column and the LDS operands).  Canonical spec: s = sum of individually rounded squares in channel
// order, nrm = sqrt(s), d[c] = x[c] / nrm; channels 150, 151 are zero padding.
#define HR_Q (HR_HP / 4)
__device__ __forceinline__ void load_norm(const float* __restrict__ row, f32x4 (&d)[HR_Q]) {
    const float2* r2 = (const float2*)row;              // rows are 600 B: 8-byte aligned
    float s = 0.0f;
#pragma unroll 5
    for (int c = 0; c < HR_H / 2; ++c) {
        const float2 v = r2[c];
        { const float p = v.x * v.x; s = s + p; }
        { const float p = v.y * v.y; s = s + p; }
    }
    const float nrm = sqrtf(s);
    // second pass over the (cache-resident) row, a few channels at a time: a correctly rounded division is a dozen instructions
    // with their own temporaries, and 150 of them scheduled at once do not fit beside the 150 results
#pragma unroll
    for (int q = 0; q < HR_Q; ++q) {
        if ((q & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        const float2 a = r2[2 * q];
        d[q][0] = a.x / nrm;
        d[q][1] = a.y / nrm;
        if (4 * q + 2 < HR_H) { const float2 b = r2[2 * q + 1]; d[q][2] = b.x / nrm; d[q][3] = b.y / nrm; }
        else { d[q][2] = 0.0f; d[q][3] = 0.0f; }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// t = sum_c w[c] * d[c], sequential in c, every product and sum rounded (w: LDS, read as broadcast float4s)
__device__ __forceinline__ float dot_reg(const float* __restrict__ w_s, const f32x4 (&d)[HR_Q]) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < HR_Q; ++q) {
        // (at most eight broadcast reads of w in flight: left alone hipcc hoists all 38 -- 152 registers beside the 152 of d)
        if ((q & 7) == 0) asm volatile("" ::: "memory");
        const f32x4 w4 = ((const f32x4*)w_s)[q];
        { const float p = w4[0] * d[q][0]; t = t + p; }
        { const float p = w4[1] * d[q][1]; t = t + p; }
        if (4 * q + 2 < HR_H) { const float p = w4[2] * d[q][2]; t = t + p; }
        if (4 * q + 3 < HR_H) { const float p = w4[3] * d[q][3]; t = t + p; }
    }
    asm volatile("" ::: "memory");
    return t;
}

__device__ __forceinline__ void publish(float* __restrict__ db_s, const f32x4 (&d)[HR_Q]) {
#pragma unroll
    for (int q = 0; q < HR_Q; ++q) ((f32x4*)db_s)[q] = d[q];
}

// wave-uniform values the compiler cannot prove uniform (loaded through vector memory / LDS): into SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni64(long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffl));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}

struct HerdArgs {
    const float* rep; const long* seg; const int* quota; const int* max_steps;
    float* D; unsigned char* chosen; int* sel; int* sel_cnt; int* steps_out; int* work;
};

// ---- role A: one wave, n <= 64, lane j owns candidate j
__device__ __forceinline__ void herd_wave(const HerdArgs& a, int g, float* __restrict__ w_s, float* __restrict__ db_s, int lane) {
    const long off = uni64(a.seg[g]);
    const int n = uni((int)(a.seg[g + 1] - off));
    const int m = uni(min(a.quota[g], n));
    const int lim = uni(a.max_steps[g]);
    float* Dg = a.D + (size_t)off * HR_H;                // [H][n] scratch: only for the (sequential over j) channel means
    f32x4 d[HR_Q];
    const bool own = lane < n;
    if (own) {
        load_norm(a.rep + (size_t)(off + lane) * HR_H, d);
#pragma unroll
        for (int c = 0; c < HR_H; ++c) Dg[(size_t)c * n + lane] = d[c >> 2][c & 3];
    } else {
#pragma unroll
        for (int q = 0; q < HR_Q; ++q) d[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the wave's own stores, then its loads of them
    float mu_r[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int c = lane + 64 * r;
        float s = 0.0f;
        if (c < HR_H) for (int j = 0; j < n; ++j) s = s + Dg[(size_t)c * n + j];
        mu_r[r] = s / (float)n;
        if (c < HR_HP) w_s[c] = c < HR_H ? mu_r[r] : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    unsigned long long taken = 0ull;                             // uniform: bit j = candidate j already selected
    int nsel = 0, step = 0;
    while (nsel != m && step < lim) {
        Best b;
        b.v = own ? dot_reg(w_s, d) : -INFINITY;
        b.i = own ? lane : HR_EMPTY;
        b = wave_argmax(b);
        const int best = uni(b.i);
        if (lane == best) publish(db_s, d);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = lane + 64 * r;
            if (c < HR_H) { const float s = w_s[c] + mu_r[r]; w_s[c] = s - db_s[c]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        ++step;
        if (!((taken >> best) & 1ull)) {
            taken |= 1ull << best;
            if (lane == 0) a.sel[off + nsel] = best;
            ++nsel;
        }
    }
    if (lane == 0) { a.sel_cnt[g] = nsel; if (a.steps_out) a.steps_out[g] = step; }
}

// ---- roles B / C: one workgroup of HR_T threads, thread t owns candidate t in registers; C: the next HR_LC candidates in LDS,
//      the rest streamed from the scratch copy Dg
__device__ __forceinline__ void herd_block(const HerdArgs& a, int g, float* __restrict__ w_s, float* __restrict__ db_s,
                                           float* __restrict__ Dl, unsigned* __restrict__ bits, Best* __restrict__ red,
                                           int* __restrict__ s_ctl, int tid) {
    const long off = uni64(a.seg[g]);
    const int n = uni((int)(a.seg[g + 1] - off));
    const int m = uni(min(a.quota[g], n));
    const int lim = uni(a.max_steps[g]);
    const int lane = tid & 63, wave = tid >> 6;
    const float* R = a.rep + (size_t)off * HR_H;
    float* Dg = a.D + (size_t)off * HR_H;                // [H][n]
    f32x4 d[HR_Q];
    const bool own = tid < n;
    if (own) {
        load_norm(R + (size_t)tid * HR_H, d);
#pragma unroll
        for (int c = 0; c < HR_H; ++c) Dg[(size_t)c * n + tid] = d[c >> 2][c & 3];
    } else {
#pragma unroll
        for (int q = 0; q < HR_Q; ++q) d[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    for (int j = HR_T + tid; j < n; j += HR_T) {        // candidates beyond the register-resident ones: two passes over the row
        const float* row = R + (size_t)j * HR_H;
        float s = 0.0f;
        for (int c = 0; c < HR_H; ++c) { const float x = row[c]; const float p = x * x; s = s + p; }
        const float nrm = sqrtf(s);
        for (int c = 0; c < HR_H; ++c) Dg[(size_t)c * n + j] = row[c] / nrm;
    }
    const bool lbits = n <= HR_BITS;
    if (lbits) for (int i = tid; i < (n + 31) / 32; i += HR_T) bits[i] = 0u;
    __syncthreads();
    float mu_r = 0.0f;
    if (tid < HR_H) {
        float s = 0.0f;
        for (int j = 0; j < n; ++j) s = s + Dg[(size_t)tid * n + j];
        mu_r = s / (float)n;
        w_s[tid] = mu_r;
    } else if (tid < HR_HP) w_s[tid] = 0.0f;
    const int n_lds = min(max(n - HR_T, 0), HR_LC);     // candidates HR_T .. HR_T + n_lds - 1
    if (tid < n_lds) {
        for (int c = 0; c < HR_H; ++c) Dl[c * HR_LC + tid] = Dg[(size_t)c * n + HR_T + tid];
    }
    __syncthreads();
    int nsel = 0, step = 0;
    while (nsel != m && step < lim) {
        Best b;
        b.v = own ? dot_reg(w_s, d) : -INFINITY;
        b.i = own ? tid : HR_EMPTY;
        if (tid < n_lds) {
            float t = 0.0f;
#pragma unroll 10
            for (int c = 0; c < HR_H; ++c) { const float p = w_s[c] * Dl[c * HR_LC + tid]; t = t + p; }
            const int j = HR_T + tid;
            if (beats(t, j, b.v, b.i)) { b.v = t; b.i = j; }     // (thread tid < n_lds owns a register candidate too: b.i is real)
        }
        for (int j = HR_T + HR_LC + (HR_T - 1 - tid); j < n; j += HR_T) {     // streamed: handed out from the high thread ids down
            float t = 0.0f;
#pragma unroll 10
            for (int c = 0; c < HR_H; ++c) { const float p = w_s[c] * Dg[(size_t)c * n + j]; t = t + p; }
            if (b.i == HR_EMPTY || beats(t, j, b.v, b.i)) { b.v = t; b.i = j; }
        }
        b = wave_argmax(b);
        if (lane == 0) red[wave] = b;
        __syncthreads();
        Best bb = red[0];
#pragma unroll
        for (int k = 1; k < HR_T / 64; ++k) {
            const Best o = red[k];
            if (o.i != HR_EMPTY && (bb.i == HR_EMPTY || beats(o.v, o.i, bb.v, bb.i))) bb = o;
        }
        const int best = uni(bb.i);
        if (tid == best) publish(db_s, d);
        if (tid == 0) {
            bool fresh;
            if (lbits) { const unsigned wd = bits[best >> 5], bt = 1u << (best & 31); fresh = !(wd & bt); bits[best >> 5] = wd | bt; }
            else { fresh = a.chosen[off + best] == 0; a.chosen[off + best] = 1; }
            if (fresh) { a.sel[off + nsel] = best; ++nsel; }
            s_ctl[0] = nsel;
        }
        __syncthreads();
        if (tid < HR_H) {
            float db;
            if (best < HR_T) db = db_s[tid];
            else if (best < HR_T + HR_LC) db = Dl[tid * HR_LC + (best - HR_T)];
            else db = Dg[(size_t)tid * n + best];
            const float s = w_s[tid] + mu_r;
            w_s[tid] = s - db;
        }
        nsel = uni(s_ctl[0]);
        ++step;
        __syncthreads();
    }
    if (tid == 0) { a.sel_cnt[g] = nsel; if (a.steps_out) a.steps_out[g] = step; }
}

__global__ __launch_bounds__(HR_T) void k_herding_reg(HerdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    float* w_all = (float*)hsm;                                   // [8][HR_HP]   (roles B / C use wave 0's)
    float* db_all = w_all + 8 * HR_HP;                            // [8][HR_HP]
    Best* red = (Best*)(db_all + 8 * HR_HP);                      // [8]
    int* s_ctl = (int*)(red + 8);                                 // [4]
    unsigned* bits = (unsigned*)(s_ctl + 4);                      // [HR_BITS / 32]
    float* Dl = (float*)(bits + HR_BITS / 32);                    // [HR_H][HR_LC]
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int nC = uni(a.work[0]), nB = uni(a.work[1]), nA = uni(a.work[2]);
    const int items = nC + nB + (nA + 7) / 8;
    const int* list = a.work + HR_HDR;
    for (;;) {
        // (a pure work ticket: the list it indexes was written by the PREVIOUS launch (k_herd_work), and no workgroup reads what another
        //  writes in this launch -- atomicity is all it needs, no ordering; tests/stress_handoffs.py runs it 200 times cold and warm)
        if (tid == 0) s_ctl[1] = atomicAdd(&a.work[3], 1);
        __syncthreads();
        const int it = uni(s_ctl[1]);
        __syncthreads();
        if (it >= items) break;
        if (it < nC + nB) {
            herd_block(a, uni(list[it]), w_all, db_all, Dl, bits, red, s_ctl, tid);
        } else {
            const int k = (it - nC - nB) * 8 + wave;
            if (k < nA) herd_wave(a, uni(list[nC + nB + k]), w_all + wave * HR_HP, db_all + wave * HR_HP, lane);
        }
        __syncthreads();
    }
}
static const size_t kHerdLds = (size_t)(16 * HR_HP) * 4 + 8 * sizeof(Best) + 16 + HR_BITS / 8 + (size_t)HR_H * HR_LC * 4;

extern "C" {

// rep [n_total,H] candidate representations in group order; seg [G+1] group offsets (int64); quota [G]; max_steps [G]
// (= number of integers k with k < 1.1*min(quota,n) in float64, computed by the host as the reference's loop does).
// Scratch: D n_total*H + G + 64 floats (normalised columns of the groups, then the device-built work list), chosen n_total bytes
// (zeroed here).  sel [n_total]: first sel_cnt[g] entries of each group's span are the selected LOCAL indices in selection order.
static int herding_generic(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total, int H,
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

#ifdef ADER_XCHECK   // the generic kernel under its own name: kernel-vs-kernel checks of the test build only (libader_xcheck.so)
int ader_herding_select_generic(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total, int H,
                                float* D, unsigned char* chosen, int* sel, int* sel_cnt, int* steps_out, void* stream) {
    return herding_generic(rep, seg, quota, max_steps, G, n_total, H, D, chosen, sel, sel_cnt, steps_out, stream);
}
#endif

int ader_herding_select(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total, int H,
                        float* D, unsigned char* chosen, int* sel, int* sel_cnt, int* steps_out, void* stream) {
    if (G <= 0) return 0;
    if (H != HR_H) return herding_generic(rep, seg, quota, max_steps, G, n_total, H, D, chosen, sel, sel_cnt, steps_out, stream);
    hipStream_t st = (hipStream_t)stream;
    static int cus_dev[ADER_MAX_DEV] = {};
    int& cus = cus_dev[ader_cur_dev()];
    static bool attr_dev[ADER_MAX_DEV] = {};
    bool& attr = attr_dev[ader_cur_dev()];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_herding_reg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHerdLds);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -3;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        attr = true;
    }
    hipError_t e = hipMemsetAsync(chosen, 0, (size_t)n_total, st);
    if (e != hipSuccess) return (int)e;
    HerdArgs a;
    a.rep = rep; a.seg = seg; a.quota = quota; a.max_steps = max_steps; a.D = D; a.chosen = chosen; a.sel = sel; a.sel_cnt = sel_cnt;
    a.steps_out = steps_out;
    a.work = (int*)(D + (size_t)n_total * H);
    hipLaunchKernelGGL(k_herd_prep, dim3(1), dim3(1024), 0, st, seg, quota, G, a.work, sel_cnt, steps_out);
    HIP_LAUNCH_CHECK();
    const int grid = G < cus ? G : cus;                 // persistent: one workgroup per CU pulls items from the ticket counter
    hipLaunchKernelGGL(k_herding_reg, dim3(grid), dim3(HR_T), kHerdLds, st, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
