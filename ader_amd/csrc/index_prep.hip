// Index preparation of the fused table update: the id-sorted lists of the sparse gradient terms.
// Reference: the gradient of tf.nn.embedding_lookup (modules.py:127) and of the one-hot labels (ADER.py:89-93) are sparse in the
// table rows; the fused update adds them per row from lists sorted by item id (deterministic order, no atomics).  This file
// builds those lists on the device: a counting sort by 64-id bucket (the granularity of the update kernels' tiles) whose
// arbitrary atomic arrival order is then replaced by the rank of every entry's position inside its bucket -- four tiny, wide
// kernels on the side stream under the logit forward (a single-workgroup radix sort, tried first, held one CU for 0.19 ms and
// cost the 512-workgroup logit forward a whole extra round).
#include "common.h"
#include "../../include/ader_hip.h"

// key of an id: 0 for id 0 (padding / no label), 1 + (id - 1) / 64 otherwise -- bucket j of the update kernels is key j + 1
__device__ __forceinline__ int key_of(int id, int gran, int nkeys) {
    if (id <= 0) return 0;
    const int k = 1 + (id - 1) / gran;
    return k < nkeys ? k : nkeys - 1;                 // (ids beyond the catalog -- flagged elsewhere -- land in the last bucket)
}

// Pipeline (all kernels tiny and wide, so they share the CUs with the logit forward without displacing a workgroup for long):
//   count   : cnt[list][key] += 1 per entry (integer atomics)
//   scan    : one workgroup per list: exclusive prefix over the keys -> first[key]; writes the bucket offsets start[j] = first[j+1]
//   scatter : slot = first[key] + ticket (atomic) -> tmp[slot] = position            (order inside a bucket: arbitrary)
//   rank    : one thread per slot: rank = #entries of its bucket with a smaller (id, position) -> final slot first[key] + rank
// The result is independent of the atomics' arrival order: inside a bucket the entries are in (id, position) order (so every table
// row receives its contributions in position order: bit-reproducible -- and as one run), buckets are in id order.
__global__ __launch_bounds__(256) void k_ip_count(const int* __restrict__ ids0, int n0, const int* __restrict__ ids1, int n1, int gran,
                                                  int nkeys, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // padding entries (id 0: ~90 % of the positions of real sessions) are left out altogether: the update never reads them, and
    // tens of thousands of atomics on ONE counter would serialise
    if (i < n0) { const int k = key_of(ids0[i], gran, nkeys); if (k) atomicAdd(cnt + k, 1); }
    else if (i < n0 + n1) { const int k = key_of(ids1[i - n0], gran, nkeys); if (k) atomicAdd(cnt + nkeys + k, 1); }
}

// grid = 2 (one workgroup per list), 256 threads (small enough to start on a CU that the logit forward occupies).  cnt [2][nkeys] -> first [2][nkeys + 1] (exclusive prefix, total at the end);
// start_l[j] = first[j + 1], j = 0 .. nkeys - 1 (= the nb + 1 bucket offsets); cnt is cleared for the scatter tickets.
__global__ __launch_bounds__(256) void k_ip_scan(int* __restrict__ cnt, int nkeys, int* __restrict__ first, int* __restrict__ start0,
                                                  int* __restrict__ start1) {
    __shared__ int part[256];
    const int lst = blockIdx.x, t = threadIdx.x;
    int* c = cnt + lst * nkeys;
    int* f = first + lst * (nkeys + 1);
    int* st = lst ? start1 : start0;
    const int per = (nkeys + 255) / 256;
    const int lo = min(nkeys, t * per), hi = min(nkeys, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += c[k];
    part[t] = s;
    __syncthreads();
    if (t < 64) {                                   // one wave scans the 256 partial sums (4 per lane)
        int acc[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = tot; tot += part[t * 4 + k]; }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (t >= o) incl += v;
        }
        const int base = incl - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) part[t * 4 + k] = base + acc[k];
    }
    __syncthreads();
    int run = part[t];
    for (int k = lo; k < hi; ++k) {
        const int v = c[k];
        f[k] = run;
        if (k >= 1) st[k - 1] = run;
        c[k] = 0;
        run += v;
    }
    if (hi == nkeys && lo < hi) { f[nkeys] = run; st[nkeys - 1] = run; }
}

__global__ __launch_bounds__(256) void k_ip_scatter(const int* __restrict__ ids0, int n0, const int* __restrict__ ids1, int n1, int gran,
                                                    int nkeys, int* __restrict__ cnt, const int* __restrict__ first,
                                                    int* __restrict__ tmp, int* __restrict__ tmp_id) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (the id travels with the position: the rank pass then reads two sequential arrays instead of one array and a dependent random
    //  load per comparison -- a hot item's bucket holds hundreds of entries and every one of them walks the whole bucket)
    if (i < n0) {
        const int id = ids0[i], k = key_of(id, gran, nkeys);
        if (k) { const int s = first[k] + atomicAdd(cnt + k, 1); tmp[s] = i; tmp_id[s] = id; }
    } else if (i < n0 + n1) {
        const int id = ids1[i - n0], k = key_of(id, gran, nkeys);
        if (k) { const int s = n0 + first[nkeys + 1 + k] + atomicAdd(cnt + nkeys + k, 1); tmp[s] = i - n0; tmp_id[s] = id; }
    }
}

__global__ __launch_bounds__(256) void k_ip_rank(const int* __restrict__ ids0, int n0, const int* __restrict__ ids1, int n1, int gran,
                                                 int nkeys, const int* __restrict__ first, const int* __restrict__ tmp,
                                                 const int* __restrict__ tmp_id, int* __restrict__ sid0, int* __restrict__ srow0, int* __restrict__ sid1,
                                                 int* __restrict__ srow1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 + n1) return;
    const int lst = i >= n0;
    const int* tp = tmp + (lst ? n0 : 0);
    const int* ti = tmp_id + (lst ? n0 : 0);
    const int* f = first + (lst ? nkeys + 1 : 0);
    const int s = lst ? i - n0 : i;
    if (s >= f[nkeys]) return;                        // only the real (non-padding) entries were scattered: slots [0, total)
    const int my = tp[s];
    const int id = ti[s];
    const int k = key_of(id, gran, nkeys);
    const int k0 = f[k], k1 = f[k + 1];
    // order inside a bucket: by id, then by position (a table row's entries are consecutive and in position order: the update
    // kernels sum a row's run in a register).  Quadratic in the bucket's size: buckets hold a handful of entries, a hot item's
    // a few hundred.
    int rank = 0;
    int u = k0;
    for (; u + 4 <= k1; u += 4) {
        int pu[4], iu[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { pu[j] = tp[u + j]; iu[j] = ti[u + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) rank += (iu[j] < id) || (iu[j] == id && pu[j] < my);
    }
    for (; u < k1; ++u) {
        const int pu = tp[u], iu = ti[u];
        rank += (iu < id) || (iu == id && pu < my);
    }
    (lst ? sid1 : sid0)[k0 + rank] = id;
    (lst ? srow1 : srow0)[k0 + rank] = my;
}

// The whole chain as ONE launch for a small problem (catalogs of the shipped datasets' size: at most 2,049 keys, 128 k entries): one
// 1,024-thread workgroup, counters and offsets in LDS, the phases of the kernels above separated by workgroup barriers (its global
// writes -- tmp, tmp_id, the lists -- are read back by the same workgroup only), and the per-tile records of the update kernels
// (k_tile_meta's: {k0, k1, first 8 (id, row)} per list) written at the end.  Same lists bit for bit: the rank fixes every slot.
// Five launches and a memset fewer per step where the host, not the GPU, sets the pace.
#define IP1_T 1024
__global__ __launch_bounds__(IP1_T) void k_ip_one(const int* __restrict__ ids0, int n0, const int* __restrict__ ids1, int n1, int gran,
                                                  int nkeys, int* __restrict__ tmp, int* __restrict__ tmp_id, int* __restrict__ sid0,
                                                  int* __restrict__ srow0, int* __restrict__ start0, int* __restrict__ sid1,
                                                  int* __restrict__ srow1, int* __restrict__ start1, int* __restrict__ rec, int tm_list) {
    extern __shared__ int ip_sm[];
    int* cnt = ip_sm;                         // [2][nkeys]
    int* first = cnt + 2 * nkeys;             // [2][nkeys + 1]
    int* part = first + 2 * (nkeys + 1);      // [IP1_T]
    const int tid = threadIdx.x, n = n0 + n1;
    for (int i = tid; i < 2 * nkeys; i += IP1_T) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += IP1_T) {
        const int lst = i >= n0, k = key_of(lst ? ids1[i - n0] : ids0[i], gran, nkeys);
        if (k) atomicAdd(cnt + lst * nkeys + k, 1);
    }
    __syncthreads();
    {   // exclusive prefix per list: 512 threads per list, `per` consecutive keys each, Hillis-Steele over the 512 partial sums
        const int lst = tid >> 9, t = tid & 511;
        const int per = (nkeys + 511) / 512;
        const int lo = min(nkeys, t * per), hi = min(nkeys, lo + per);
        int* c = cnt + lst * nkeys;
        int* f = first + lst * (nkeys + 1);
        int* st = lst ? start1 : start0;
        int sum = 0;
        for (int k = lo; k < hi; ++k) sum += c[k];
        part[tid] = sum;
        __syncthreads();
        int incl = sum;
        for (int o = 1; o < 512; o <<= 1) {
            const int v = (t >= o) ? part[tid - o] : 0;
            __syncthreads();
            incl += v;
            part[tid] = incl;
            __syncthreads();
        }
        int run = incl - sum;
        for (int k = lo; k < hi; ++k) {
            const int v = c[k];
            f[k] = run;
            if (k >= 1) st[k - 1] = run;
            c[k] = 0;                           // (tickets of the scatter)
            run += v;
        }
        if (hi == nkeys && lo < hi) { f[nkeys] = run; st[nkeys - 1] = run; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += IP1_T) {
        const int lst = i >= n0, id = lst ? ids1[i - n0] : ids0[i], k = key_of(id, gran, nkeys);
        if (k) {
            const int s_ = (lst ? n0 : 0) + first[lst * (nkeys + 1) + k] + atomicAdd(cnt + lst * nkeys + k, 1);
            tmp[s_] = lst ? i - n0 : i;
            tmp_id[s_] = id;
        }
    }
    __syncthreads();
    const int tot0 = first[nkeys], tot1 = first[(nkeys + 1) + nkeys];
    for (int i = tid; i < tot0 + tot1; i += IP1_T) {
        const int lst = i >= tot0, s_ = lst ? i - tot0 : i;
        const int* tp = tmp + (lst ? n0 : 0);
        const int* ti = tmp_id + (lst ? n0 : 0);
        const int* f = first + lst * (nkeys + 1);
        const int my = tp[s_], id = ti[s_];
        const int k = key_of(id, gran, nkeys);
        const int k0 = f[k], k1 = f[k + 1];
        int rank = 0, u = k0;
        for (; u + 4 <= k1; u += 4) {
            int pu[4], iu[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { pu[j] = tp[u + j]; iu[j] = ti[u + j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) rank += (iu[j] < id) || (iu[j] == id && pu[j] < my);
        }
        for (; u < k1; ++u) {
            const int pu = tp[u], iu = ti[u];
            rank += (iu < id) || (iu == id && pu < my);
        }
        (lst ? sid1 : sid0)[k0 + rank] = id;
        (lst ? srow1 : srow0)[k0 + rank] = my;
    }
    if (!rec) return;
    __syncthreads();
    const int ntiles = nkeys - 1;
    for (int i = tid; i < 2 * ntiles; i += IP1_T) {
        const int tile = i >> 1, lst = i & 1;
        const int* f = first + lst * (nkeys + 1);
        const int* ids = lst ? sid1 : sid0;
        const int* rows = lst ? srow1 : srow0;
        int* mt = rec + (size_t)i * tm_list;
        const int k0 = f[tile + 1], k1 = f[tile + 2];
        mt[0] = k0; mt[1] = k1;
        for (int j = 0; j < 8; ++j) {
            const bool in = k0 + j < k1;
            mt[2 + 2 * j] = in ? ids[k0 + j] : 0;
            mt[3 + 2 * j] = in ? rows[k0 + j] : 0;
        }
    }
}

extern "C" {

// scratch ints needed by ader_sparse_lists: counters [2][nkeys], first [2][nkeys + 1], tmp [n_sp + n_tg], tmp_id [n_sp + n_tg]
static int ip_nkeys(int N) { return (N + ader_fused_bucket_gran() - 1) / ader_fused_bucket_gran() + 1; }
int ader_sparse_lists_scratch_n(int n_sp, int n_tg, int N) { return 2 * ip_nkeys(N) + 2 * (ip_nkeys(N) + 1) + 2 * (n_sp + n_tg); }
// number of bucket offsets per list: ceil(N / gran) + 1
int ader_sparse_lists_starts(int N) { return ip_nkeys(N); }

// Bucketed lists of the two sparse table-gradient terms (input positions seq [n_sp], labels lab [n_tg]; id 0 = none) for
// ader_tab_update / ader_tab_update_sh: entries grouped by 64-id bucket in id order, inside a bucket by (id, POSITION) (so the
// contributions to a table row are consecutive and added in position order: bit-reproducible), their ids and positions, and the bucket offsets
// start[j] (first entry of bucket j = ids [64 j + 1, 64 j + 65)), j = 0 .. ceil(N/64).  Padding entries (id 0) are left out:
// the lists hold start[ceil(N/64)] entries.
int ader_sparse_lists(const int* seq, int n_sp, const int* lab, int n_tg, int N, int* scratch, int* sp_ids, int* sp_rows,
                      int* sp_start, int* tg_ids, int* tg_rows, int* tg_start, void* stream) {
    if (n_sp < 0 || n_tg < 0 || N < 1) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int gran = ader_fused_bucket_gran(), nkeys = ip_nkeys(N), n = n_sp + n_tg;
    int* cnt = scratch;
    int* first = cnt + 2 * nkeys;
    int* tmp = first + 2 * (nkeys + 1);
    int* tmp_id = tmp + n;
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)2 * nkeys * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int g = (n + 255) / 256;
    if (g > 0) hipLaunchKernelGGL(k_ip_count, dim3(g), dim3(256), 0, st, seq, n_sp, lab, n_tg, gran, nkeys, cnt);
    hipLaunchKernelGGL(k_ip_scan, dim3(2), dim3(256), 0, st, cnt, nkeys, first, sp_start, tg_start);
    if (g > 0) {
        hipLaunchKernelGGL(k_ip_scatter, dim3(g), dim3(256), 0, st, seq, n_sp, lab, n_tg, gran, nkeys, cnt, (const int*)first, tmp, tmp_id);
        hipLaunchKernelGGL(k_ip_rank, dim3(g), dim3(256), 0, st, seq, n_sp, lab, n_tg, gran, nkeys, (const int*)first, (const int*)tmp,
                           (const int*)tmp_id, sp_ids, sp_rows, tg_ids, tg_rows);
    }
    HIP_LAUNCH_CHECK();
    return 0;
}

// ader_sparse_lists followed by ader_tab_tile_meta (rec: ader_tab_meta_ints(N) ints, or NULL for the lists alone) -- as ONE launch
// when the problem is small enough for one workgroup's LDS (at most 2,049 keys = 131,072 items, 131,072 entries), as the chain of
// launches above otherwise.  Same outputs either way.
int ader_sparse_lists_meta(const int* seq, int n_sp, const int* lab, int n_tg, int N, int* scratch, int* sp_ids, int* sp_rows,
                           int* sp_start, int* tg_ids, int* tg_rows, int* tg_start, int* rec, void* stream) {
    if (n_sp < 0 || n_tg < 0 || N < 1) return -2;
    const int gran = ader_fused_bucket_gran(), nkeys = ip_nkeys(N), n = n_sp + n_tg;
    if (nkeys <= 2049 && n <= 131072 && n > 0) {
        int* tmp = scratch + 2 * nkeys + 2 * (nkeys + 1);
        const size_t lds = (size_t)(2 * nkeys + 2 * (nkeys + 1) + IP1_T) * sizeof(int);
        hipLaunchKernelGGL(k_ip_one, dim3(1), dim3(IP1_T), lds, (hipStream_t)stream, seq, n_sp, lab, n_tg, gran, nkeys, tmp, tmp + n, sp_ids,
                           sp_rows, sp_start, tg_ids, tg_rows, tg_start, rec, 18);
        HIP_LAUNCH_CHECK();
        return 0;
    }
    int rc = ader_sparse_lists(seq, n_sp, lab, n_tg, N, scratch, sp_ids, sp_rows, sp_start, tg_ids, tg_rows, tg_start, stream);
    if (rc || !rec) return rc;
    return ader_tab_tile_meta(sp_ids, sp_rows, sp_start, tg_ids, tg_rows, tg_start, N, rec, stream);
}

}  // extern "C"
