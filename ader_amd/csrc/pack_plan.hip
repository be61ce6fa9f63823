// Bookkeeping of the catalog-sharded data-parallel step's PACKED row exchange (Engine._train_step_catalog) as ONE
// launch.  Rank r owns table rows [1 + r S, (r + 1) S]; every rank holds the gathered ids of all ranks' positions ids_g [W][n_all]
// (n_pos input positions, then the labels).  Only the rows a position needs travel: the owner sends them to the position's rank
// (uneven all-to-all), and after the backward pass the gradient rows travel the other way.  This kernel derives, from ids_g alone,
//   cnt      [2][W][W]  cnt[0][o][d] = positions of rank d owned by rank o (all positions), cnt[1]: input positions only
//                       (the split sizes of the all-to-alls: every rank computes the same matrix)
//   send_id  [<= W n_all]  ids of the rows THIS rank sends, ordered by (destination, position)
//   ids_back [<= W n_pos]  ids of the gradient rows this rank will receive, same order, input positions only
//   perm     [n_all]       this rank's positions: padding (id 0) first, then grouped by owner, position order inside a group
//                          (= the order in which the received rows arrive)
//   back_src [<= n_pos]    this rank's input positions grouped by owner (= the order in which its gradient rows leave)
// replacing ~20 torch launches (where / cumsum / scatter / index_copy) per step.  Index work: exact.  Replaces nothing of the
// reference (single device, main.py:96); it serves the gather of modules.py:127 and its gradient under data parallelism.
#include "common.h"
#include "../../include/ader_hip.h"

#define PP_T 1024

__device__ __forceinline__ int pp_owner(int id, int S, int W) {       // -1: padding
    if (id <= 0) return -1;
    const int o = (id - 1) / S;
    return o < W ? o : W - 1;
}

// exclusive prefix of one int per thread over the workgroup (1024 threads); returns the total in `tot`
__device__ __forceinline__ int pp_scan(int v, int* part, int& tot) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) part[w] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < PP_T / 64; ++k) { const int p = part[k]; if (k < w) base += p; total += p; }
    __syncthreads();
    tot = total;
    return base + incl - v;
}

// stable compaction of the indices i in [0, n) with pred(i) into out[base + rank] = val(i); returns the count
template <class Pred, class Val>
__device__ __forceinline__ int pp_compact(int n, Pred pred, Val val, long* out, int base, int* part) {
    const int per = (n + PP_T - 1) / PP_T;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += pred(i) ? 1 : 0;
    int tot;
    int slot = base + pp_scan(c, part, tot);
    for (int i = lo; i < hi; ++i) if (pred(i)) out[slot++] = val(i);
    return tot;
}

__global__ __launch_bounds__(PP_T) void k_pack_plan(const int* __restrict__ ids_g, int W, int n_all, int n_pos, int r, int S,
                                                    int* __restrict__ cnt, long* __restrict__ send_id, long* __restrict__ ids_back,
                                                    long* __restrict__ perm, long* __restrict__ back_src) {
    __shared__ int part[PP_T / 64];
    __shared__ int c_l[2 * 16 * 16];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * W * W; i += PP_T) c_l[i] = 0;
    __syncthreads();
    const int n = W * n_all;
    for (int j = t; j < n; j += PP_T) {
        const int d = j / n_all, p = j - d * n_all;
        const int o = pp_owner(ids_g[j], S, W);
        if (o >= 0) {
            atomicAdd(&c_l[o * W + d], 1);
            if (p < n_pos) atomicAdd(&c_l[W * W + o * W + d], 1);
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * W * W; i += PP_T) cnt[i] = c_l[i];
    // rows I send: owned entries in (destination, position) order = flat order of ids_g
    pp_compact(n, [&](int j) { return pp_owner(ids_g[j], S, W) == r; }, [&](int j) { return (long)ids_g[j]; }, send_id, 0, part);
    pp_compact(n, [&](int j) { return pp_owner(ids_g[j], S, W) == r && (j % n_all) < n_pos; }, [&](int j) { return (long)ids_g[j]; },
               ids_back, 0, part);
    // my positions: padding first, then by owner
    const int* mine = ids_g + (size_t)r * n_all;
    int base = 0;
    for (int key = -1; key < W; ++key)
        base += pp_compact(n_all, [&](int p) { return pp_owner(mine[p], S, W) == key; }, [&](int p) { return (long)p; }, perm, base, part);
    base = 0;
    for (int o = 0; o < W; ++o)
        base += pp_compact(n_pos, [&](int p) { return pp_owner(mine[p], S, W) == o; }, [&](int p) { return (long)p; }, back_src, base, part);
}

extern "C" {

// ids_g [W][n_all] int32 (device); outputs as described above (int64 index arrays: send_id / ids_back hold W n_all entries at most,
// perm n_all, back_src n_pos; cnt 2 W W ints).  W <= 16.
int ader_pack_plan(const int* ids_g, int W, int n_all, int n_pos, int rank, int shard_items, int* cnt, long* send_id, long* ids_back,
                   long* perm, long* back_src, void* stream) {
    if (W < 1 || W > 16 || n_all <= 0 || n_pos < 0 || n_pos > n_all || rank < 0 || rank >= W || shard_items <= 0) return -2;
    hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(PP_T), 0, (hipStream_t)stream, ids_g, W, n_all, n_pos, rank, shard_items, cnt, send_id,
                       ids_back, perm, back_src);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
