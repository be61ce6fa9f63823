// Packing plan of the packed session kernels (seqp_fwd.hip, seqp_bwd.hip): two small launches.
//
// The reference left-pads every session to maxlen (util.py:161-169); on the shipped splits ~90 % of the positions are padding.
// A padded position influences no real one -- its key is masked (modules.py:188-193), its outputs are re-zeroed (ADER.py:80),
// its gradient is exactly zero -- so the session kernels may drop those rows altogether.  This file lays the REAL positions
// of a batch out in 64-row tiles: several short sessions share a tile (attention becomes block-diagonal inside it), a tile never
// splits a session, sessions keep their batch order inside a tile.  The activations of the packed kernels live in this layout
// ([tile*64 + row, H]); only the representations [B,H], the compact tensors of a pruned last block [B,..] and the gradient rows
// of the input embeddings [B*T,H] stay session-indexed.
//
// Packing rule (no sequential dependency between sessions, so it is two scans): the SHORT sessions (<= 16 positions: 97 % of the
// shipped data) form one stream of rows in batch order and a session goes to the tile floor(start / w) of the stream offset it
// starts at; because a session is shorter than 64 - w + 1 the rows that land in a tile never exceed 64 (16 <= w <= 49).  w
// shrinks when the batch would otherwise fill fewer tiles than the chip has CUs (`target`): the session kernels are latency-bound
// per tile, a half-filled tile is faster than a full one, and an idle CU is worth nothing -- at the default w = 17 a tile holds at
// most 32 rows, which is what the forward's 16-column mapping is for.  Every longer session gets a tile of its own (17..32
// positions: still a small tile; a tile with more rows takes the 32x32 mapping and, being the slowest workgroup, sets the launch's
// duration -- rounds 5a's medium class, two or three such sessions per 64-row tile, put one in nearly every batch).
//
// Launch 1 (k_plan_len, one workgroup per 64 sessions): the first item of every session (its ids read once, coalesced; an LDS
// atomicMin over the positions) -> slen; the workgroup whose ticket is last then runs the scans for the whole batch (release fence
// + acq_rel ticket + acquire fence, see the hand-off comment in the kernel) -> srow0, tile_rows, hdr.  Launch 2 (k_plan_rows, one thread per position): the per-row records.
// (The first form of this file was ONE 1,024-thread workgroup: 22-32 us for 400-600 sessions -- a single CU reads the batch's
// 120 KB of ids at ~10 B/clock, twice.)
//
// Outputs (AderSeqPack): hdr = {tiles, 64 tiles, real positions, w, -, -, -, ticket}; per tile its row count; per packed row the
// item id, the local position b*T + t, the GLOBAL position (dropout counters: a data-parallel rank draws the masks of its global
// rows, include/ader_hip.h AderDrop) and info = first row of its session in the tile | last-row flag << 6 | t << 8 | b << 16;
// per session its first packed row and its length.  An all-padding session keeps position T-1 (one row of id 0), so that every
// session has a last row.  gfx950 only.
#include "common.h"
#include "../../include/ader_hip.h"

#define PLAN_MAXB 4096
#define PLAN_THREADS 256
#define PLAN_SPW 64                 // sessions per workgroup of the length pass

#ifdef PLAN_STAMP     // diagnostic build only (tools/build_variant.sh ... -DPLAN_STAMP): clocks per phase of thread 0 of the last workgroup
__device__ unsigned long long plan_dbg[16];
#define PST_INIT unsigned long long tprev_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_) :: "memory");
#define PST(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  if (threadIdx.x == 0) plan_dbg[k_] = t_ - tprev_; tprev_ = t_; }
extern "C" int ader_dbg_read_plan(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(plan_dbg), (size_t)n * 8); }
#else
#define PST_INIT
#define PST(k_)
#endif

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// exclusive scan of one int per thread over the workgroup; returns the exclusive prefix, *total = sum.  tmp: PLAN_THREADS / 64 ints of LDS
__device__ __forceinline__ int block_excl_scan(int v, int* tmp, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int inc = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PLAN_THREADS / 64; ++w) {
        const int x = tmp[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(PLAN_THREADS) void k_plan_len(const int* __restrict__ seq, int B, int T, int w1_min, int w1_max, int target,
                                                          AderSeqPack o) {
    __shared__ int fz_l[PLAN_SPW];
    __shared__ int last_l;
    __shared__ unsigned char len_l[PLAN_MAXB];
    __shared__ int first_l[PLAN_MAXB], end_l[PLAN_MAXB];
    __shared__ unsigned short tile_l[PLAN_MAXB];
    __shared__ int off_l[PLAN_MAXB];
    __shared__ int tmp[PLAN_THREADS / 64];
    const int tid = threadIdx.x;
    PST_INIT
    // ---- lengths of this workgroup's sessions
    const int sb = blockIdx.x * PLAN_SPW, ns = min(PLAN_SPW, B - sb);
    const int n = ns * T;                                        // <= 4096 ids: sixteen loads per thread, all in flight
    if (tid < PLAN_SPW) fz_l[tid] = T - 1;                       // first real position (T-1 for an all-padding session)
    __syncthreads();
    {
        const int* __restrict__ p = seq + (size_t)sb * T;
        int v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = tid + u * PLAN_THREADS; v[u] = (i < n) ? p[i] : 0; }
        const int ds = PLAN_THREADS / T, dt = PLAN_THREADS - ds * T;
        int sI = tid / T, tI = tid - sI * T;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (v[u] != 0) atomicMin(&fz_l[sI], tI);
            sI += ds; tI += dt;
            if (tI >= T) { tI -= T; ++sI; }
        }
    }
    __syncthreads();
    // Hand-off to the last-arriving workgroup, in the memory model's own terms (the classic fence + ticket reduction): every storing
    // thread's release fence at agent scope orders its slen stores before anything that follows the barrier; the ticket is one
    // acq_rel read-modify-write at agent scope per workgroup, so the chain of tickets carries every earlier workgroup's release to the
    // last arriver, whose acquire fence then makes those stores visible to all of its threads behind the barrier.  (Rounds 4-5 relied
    // on relaxed atomics + a hand-written s_waitcnt vmcnt(0) and on what the caches happen to do; k_tabp's cold-process corruption was
    // a hand-off of that kind.  The fences cost ~1 us per launch; tests/stress_handoffs.py hammers this launch cold and warm.)
    if (tid < ns) __hip_atomic_store(o.slen + sb + tid, T - fz_l[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) last_l = (__hip_atomic_fetch_add(o.hdr + 7, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last_l) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    PST(0)
    // ================= the last workgroup: scans over all sessions
    for (int i = tid; i < B; i += PLAN_THREADS) len_l[i] = (unsigned char)__hip_atomic_load(o.slen + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < B; i += PLAN_THREADS) { first_l[i] = 0x7fffffff; end_l[i] = 0; }
    __syncthreads();
    // ---- stream offsets per class (thread t owns sessions [t spt, (t+1) spt))
    const int spt = (B + PLAN_THREADS - 1) / PLAN_THREADS;
    const int s0 = tid * spt, s1 = min(B, s0 + spt);
    int a1 = 0, a3 = 0, tot = 0;
    for (int s = s0; s < s1; ++s) {
        const int ln = len_l[s];
        tot += ln;
        if (ln <= 16) a1 += ln; else a3 += 1;
    }
    PST(1)
    int P1, C3, TOT;
    int e1 = block_excl_scan(a1, tmp, &P1);
    int e3 = block_excl_scan(a3, tmp, &C3);
    block_excl_scan(tot, tmp, &TOT);
    int w1 = w1_max;
    if (target > 0) w1 = max(w1_min, min(w1_max, (P1 + target - 1) / target));
    w1 = max(16, min(49, w1));          // (>= 16: every window then holds a session start, so the raw tile count never exceeds B)
    PST(2)
    const int n1 = P1 > 0 ? (P1 - 1) / w1 + 1 : 0;
    // (every window [k w, (k+1) w) below the stream's last start holds at least one session start, because no session is longer than w;
    //  the window that holds the stream's end may hold none: the tile numbering is compacted below)
    for (int s = s0; s < s1; ++s) {
        const int ln = len_l[s];
        int tile, st;
        if (ln <= 16) { st = e1; tile = e1 / w1; e1 += ln; }
        else { st = 0; tile = n1 + e3; e3 += 1; }
        tile_l[s] = (unsigned short)tile; off_l[s] = st;
        atomicMin(&first_l[tile], st);
        atomicMax(&end_l[tile], st + ln);
    }
    __syncthreads();
    PST(3)
    // ---- compact the tile numbering (a window without a session start is no tile)
    const int nt_raw = n1 + C3;
    int cnt = 0;
    const int tpt = (nt_raw + PLAN_THREADS - 1) / PLAN_THREADS;
    const int t0 = tid * tpt, t1 = min(nt_raw, t0 + tpt);
    for (int t = t0; t < t1; ++t) cnt += (end_l[t] > 0) ? 1 : 0;
    int ntiles;
    int tb = block_excl_scan(cnt, tmp, &ntiles);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {             // end_l[raw tile] becomes its compact index
        const int e = end_l[t];
        if (e > 0) {
            o.tile_rows[tb] = e - first_l[t];
            end_l[t] = tb++;
        } else end_l[t] = -1;
    }
    __syncthreads();
    PST(4)
    // ---- per session: first packed row
    for (int sx = s0; sx < s1; ++sx) {
        const int raw = tile_l[sx];
        o.srow0[sx] = end_l[raw] * 64 + off_l[sx] - first_l[raw];
    }
    if (tid == 0) { o.hdr[0] = ntiles; o.hdr[1] = ntiles * 64; o.hdr[2] = TOT; o.hdr[3] = w1; o.hdr[7] = 0; }     // (ticket re-armed for the next launch)
    PST(5)
}

// position t of session s is packed row srow0[s] + t - (T - slen[s]) when t >= T - slen[s]
__global__ __launch_bounds__(256) void k_plan_rows(const int* __restrict__ seq, int B, int T, int row0, int split_rows, int row0_ex,
                                                   AderSeqPack o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    const int s = i / T, t = i - s * T;
    const int id = seq[i], ln = o.slen[s], p0 = o.srow0[s];
    const int k = t - (T - ln);
    if (k < 0) return;
    const int pr = p0 + k;
    const int gb = (split_rows >= 0 && s >= split_rows) ? s - split_rows + row0_ex : s + row0;
    o.ids[pr] = id;
    o.lpos[pr] = i;
    o.gpos[pr] = (unsigned)gb * (unsigned)T + (unsigned)t;
    o.info[pr] = (p0 & 63) | ((k == ln - 1) ? 64 : 0) | (t << 8) | (s << 16);
}

extern "C" {

int ader_seq_pack_plan(const int* seq, int B, int T, int row0, int split_rows, int row0_ex, int w1_min, int w1_max, int target,
                       const AderSeqPack* out, void* stream) {
    if (B <= 0) return 0;
    if (B > PLAN_MAXB || T < 1 || T > 64 || !out) return -2;
    hipLaunchKernelGGL(k_plan_len, dim3((B + PLAN_SPW - 1) / PLAN_SPW), dim3(PLAN_THREADS), 0, (hipStream_t)stream, seq, B, T, w1_min, w1_max,
                       target, *out);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_plan_rows, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, seq, B, T, row0, split_rows, row0_ex, *out);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
