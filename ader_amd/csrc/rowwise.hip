// Row-wise / element-wise kernels of the ADER hot path (HBM- or Infinity-Cache-bound):
// embedding gather + prologue, LayerNorm fwd/bwd, dropout-gradient masks, slab reductions, Adam.
// Reference semantics cited per kernel; gfx950 only.
#include "common.h"
#include "../../include/ader_hip.h"

#define MAXPL 4   // max elements per lane in row kernels: H <= 256

// ---------------------------------------------------------------------------------------------
// x0 = dropout(E0[seq]*sqrt(H) + P[t]) * (seq != 0)
//   reference: modules.py:118-130 (row 0 of the table reads as zeros, items scaled by sqrt(H)),
//   ADER.py:41-60 (positional table, dropout, mask).  One wave per (b,t) row; the table row is read
//   in place (no zero-pad concat copy of the whole table as the TF graph does).
struct __attribute__((packed, aligned(8))) EmbVec { f32x4 v; };      // 16-byte vector at an 8-byte aligned address (rows: 4 H bytes, H even)
// Four rows per wave, all of their pieces in flight before the first use (16 waves per CU x 4 random rows: the shape that reads
// whole rows of a table far larger than the caches at 5.5+ TB/s, MI355X_MICROARCH.md "Indexed rows"); lane l owns the 16-byte piece
// l of a row (H = 150: 37.5 pieces; the half piece and odd H / 4 tails go through the per-element path below).
// H = 150 (the reference's hidden size): the four rows of a wave as 300 8-byte pieces (75 per row) over five wave-instructions --
// every lane busy (the 16-byte form leaves 26 of 64 lanes idle on a 600-byte row), naturally aligned (rows of odd ids start at
// 8 mod 16), and the output streamed with nontemporal stores.  Measured on 409,600 random rows of the 600 MB table (bare gather,
// tools/gather_probe.hip): 3.87 TB/s in the 16-byte form, 4.44 with nontemporal stores, 4.74 packed, **5.43 TB/s packed + nontemporal**
// (0.68 of the HBM peak); eight or sixteen rows per wave: slower.
typedef float f32x2e __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_embed_fwd150(const int* __restrict__ seq, const float* __restrict__ emb,
                                                      const float* __restrict__ pos, float* __restrict__ x, int rows, int T, int V,
                                                      float sqrtH, DropArgs d, int* __restrict__ status) {
    constexpr int H = 150, PR = 75, NP = 4 * PR, NI = (NP + 63) / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * 4;
    if (row0 >= rows) return;
    int myid = 0;
    if (lane < 4 && row0 + lane < rows) {
        myid = seq[row0 + lane];
        if (myid < 0 || myid >= V) { atomicOr(status, ADER_ST_BAD_ID); myid = 0; }
    }
    f32x2e e[NI], q[NI];
    int id[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {          // (row 0 of the table is read for padding ids and ignored: no load under a branch)
        const int p = min(lane + 64 * k, NP - 1);
        const int u = p / PR, c = 2 * (p - u * PR);
        id[k] = __shfl(myid, u, 64);
        const int row = min(row0 + u, rows - 1);
        e[k] = *(const f32x2e*)(emb + (size_t)id[k] * H + c);
        q[k] = *(const f32x2e*)(pos + (size_t)(row % T) * H + c);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = lane + 64 * k;
        const int u = min(p, NP - 1) / PR, c = 2 * (min(p, NP - 1) - u * PR);
        const int row = row0 + u;
        if (p >= NP || row >= rows) continue;
        f32x2e o;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v = (id[k] != 0 ? e[k][j] * sqrtH : 0.0f) + q[k][j];
            v = drop_apply(d, (uint32_t)row * (uint32_t)H + (uint32_t)(c + j), v);
            o[j] = (id[k] != 0) ? v : 0.0f;
        }
        __builtin_nontemporal_store(o, (f32x2e*)(x + (size_t)row * H + c));
    }
}

__global__ __launch_bounds__(256) void k_embed_fwd(const int* __restrict__ seq, const float* __restrict__ emb,
                                                   const float* __restrict__ pos, float* __restrict__ x,
                                                   int rows, int T, int H, int V, float sqrtH, DropArgs d,
                                                   int* __restrict__ status) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gstride = gridDim.x * 16;                         // rows per sweep of the grid (a wave: 4 rows per step)
    int row0 = (blockIdx.x * 4 + wave) * 4;
    if (row0 >= rows) return;
    const bool wide = (H & 1) == 0 && H <= 256;
    const int c = 4 * lane;                                     // first channel of this lane's 16-byte piece
    const bool full = c + 4 <= H, part = !full && c < H;        // H even: a partial piece holds exactly 2 channels
    int idn[4];                                                 // ids of the NEXT step: requested a step ahead (persistent waves)
#pragma unroll
    for (int u = 0; u < 4; ++u) idn[u] = seq[min(row0 + u, rows - 1)];
    for (; row0 < rows; row0 += gstride) {
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int v = row0 + u < rows ? idn[u] : 0;
            if (v < 0 || v >= V) {
                if (lane == 0) atomicOr(status, ADER_ST_BAD_ID);
                v = 0;
            }
            id[u] = v;
        }
        if (row0 + gstride < rows) {
#pragma unroll
            for (int u = 0; u < 4; ++u) idn[u] = seq[min(row0 + gstride + u, rows - 1)];
        }
        if (wide) {
            f32x4 e[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {       // (row 0 of the table is read for padding ids and ignored: no load under a branch)
                const int row = min(row0 + u, rows - 1);
                const float* ep = emb + (size_t)id[u] * H + c;
                const float* pp = pos + (size_t)(row % T) * H + c;
                e[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; q[u] = e[u];
                if (full) { e[u] = ((const EmbVec*)ep)->v; q[u] = ((const EmbVec*)pp)->v; }
                else if (part) { e[u][0] = ep[0]; e[u][1] = ep[1]; q[u][0] = pp[0]; q[u][1] = pp[1]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = row0 + u;
                if (row >= rows || c >= H) continue;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = (id[u] != 0 ? e[u][j] * sqrtH : 0.0f) + q[u][j];
                    v = drop_apply(d, (uint32_t)row * (uint32_t)H + (uint32_t)(c + j), v);
                    o[j] = (id[u] != 0) ? v : 0.0f;
                }
                float* op = x + (size_t)row * H + c;
                if (full) ((EmbVec*)op)->v = o;
                else { op[0] = o[0]; op[1] = o[1]; }
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {       // any H: element by element
                const int row = row0 + u;
                if (row >= rows) break;
                const float* e = emb + (size_t)id[u] * H;
                const float* p = pos + (size_t)(row % T) * H;
                float* o = x + (size_t)row * H;
                for (int cc = lane; cc < H; cc += 64) {
                    float v = (id[u] != 0 ? e[cc] * sqrtH : 0.0f) + p[cc];
                    v = drop_apply(d, (uint32_t)row * (uint32_t)H + (uint32_t)cc, v);
                    o[cc] = (id[u] != 0) ? v : 0.0f;
                }
            }
        }
    }
}

// Backward of the prologue: leaves g = dx0 * mask * keep * scale in place (the gradient w.r.t. the positional rows before the batch
// sum; consumed by the fused table update through an id-sorted list, or by ader_scatter_rows_ordered) -- no atomics.  (The float-atomic
// scatter forms of rounds 1-3, ader_embed_bwd / ader_scatter_rows, are gone: their sums depended on the arrival order.)
__global__ __launch_bounds__(256) void k_embed_bwd_rows(const int* __restrict__ seq, float* __restrict__ dx, int rows, int H, int V,
                                                        DropArgs d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    int id = seq[row];
    if (id < 0 || id >= V) id = 0;
    float* g = dx + (size_t)row * H;
    for (int c = lane; c < H; c += 64) {
        float v = 0.0f;
        if (id != 0) {
            v = g[c];
            if (d.thr != 0) v = drop_keep(d, (uint32_t)row * (uint32_t)H + (uint32_t)c) ? v * d.scale : 0.0f;
        }
        g[c] = v;
    }
}

// EWC baseline (reference EWC.py:115-164).  Fisher accumulation: F += scale * g * g  (EWC.py:160-163: squared per-sample
// gradients, divided by the sample count at the end).
__global__ __launch_bounds__(256) void k_sq_accum(const float* __restrict__ g, float* __restrict__ F, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = g[i];
        F[i] += scale * x * x;
    }
}
// Penalty lambda/2 * sum F (theta - theta_prev)^2 (EWC.py:121-124): grad += lambda * F * (theta - theta_prev); per-block
// partial sums of the penalty go to part[blockIdx.x] (summed in a fixed order by k_sum_add: deterministic).
__global__ __launch_bounds__(256) void k_ewc_penalty(const float* __restrict__ theta, const float* __restrict__ prev,
                                                     const float* __restrict__ F, float* __restrict__ grad, size_t n, float lam,
                                                     float* __restrict__ part) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = theta[i] - prev[i], f = F[i];
        grad[i] += lam * f * d;
        acc += f * d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_sum_add(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] += scale * red[0];
}

// demb[id] += rows * scale without atomics: entries bucketed by id (ader_sparse_lists: buckets of `gran` ids in id order, inside a bucket in POSITION
// order).  One workgroup per bucket, thread c owns column c and adds the bucket's rows one after the other: the sum of a table row is
// formed in position order -- bit-reproducible, and bit-IDENTICAL on every rank that holds the same gathered rows (the float
// atomics of k_scatter_rows arrive in a run- and rank-dependent order, which let data-parallel replicas drift apart).
__global__ __launch_bounds__(192) void k_scatter_rows_ordered(const int* __restrict__ ids, const int* __restrict__ rws,
                                                              const int* __restrict__ start, const float* __restrict__ rows,
                                                              float* __restrict__ demb, int H, int V, float scale) {
    const int k0 = start[blockIdx.x], k1 = start[blockIdx.x + 1];
    for (int c = threadIdx.x; c < H; c += 192) {
        for (int k = k0; k < k1; ++k) {                     // list order = position order inside the bucket
            const int id = ids[k];
            if (id <= 0 || id >= V) continue;
            demb[(size_t)id * H + c] += rows[(size_t)rws[k] * H + c] * scale;
        }
    }
}

// dpos[t][c] = sum_b g[b*T + t][c].  Each workgroup owns 32 consecutive (t,c) outputs; 8 b-slices accumulate
// sequentially and are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_pos_grad(const float* __restrict__ g, float* __restrict__ dpos, int B, int T, int H) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const int TH = T * H;
    float acc = 0.0f;
    if (i < TH) {
        const int per = (B + 7) / 8;
        const int b0 = sl * per, b1 = min(B, b0 + per);
#pragma unroll 4
        for (int b = b0; b < b1; ++b) acc += g[(size_t)b * TH + i];
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < TH) {
        float a = red[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) a += red[k][o];
        dpos[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward, reference modules.py:44-48: biased variance, eps inside the sqrt,
//   y = gamma * ((x-mean)/sqrt(var+eps)) + beta.   One wave per row.
// Optional outputs: xnz = sign(|sum_c x|), ynz = sign(|sum_c y|) -- the key / query masks of
// multihead_attention (modules.py:188, 208) when called on the attention block input.
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, long x_rs, float* __restrict__ y, long y_rs,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ mean_o, float* __restrict__ std_o,
                                                float* __restrict__ xnz, float* __restrict__ ynz, int rows, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * x_rs;
    float v[MAXPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < H) ? xr[c] : 0.0f;
        s += v[i];
    }
    s = wave_sum(s);
    const float mean = s / (float)H;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        const float dlt = (c < H) ? (v[i] - mean) : 0.0f;
        q += dlt * dlt;
    }
    q = wave_sum(q);
    const float sd = sqrtf(q / (float)H + LN_EPS);
    float* yr = y + (size_t)row * y_rs;
    float ys = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        if (c < H) {
            const float o = gamma[c] * ((v[i] - mean) / sd) + beta[c];
            yr[c] = o;
            ys += o;
        }
    }
    if (ynz) ys = wave_sum(ys);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (std_o) std_o[row] = sd;
        if (xnz) xnz[row] = (s != 0.0f) ? 1.0f : 0.0f;
        if (ynz) ynz[row] = (ys != 0.0f) ? 1.0f : 0.0f;
    }
}

// LayerNorm backward.  dx = (1/sd) * (dxh - mean(dxh) - xh*mean(dxh*xh)), dxh = dy*gamma; dx (+)= add.
// Per-WG partial sums of dgamma = sum dy*xh and dbeta = sum dy go to slab[WG][2][H] (deterministic
// reduction by k_reduce_slabs).  Grid-stride over rows.
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, long dy_rs, const float* __restrict__ x, long x_rs,
                                                const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                const float* __restrict__ std_i, const float* __restrict__ add, long add_rs,
                                                float* __restrict__ dx, long dx_rs, float* __restrict__ slab, int rows, int H) {
    __shared__ float red[4][2][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float dg[MAXPL], db[MAXPL], gm[MAXPL];
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        dg[i] = 0.0f; db[i] = 0.0f;
        const int c = lane + 64 * i;
        gm[i] = (c < H) ? gamma[c] : 0.0f;
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* dyr = dy + (size_t)row * dy_rs;
        const float* xr = x + (size_t)row * x_rs;
        const float mean = mean_i[row], sd = std_i[row];
        float xh[MAXPL], dxh[MAXPL];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXPL; ++i) {
            const int c = lane + 64 * i;
            float g = 0.0f;
            xh[i] = 0.0f;
            if (c < H) { g = dyr[c]; xh[i] = (xr[c] - mean) / sd; }
            dxh[i] = g * gm[i];
            s1 += dxh[i];
            s2 += dxh[i] * xh[i];
            dg[i] += g * xh[i];
            db[i] += g;
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        float* dxr = dx + (size_t)row * dx_rs;
#pragma unroll
        for (int i = 0; i < MAXPL; ++i) {
            const int c = lane + 64 * i;
            if (c < H) {
                float o = (dxh[i] - s1 - xh[i] * s2) / sd;
                if (add) o += add[(size_t)row * add_rs + c];
                dxr[c] = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) { red[wave][0][lane + 64 * i] = dg[i]; red[wave][1][lane + 64 * i] = db[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        const float a = ((red[0][0][c] + red[1][0][c]) + red[2][0][c]) + red[3][0][c];
        const float b = ((red[0][1][c] + red[1][1][c]) + red[2][1][c]) + red[3][1][c];
        slab[((size_t)blockIdx.x * 2 + 0) * H + c] = a;
        slab[((size_t)blockIdx.x * 2 + 1) * H + c] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward entry of the FFN tail  x2 = (dropout(h2) + y) * mask   (modules.py:262-266, ADER.py:80):
//   g = dx2 * mask (gradient of the residual y);  dh2 = g * keep * scale.
__global__ __launch_bounds__(256) void k_mask_dropgrad(const float* __restrict__ dx2, const int* __restrict__ seq,
                                                       float* __restrict__ g, float* __restrict__ dh2, int rows, int H,
                                                       int row_mul, int row_add, DropArgs d) {
    const size_t n = (size_t)rows * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / H);
        const int c = (int)(i - (size_t)row * H);
        const int rf = row * row_mul + row_add;               // row of the full [B*T,H] tensor
        float v = (seq[rf] != 0) ? dx2[i] : 0.0f;
        g[i] = v;
        if (d.thr != 0) v = drop_keep(d, (uint32_t)rf * (uint32_t)H + (uint32_t)c) ? v * d.scale : 0.0f;
        dh2[i] = v;
    }
}

// dst[(r*row_mul + row_add), :] += src[r, :]
__global__ __launch_bounds__(256) void k_add_rows(const float* __restrict__ src, float* __restrict__ dst, int rows, int H,
                                                  int row_mul, int row_add) {
    const size_t n = (size_t)rows * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / H);
        const int c = (int)(i - (size_t)row * H);
        dst[((size_t)row * row_mul + row_add) * H + c] += src[i];
    }
}

// dst[r][c] = sum_s src[s*slab_stride + r*ld + c].  Rows r < n_rows go to dst, row n_rows (if dst_extra) goes to
// dst_extra -- the "ones row" of the weight-gradient GEMM = bias gradient.  A workgroup owns 64 outputs; 16 thread
// groups sum interleaved slab subsets (s = k mod 16, ascending, 8 loads in flight) and are combined in a fixed order:
// deterministic, and the serial chain per thread is S/16 loads instead of S (the reads are latency-bound).
#define RS_G 16
#define RS_O 16     // outputs per workgroup: 256-thread workgroups of <= 64 registers slot into any CU next to the table update
__global__ __launch_bounds__(RS_G * RS_O) void k_reduce_slabs(const float* __restrict__ src, long slab_stride, int S, int ld,
                                                       int n_rows, int n_cols, float* __restrict__ dst,
                                                       float* __restrict__ dst_extra) {
    __shared__ float red[RS_G][RS_O];
    const int total = (n_rows + (dst_extra ? 1 : 0)) * n_cols;
    const int o = threadIdx.x % RS_O, sg = threadIdx.x / RS_O;
    const int i = blockIdx.x * RS_O + o;
    float acc = 0.0f;
    int r = 0, c = 0;
    if (i < total) {
        r = i / n_cols; c = i - r * n_cols;
        const float* p = src + (size_t)r * ld + c;
#pragma unroll 8
        for (int s = sg; s < S; s += RS_G) acc += p[(size_t)s * slab_stride];
    }
    red[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && i < total) {
        float a = red[0][o];
#pragma unroll
        for (int k = 1; k < RS_G; ++k) a += red[k][o];
        if (r < n_rows) dst[(size_t)r * n_cols + c] = a;
        else dst_extra[c] = a;
    }
}

// Several such reductions in ONE launch (blockIdx.y = job): the LayerNorm gamma / beta partials of all blocks of a backward pass.
#define RS_MAXJOBS 8
struct ReduceJobs {
    const float* src[RS_MAXJOBS]; float* dst[RS_MAXJOBS]; float* dst_extra[RS_MAXJOBS];
    long slab_stride[RS_MAXJOBS]; int S[RS_MAXJOBS], ld[RS_MAXJOBS], n_rows[RS_MAXJOBS], n_cols[RS_MAXJOBS];
};
__global__ __launch_bounds__(RS_G * RS_O) void k_reduce_slabs_batch(ReduceJobs j) {
    __shared__ float red[RS_G][RS_O];
    const int y = blockIdx.y;
    const int n_rows = j.n_rows[y], n_cols = j.n_cols[y], S = j.S[y];
    float* dst_extra = j.dst_extra[y];
    const int total = (n_rows + (dst_extra ? 1 : 0)) * n_cols;
    const int o = threadIdx.x % RS_O, sg = threadIdx.x / RS_O;
    const int i = blockIdx.x * RS_O + o;
    float acc = 0.0f;
    int r = 0, c = 0;
    if (i < total) {
        r = i / n_cols; c = i - r * n_cols;
        const float* p = j.src[y] + (size_t)r * j.ld[y] + c;
#pragma unroll 8
        for (int s = sg; s < S; s += RS_G) acc += p[(size_t)s * j.slab_stride[y]];
    }
    red[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && i < total) {
        float a = red[0][o];
#pragma unroll
        for (int k = 1; k < RS_G; ++k) a += red[k][o];
        if (r < n_rows) j.dst[y][(size_t)r * n_cols + c] = a;
        else dst_extra[c] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Dense Adam over one flat parameter buffer, TF ApplyAdam semantics (ADER.py:96):
//   m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= (m*lr_t)/(sqrt(v)+eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (host).
// HBM-bound: 4 streams in, 3 out, 16 B per lane per access.  Optionally refreshes the bf16 shadow of the item table
// (first `table_elems` parameters, rows of H elements -> shadow rows of 168 bf16) that the bf16 logit GEMMs stream.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define ADAM_UNROLL 4
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                              const float* __restrict__ g, size_t n, float lr_t, float omb1, float omb2, float eps,
                                              __bf16* __restrict__ shadow, size_t table_elems, int H) {
    const size_t n4 = n >> 2;
    float4* p4 = (float4*)p; float4* m4 = (float4*)m; float4* v4 = (float4*)v; const float4* g4 = (const float4*)g;
    // each workgroup walks contiguous spans of 256*ADAM_UNROLL float4; all loads of a span are issued before any math
    const size_t span = (size_t)256 * ADAM_UNROLL;
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
        float4 pp[ADAM_UNROLL], mm[ADAM_UNROLL], vv[ADAM_UNROLL], gg[ADAM_UNROLL];
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n4) {
                pp[u] = p4[i]; mm[u] = m4[i]; vv[u] = v4[i];
                const f32x4 t_ = __builtin_nontemporal_load((const f32x4*)g4 + i);   // the gradient is dead after this read
                gg[u] = make_float4(t_[0], t_[1], t_[2], t_[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i >= n4) continue;
#define ADAM1(f) mm[u].f += (gg[u].f - mm[u].f) * omb1; vv[u].f += (gg[u].f * gg[u].f - vv[u].f) * omb2; \
                 pp[u].f -= (mm[u].f * lr_t) / (sqrtf(vv[u].f) + eps);
            ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
            p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u];
            const size_t e0 = i << 2;
            if (shadow && e0 < table_elems) {           // H even: a pair of consecutive elements never straddles a table row
                const size_t r0 = e0 / H, r1 = (e0 + 2) / H;
                bf16x2_t a; a[0] = (__bf16)pp[u].x; a[1] = (__bf16)pp[u].y;
                bf16x2_t b; b[0] = (__bf16)pp[u].z; b[1] = (__bf16)pp[u].w;
                *(bf16x2_t*)(shadow + r0 * 168 + (e0 - r0 * H)) = a;
                if (e0 + 2 < table_elems) *(bf16x2_t*)(shadow + r1 * 168 + (e0 + 2 - r1 * H)) = b;
            }
        }
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mm = m[i], vv = v[i];
        const float gg = g[i];
        mm += (gg - mm) * omb1; vv += (gg * gg - vv) * omb2;
        p[i] -= (mm * lr_t) / (sqrtf(vv) + eps);
        m[i] = mm; v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, size_t n, float val) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = val;
}

// ============================================================================================= C ABI
static inline int cap_grid(size_t n, int per_block, int cap) {
    size_t g = (n + per_block - 1) / per_block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int ader_embed_fwd(const int* seq, const float* emb, const float* pos, float* x, int rows, int T, int H, int V,
                   const AderDrop* drop, int* status, void* stream) {
    if (rows <= 0) return 0;
    // 16 rows per workgroup.  (Persistent waves -- 8 workgroups per CU sweeping the rows with the next step's ids requested a step
    // ahead -- measured SLOWER: 175 vs 150-158 us for 409,600 rows; short-lived waves keep more rows in flight.)
    if (H == 150) hipLaunchKernelGGL(k_embed_fwd150, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, seq, emb, pos, x, rows, T, V,
                                     sqrtf((float)H), drop_from(drop), status);
    else hipLaunchKernelGGL(k_embed_fwd, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, seq, emb, pos, x, rows, T, H, V,
                            sqrtf((float)H), drop_from(drop), status);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_sq_accum(const float* g, float* F, size_t n, float scale, void* stream) {
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_sq_accum, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, F, n, scale);
    HIP_LAUNCH_CHECK();
    return 0;
}

// loss[0] += lambda/2 * sum F (theta - prev)^2;  grad += lambda * F * (theta - prev).  part: 1024 floats of scratch.
int ader_ewc_penalty(const float* theta, const float* prev, const float* F, float* grad, size_t n, float lambda_, float* part,
                     float* loss, void* stream) {
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_ewc_penalty, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, theta, prev, F, grad, n, lambda_, part);
    hipLaunchKernelGGL(k_sum_add, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, (int)blocks, 0.5f * lambda_, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// demb[ids[k]] += rows[rws[k]] * scale in list order, bucket by bucket (lists from ader_sparse_lists over the gathered ids; nb =
// number of buckets): deterministic -- every table row is summed in position order.
int ader_scatter_rows_ordered(const int* ids, const int* rws, const int* start, int nb, const float* rows, int H, int V, float scale,
                              float* demb, void* stream) {
    if (nb <= 0) return 0;
    hipLaunchKernelGGL(k_scatter_rows_ordered, dim3(nb), dim3(192), 0, (hipStream_t)stream, ids, rws, start, rows, demb, H, V, scale);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_embed_bwd_rows(const int* seq, float* dx, float* dpos, int B, int T, int H, int V, const AderDrop* drop, void* stream) {
    const int rows = B * T;
    if (rows <= 0) return 0;
    if (seq)     // seq == NULL: dx already holds the masked / dropout-scaled rows (ader_seq_bwd_qkv with emb_bwd)
        hipLaunchKernelGGL(k_embed_bwd_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, seq, dx, rows, H, V,
                           drop_from(drop));
    hipLaunchKernelGGL(k_pos_grad, dim3((T * H + 31) / 32), dim3(256), 0, (hipStream_t)stream, dx, dpos, B, T, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_ln_fwd(const float* x, long x_rs, float* y, long y_rs, const float* gamma, const float* beta, float* mean_o,
                float* std_o, float* xnz, float* ynz, int rows, int H, void* stream) {
    if (rows <= 0) return 0;
    if (H > 64 * MAXPL) return -2;
    hipLaunchKernelGGL(k_ln_fwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, x_rs, y, y_rs, gamma, beta, mean_o,
                       std_o, xnz, ynz, rows, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

// slab must hold ader_ln_bwd_slabs(rows) * 2 * H floats; dgamma/dbeta receive the reduced sums.
int ader_ln_bwd_slabs(int rows) { return cap_grid((size_t)rows, 16, 1024); }

int ader_ln_bwd(const float* dy, long dy_rs, const float* x, long x_rs, const float* gamma, const float* mean_i,
                const float* std_i, const float* add, long add_rs, float* dx, long dx_rs, float* slab, float* dgamma,
                float* dbeta, int rows, int H, void* stream) {
    if (rows <= 0) return 0;
    if (H > 64 * MAXPL) return -2;
    const int G = ader_ln_bwd_slabs(rows);
    hipLaunchKernelGGL(k_ln_bwd, dim3(G), dim3(256), 0, (hipStream_t)stream, dy, dy_rs, x, x_rs, gamma, mean_i, std_i, add, add_rs,
                       dx, dx_rs, slab, rows, H);
    // slab layout [G][2][H]: row 0 = dgamma partial, row 1 = dbeta partial.  dgamma == NULL: the caller reduces the slabs later
    // (ader_reduce_slabs(slab, 2 H, ader_ln_bwd_slabs(rows), H, 1, H, dgamma, dbeta) -- off the critical path of the backward pass)
    if (dgamma)
        hipLaunchKernelGGL(k_reduce_slabs, dim3((2 * H + RS_O - 1) / RS_O), dim3(RS_G * RS_O), 0, (hipStream_t)stream, slab, (long)2 * H, G, H, 1, H,
                           dgamma, dbeta);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_mask_dropgrad(const float* dx2, const int* seq, float* g, float* dh2, int rows, int H, int row_mul, int row_add,
                       const AderDrop* drop, void* stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_mask_dropgrad, dim3(cap_grid((size_t)rows * H, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dx2, seq, g,
                       dh2, rows, H, row_mul, row_add, drop_from(drop));
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_add_rows(const float* src, float* dst, int rows, int H, int row_mul, int row_add, void* stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_add_rows, dim3(cap_grid((size_t)rows * H, 256, 2048)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, H,
                       row_mul, row_add);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_reduce_slabs(const float* src, long slab_stride, int S, int ld, int n_rows, int n_cols, float* dst, float* dst_extra,
                      void* stream) {
    const int total = (n_rows + (dst_extra ? 1 : 0)) * n_cols;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((total + RS_O - 1) / RS_O), dim3(RS_G * RS_O), 0, (hipStream_t)stream, src, slab_stride, S, ld, n_rows,
                       n_cols, dst, dst_extra);
    HIP_LAUNCH_CHECK();
    return 0;
}

// n <= 8 reductions of ader_reduce_slabs in one launch (same arithmetic and summation order per job)
int ader_reduce_slabs_batch(const float* const* src, const long* slab_stride, const int* S, const int* ld, const int* n_rows,
                            const int* n_cols, float* const* dst, float* const* dst_extra, int n, void* stream) {
    if (n <= 0) return 0;
    if (n > RS_MAXJOBS) return -2;
    ReduceJobs j = {};
    int maxtot = 0;
    for (int y = 0; y < n; ++y) {
        j.src[y] = src[y]; j.dst[y] = dst[y]; j.dst_extra[y] = dst_extra[y]; j.slab_stride[y] = slab_stride[y];
        j.S[y] = S[y]; j.ld[y] = ld[y]; j.n_rows[y] = n_rows[y]; j.n_cols[y] = n_cols[y];
        const int total = (n_rows[y] + (dst_extra[y] ? 1 : 0)) * n_cols[y];
        if (total > maxtot) maxtot = total;
    }
    if (maxtot <= 0) return 0;
    hipLaunchKernelGGL(k_reduce_slabs_batch, dim3((maxtot + RS_O - 1) / RS_O, n), dim3(RS_G * RS_O), 0, (hipStream_t)stream, j);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_adam_step(float* p, float* m, float* v, const float* g, size_t n, float lr_t, float beta1, float beta2, float eps,
                   void* shadow, size_t table_elems, int H, void* stream) {
    if (n == 0) return 0;
    if (shadow && ((H & 1) || (table_elems & 1))) return -2;
    hipLaunchKernelGGL(k_adam, dim3(cap_grid(n / 4 + 1, 256 * ADAM_UNROLL, 2048)), dim3(256), 0, (hipStream_t)stream, p, m, v, g, n, lr_t,
                       1.0f - beta1, 1.0f - beta2, eps, (__bf16*)shadow, table_elems, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_fill(float* p, size_t n, float val, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_fill, dim3(cap_grid(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, n, val);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Catalog-sharded data parallelism (engine._train_step_catalog): the table rows a step's inputs need travel between ranks.
// k_gather_owned : out[p][:] = table[ids[p]][:] if lo < ids[p] <= hi (this rank owns the row) else 0      (sender side)
// k_scatter_owned: position p of MY inputs holds id = ids[p], owned by rank (id-1)/shard: take that rank's slice of the
//                  received buffer recv[owner][p][:] and write it to table[id] (p < n_tab) or to extra[p - n_tab] (labels)
__global__ __launch_bounds__(256) void k_gather_owned(const float* __restrict__ table, const int* __restrict__ ids, int n, int H,
                                                      int lo, int hi, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= n) return;
    const int id = ids[p];
    const bool own = id > lo && id <= hi;
    const float* src = table + (size_t)id * H;
    float* dst = out + (size_t)p * H;
    for (int c = lane; c < H; c += 64) dst[c] = own ? src[c] : 0.0f;
}

__global__ __launch_bounds__(256) void k_scatter_owned(const float* __restrict__ recv, const int* __restrict__ ids, int n, int n_tab,
                                                       int H, int shard, int world, float* __restrict__ table,
                                                       float* __restrict__ extra) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= n) return;
    const int id = ids[p];
    if (id <= 0) {                                                  // padding position: nothing to place; label rows read zeros
        if (p >= n_tab) for (int c = lane; c < H; c += 64) extra[(size_t)(p - n_tab) * H + c] = 0.0f;
        return;
    }
    int owner = (id - 1) / shard;
    if (owner >= world) owner = world - 1;
    const float* src = recv + ((size_t)owner * n + p) * H;
    float* dst = (p < n_tab) ? table + (size_t)id * H : extra + (size_t)(p - n_tab) * H;
    for (int c = lane; c < H; c += 64) dst[c] = src[c];
}

extern "C" {

int ader_gather_owned(const float* table, const int* ids, int n, int H, int lo, int hi, float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gather_owned, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, n, H, lo, hi, out);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_scatter_owned(const float* recv, const int* ids, int n, int n_tab, int H, int shard, int world, float* table, float* extra,
                       void* stream) {
    if (n <= 0) return 0;
    if (shard <= 0 || world <= 0 || n_tab > n) return -2;
    hipLaunchKernelGGL(k_scatter_owned, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, recv, ids, n, n_tab, H, shard, world,
                       table, extra);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
