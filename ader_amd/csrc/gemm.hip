// f32-MFMA row GEMMs of the SASRec blocks (reference modules.py:172-174 dense Q/K/V projections,
// modules.py:254-261 conv1d(k=1) FFN layers) and their backward products.
//   ader_gemm_rows : C[M,H] = epilogue(A[M,H] . W (+bias))      W is [H,H] row-major; TRANS_B uses W^T
//   ader_gemm_atb  : dW[H,H] = A^T . G,  db[H] = colsum(G)       (row-reduction, deterministic slabs)
// v_mfma_f32_16x16x4_f32 is an exact f32 fma chain, so results match an f32 CPU evaluation to
// summation-order effects only.  H <= 160 (padded to 10 column blocks of 16); gfx950 only.
#include "common.h"
#include "../../include/ader_hip.h"

#define HP 160          // padded hidden width (10 MFMA column blocks)
#define LDA 162         // A tile row stride: == 2 (mod 32) -> (row, k) operand reads hit 32 distinct banks
#define LDW 177         // W tile row stride: odd -> transposed staging writes and (k, n) reads are conflict-free
#define LDT 176         // atb tiles: == 16 (mod 32) -> transposed (k, m) reads conflict-free
#define TM 64           // rows per tile

enum { EPI_BIAS = 0, EPI_BIAS_RELU_DROP = 1, EPI_BIAS_DROP_RES_MASK = 2, EPI_RELUDROPGRAD = 3, EPI_ADD = 4 };

struct GemmArgs {
    const float* A; const float* W; const float* bias; float* C;
    const float* aux;      // EPI 2: residual; EPI 3: saved post-dropout activation; EPI 4: addend (may alias C)
    const int* seq;        // EPI 2: row mask source (seq != 0)
    int M, H;
    int row_mul, row_add;  // full-tensor row of local row m = m*row_mul + row_add (dropout counter / seq mask of a row subset)
    DropArgs drop;
};

template <int EPI, bool TRANS_B>
__global__ __launch_bounds__(512) void k_gemm_rows(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W_l = smem;                 // [HP][LDW]
    float* A_l = smem + HP * LDW;      // [TM][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = g.H, ksteps = (H + 3) >> 2;

    for (int i = tid; i < HP * LDW; i += 512) W_l[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < H * H; i += 512) {
        const int r = i / H, c = i - r * H;
        if (TRANS_B) W_l[c * LDW + r] = g.W[i];
        else         W_l[r * LDW + c] = g.W[i];
    }
    const int mw = wave & 3, nw = wave >> 2;       // wave -> 16 rows x 80 columns
    const int n_tiles = (g.M + TM - 1) / TM;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        __syncthreads();                            // previous tile's reads of A_l done (and W_l staged)
        for (int i = tid; i < TM * LDA; i += 512) {
            const int r = i / LDA, c = i - r * LDA;
            const int m = m0 + r;
            A_l[i] = (m < g.M && c < H) ? g.A[(size_t)m * H + c] : 0.0f;
        }
        __syncthreads();
        f32x4 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<5>(A_l + mw * 16 * LDA, LDA, 1, W_l + nw * 80, LDW, 1, ksteps, acc, lane);
        // epilogue: D col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int n = nw * 80 + j * 16 + (lane & 15);
            if (n >= H) continue;
            const float bv = (EPI <= EPI_BIAS_DROP_RES_MASK && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mw * 16 + (lane >> 4) * 4 + r;
                if (m >= g.M) continue;
                const size_t idx = (size_t)m * H + n;
                const int mf = m * g.row_mul + g.row_add;
                const uint32_t didx = (uint32_t)mf * (uint32_t)H + (uint32_t)n;
                float v = acc[j][r] + bv;
                if (EPI == EPI_BIAS_RELU_DROP) {
                    v = fmaxf(v, 0.0f);
                    v = drop_apply(g.drop, didx, v);
                } else if (EPI == EPI_BIAS_DROP_RES_MASK) {
                    v = drop_apply(g.drop, didx, v);
                    v = (g.seq[mf] != 0) ? (v + g.aux[idx]) : 0.0f;
                } else if (EPI == EPI_RELUDROPGRAD) {
                    v = (g.aux[idx] != 0.0f) ? v * g.drop.scale : 0.0f;
                } else if (EPI == EPI_ADD) {
                    v = v + g.aux[idx];
                }
                g.C[idx] = v;
            }
        }
    }
}

// dW_aug[HP][HP] slab per workgroup: rows 0..H-1 = A^T.G, row H = colsum(G) (ones column appended to A).
__global__ __launch_bounds__(256) void k_gemm_atb(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ slab,
                                                  int M, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A_l = smem;                 // [TM][LDT]
    float* G_l = smem + TM * LDT;      // [TM][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mq = (wave >> 1) * 80, nq = (wave & 1) * 80;    // wave -> 80x80 quadrant of dW_aug (5x5 tiles)
    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_tiles = (M + TM - 1) / TM;
    const int r = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        __syncthreads();
        for (int i = tid; i < TM * LDT; i += 256) {
            const int rr = i / LDT, c = i - rr * LDT;
            const int m = m0 + rr;
            float a = 0.0f, gg = 0.0f;
            if (m < M) {
                if (c < H) { a = A[(size_t)m * H + c]; gg = G[(size_t)m * H + c]; }
                else if (c == H) a = 1.0f;
            }
            A_l[i] = a; G_l[i] = gg;
        }
        __syncthreads();
        const float* ap = A_l + q * LDT + mq + r;
        const float* bp = G_l + q * LDT + nq + r;
        for (int ks = 0; ks < TM / 4; ++ks) {
            float a[5], b[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) { a[i] = ap[i * 16]; b[i] = bp[i * 16]; }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            ap += 4 * LDT; bp += 4 * LDT;
        }
    }
    float* out = slab + (size_t)blockIdx.x * HP * HP;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                out[(size_t)(mq + i * 16 + q * 4 + rr) * HP + nq + j * 16 + r] = acc[i][j][rr];
}

// ============================================================================================= C ABI
static const size_t kGemmRowsLds = (size_t)(HP * LDW + TM * LDA) * sizeof(float);
static const size_t kGemmAtbLds = (size_t)(2 * TM * LDT) * sizeof(float);

template <int EPI, bool TB>
static int launch_rows(const GemmArgs& g, hipStream_t st) {
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_rows<EPI, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmRowsLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int n_tiles = (g.M + TM - 1) / TM;
    const int per = (n_tiles + 255) / 256;                 // tiles per workgroup, balanced over the 256 CUs
    const int grid = (n_tiles + per - 1) / per;
    hipLaunchKernelGGL((k_gemm_rows<EPI, TB>), dim3(grid), dim3(512), kGemmRowsLds, st, g);
    HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int ader_gemm_rows(const float* A, const float* W, const float* bias, float* C, const float* aux, const int* seq, int M, int H,
                   int epilogue, int trans_b, int row_mul, int row_add, const AderDrop* drop, void* stream) {
    if (M <= 0) return 0;
    if (H > HP || H < 1) return -2;
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.C = C; g.aux = aux; g.seq = seq; g.M = M; g.H = H;
    g.row_mul = row_mul; g.row_add = row_add;
    g.drop = drop_from(drop);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue * 2 + (trans_b ? 1 : 0)) {
        case EPI_BIAS * 2 + 0: return launch_rows<EPI_BIAS, false>(g, st);
        case EPI_BIAS * 2 + 1: return launch_rows<EPI_BIAS, true>(g, st);   // plain dX = dY . W^T (bias NULL)
        case EPI_BIAS_RELU_DROP * 2 + 0: return launch_rows<EPI_BIAS_RELU_DROP, false>(g, st);
        case EPI_BIAS_DROP_RES_MASK * 2 + 0: return launch_rows<EPI_BIAS_DROP_RES_MASK, false>(g, st);
        case EPI_RELUDROPGRAD * 2 + 1: return launch_rows<EPI_RELUDROPGRAD, true>(g, st);
        case EPI_ADD * 2 + 1: return launch_rows<EPI_ADD, true>(g, st);
        case EPI_ADD * 2 + 0: return launch_rows<EPI_ADD, false>(g, st);
        default: return -3;
    }
}

int ader_gemm_atb_slabs(int M) {
    const int n_tiles = (M + TM - 1) / TM;
    if (n_tiles < 1) return 1;
    const int per = (n_tiles + 255) / 256;                 // balanced over the 256 CUs
    return (n_tiles + per - 1) / per;
}

// slab: ader_gemm_atb_slabs(M) * 160 * 160 floats of scratch.  dW [H,H] and db [H] (may be null) are overwritten.
int ader_gemm_atb(const float* A, const float* G, float* slab, float* dW, float* db, int M, int H, void* stream) {
    if (M <= 0) return 0;
    if (H >= HP || H < 1) return -2;     // needs one spare row for the ones-column bias trick
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_atb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmAtbLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int S = ader_gemm_atb_slabs(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gemm_atb, dim3(S), dim3(256), kGemmAtbLds, st, A, G, slab, M, H);
    HIP_LAUNCH_CHECK();
    return ader_reduce_slabs(slab, (long)HP * HP, S, HP, H, H, dW, db, stream);
}

}  // extern "C"
