// Dev probe: shapes of the whole-row gather x[r] = E[ids[r]] * s + P[r % T] (600-B rows of a 600 MB table, 409,600 random rows).
// build: hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) V16 { f32x4 v; };
#define H 150
#define T 50

// V0: R rows per wave, lane = 16-byte piece of the row (38 pieces, lanes 38..63 idle), all loads before the first use
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_v0(const int* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pos,
                                            float* __restrict__ x, int rows, float s) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    if (row0 >= rows) return;
    const int c = 4 * lane;
    const bool full = c + 4 <= H, part = !full && c < H;
    int id[R];
#pragma unroll
    for (int u = 0; u < R; ++u) id[u] = ids[min(row0 + u, rows - 1)];
    f32x4 e[R], q[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int row = min(row0 + u, rows - 1);
        const float* ep = emb + (size_t)id[u] * H + c;
        const float* pp = pos + (size_t)(row % T) * H + c;
        e[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; q[u] = e[u];
        if (full) { e[u] = ((const V16*)ep)->v; q[u] = ((const V16*)pp)->v; }
        else if (part) { e[u][0] = ep[0]; e[u][1] = ep[1]; q[u][0] = pp[0]; q[u][1] = pp[1]; }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int row = row0 + u;
        if (row >= rows || c >= H) continue;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = e[u][j] * s + q[u][j];
        float* op = x + (size_t)row * H + c;
        if (full) { if (NT) __builtin_nontemporal_store(o, &((V16*)op)->v); else ((V16*)op)->v = o; }
        else { op[0] = o[0]; op[1] = o[1]; }
    }
}

// V2: 8 rows per wave as 8-byte pieces: 75 pieces per row, 600 pieces = 9.4 wave-instructions of 64 x 8 B, every lane busy;
// piece p = lane + 64 k -> row p / 75, piece p % 75
template <int R, bool NT = false>
__global__ __launch_bounds__(256) void k_v2(const int* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pos,
                                            float* __restrict__ x, int rows, float s) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    if (row0 >= rows) return;
    constexpr int NP = R * 75, NI = (NP + 63) / 64;
    // the R ids of the wave: lane u < R loads one, broadcast by readlane
    int myid = (lane < R) ? ids[min(row0 + lane, rows - 1)] : 0;
    f32x2 e[NI], q[NI];
    int prow[NI], pc[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = lane + 64 * k;
        const int u = min(p / 75, R - 1), c2 = p - (p / 75) * 75;
        prow[k] = u; pc[k] = 2 * c2;
        const int id = __shfl(myid, u, 64);
        const int row = min(row0 + u, rows - 1);
        e[k] = (f32x2){0.f, 0.f}; q[k] = e[k];
        if (p < NP) {
            e[k] = *(const f32x2*)(emb + (size_t)id * H + 2 * c2);
            q[k] = *(const f32x2*)(pos + (size_t)(row % T) * H + 2 * c2);
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = lane + 64 * k;
        const int row = row0 + prow[k];
        if (p < NP && row < rows) {
            f32x2 o;
            o[0] = e[k][0] * s + q[k][0]; o[1] = e[k][1] * s + q[k][1];
            if (NT) __builtin_nontemporal_store(o, (f32x2*)(x + (size_t)row * H + pc[k])); else *(f32x2*)(x + (size_t)row * H + pc[k]) = o;
        }
    }
}

int main() {
    const int N = 1000000, rows = 409600;
    float *emb, *pos, *x; int* ids;
    hipMalloc(&emb, (size_t)(N + 1) * H * 4); hipMalloc(&pos, T * H * 4); hipMalloc(&x, (size_t)rows * H * 4); hipMalloc(&ids, rows * 4);
    hipMemset(emb, 0, (size_t)(N + 1) * H * 4); hipMemset(pos, 0, T * H * 4);
    std::vector<int> h(rows);
    unsigned st = 12345;
    for (int i = 0; i < rows; ++i) { st = st * 1664525u + 1013904223u; h[i] = 1 + (int)((st >> 8) % N); }
    hipMemcpy(ids, h.data(), rows * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double bytes = (double)rows * (4 + 2 * H * 4);
#define RUN(name, ...)                                                                 \
    { for (int it = 0; it < 3; ++it) { __VA_ARGS__; }                                  \
      (void)hipEventRecord(a); for (int it = 0; it < 10; ++it) { __VA_ARGS__; } (void)hipEventRecord(b); (void)hipEventSynchronize(b); \
      float ms; (void)hipEventElapsedTime(&ms, a, b);                                  \
      printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9); }
    RUN("V0 4 rows/wave", k_v0<4, false><<<(rows + 15) / 16, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V0 4 rows/wave nontemporal st", k_v0<4, true><<<(rows + 15) / 16, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V0 8 rows/wave", k_v0<8, false><<<(rows + 31) / 32, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V0 8 rows/wave nontemporal st", k_v0<8, true><<<(rows + 31) / 32, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V0 2 rows/wave", k_v0<2, false><<<(rows + 7) / 8, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V2 8 rows/wave packed 8-B pieces", k_v2<8><<<(rows + 31) / 32, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V2 4 rows/wave packed 8-B pieces", k_v2<4><<<(rows + 15) / 16, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V2 8 rows/wave packed, NT st", k_v2<8, true><<<(rows + 31) / 32, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V2 4 rows/wave packed, NT st", k_v2<4, true><<<(rows + 15) / 16, 256>>>(ids, emb, pos, x, rows, 12.25f));
    RUN("V2 16 rows/wave packed 8-B pieces", k_v2<16><<<(rows + 63) / 64, 256>>>(ids, emb, pos, x, rows, 12.25f));
    return 0;
}
