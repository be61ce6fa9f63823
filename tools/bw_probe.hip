// HBM streaming probe for the table update's access pattern (not part of the product): how fast can gfx950 run
// "read p,m,v,(g) / write p,m,v" in place, as a function of the number of concurrent streams and the layout?
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o gpurun_out/bw_probe && gpurun_out/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U> __global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
    const size_t span = 256 * U;
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
        f4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; x[u] = i < n4 ? a[i] : f4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n4) b[i] = x[u]; }
    }
}

__device__ inline void upd(f4& p, f4& m, f4& v, f4 g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] += (g[k] - m[k]) * 0.1f;
        v[k] += (g[k] * g[k] - v[k]) * 0.001f;
        p[k] -= m[k] * 1e-3f / (__builtin_sqrtf(v[k]) + 1e-8f);
    }
}

// SoA in place: streams = 3 read + 3 write (+1 read with G)
template <int U, bool G> __global__ __launch_bounds__(256) void k_soa(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v,
                                                                     const f4* __restrict__ g, size_t n4) {
    const size_t span = 256 * U;
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
        f4 P[U], M[U], V[U], Gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + u * 256 + threadIdx.x;
            if (i < n4) { P[u] = p[i]; M[u] = m[i]; V[u] = v[i]; Gg[u] = G ? __builtin_nontemporal_load(&g[i]) : f4{1, 2, 3, 4}; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + u * 256 + threadIdx.x;
            if (i < n4) { upd(P[u], M[u], V[u], Gg[u]); p[i] = P[u]; m[i] = M[u]; v[i] = V[u]; }
        }
    }
}

// AoS by chunk: one array, chunk c of CH float4 holds [p(CH) | m(CH) | v(CH)] contiguously: 1 read + 1 write stream
template <int U, bool G> __global__ __launch_bounds__(256) void k_aos(f4* __restrict__ pmv, const f4* __restrict__ g, size_t n4) {
    const size_t span = 256 * U;     // one chunk = span float4 of p, then of m, then of v
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
        f4 P[U], M[U], V[U], Gg[U];
        f4* q = pmv + base * 3;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t o = u * 256 + threadIdx.x;
            P[u] = q[o]; M[u] = q[span + o]; V[u] = q[2 * span + o];
            Gg[u] = G ? __builtin_nontemporal_load(&g[base + o]) : f4{1, 2, 3, 4};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t o = u * 256 + threadIdx.x;
            upd(P[u], M[u], V[u], Gg[u]); q[o] = P[u]; q[span + o] = M[u]; q[2 * span + o] = V[u];
        }
    }
}

// read-only and write-only streams
template <int U> __global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, float* out, size_t n4) {
    const size_t span = 256 * U;
    f4 acc = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n4) acc += a[i]; }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1;
}
template <int U> __global__ __launch_bounds__(256) void k_write(f4* __restrict__ a, size_t n4) {
    const size_t span = 256 * U;
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n4) a[i] = f4{1, 2, 3, 4}; }
    }
}

template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t n = (size_t)1000064 * 160, n4 = n / 4;      // the cfg-S table
    f4 *p, *m, *v, *g, *pmv; float* out;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&g, n * 4));
    CK(hipMalloc(&pmv, n * 12 + (1 << 20))); CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); CK(hipMemset(g, 0, n * 4));
    CK(hipMemset(pmv, 0, n * 12 + (1 << 20)));
    const double MB = n * 4 / 1e6;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float t;
        t = timeit([&] { k_copy<4><<<grid, 256>>>(p, m, n4); });
        printf("grid %5d copy U4        %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 2 * MB / t / 1e6);
        t = timeit([&] { k_read<4><<<grid, 256>>>(p, out, n4); });
        printf("grid %5d read U4        %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 1 * MB / t / 1e6);
        t = timeit([&] { k_write<4><<<grid, 256>>>(p, n4); });
        printf("grid %5d write U4       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 1 * MB / t / 1e6);
        t = timeit([&] { k_soa<4, true><<<grid, 256>>>(p, m, v, g, n4); });
        printf("grid %5d soa+g U4       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 7 * MB / t / 1e6);
        t = timeit([&] { k_soa<2, true><<<grid, 256>>>(p, m, v, g, n4); });
        printf("grid %5d soa+g U2       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 7 * MB / t / 1e6);
        t = timeit([&] { k_soa<4, false><<<grid, 256>>>(p, m, v, g, n4); });
        printf("grid %5d soa   U4       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 6 * MB / t / 1e6);
        t = timeit([&] { k_aos<4, true><<<grid, 256>>>(pmv, g, n4 / 1024 * 1024); });
        printf("grid %5d aos+g U4       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 7 * MB / t / 1e6);
        t = timeit([&] { k_aos<4, false><<<grid, 256>>>(pmv, g, n4 / 1024 * 1024); });
        printf("grid %5d aos   U4       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 6 * MB / t / 1e6);
        t = timeit([&] { k_aos<2, false><<<grid, 256>>>(pmv, g, n4 / 512 * 512); });
        printf("grid %5d aos   U2       %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 6 * MB / t / 1e6);
    }
    // one-shot grids (no grid-stride loop): every workgroup one span
    {
        int grid = (int)((n4 + 1023) / 1024);
        float t = timeit([&] { k_soa<4, true><<<grid, 256>>>(p, m, v, g, n4); });
        printf("oneshot %d soa+g U4  %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 7 * MB / t / 1e6);
        t = timeit([&] { k_aos<4, false><<<grid, 256>>>(pmv, g, n4 / 1024 * 1024); });
        printf("oneshot %d aos U4    %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 6 * MB / t / 1e6);
        t = timeit([&] { k_copy<4><<<grid, 256>>>(p, m, n4); });
        printf("oneshot %d copy U4   %7.1f us  %5.2f TB/s\n", grid, t * 1e3, 2 * MB / t / 1e6);
    }
    return 0;
}
