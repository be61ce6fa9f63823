// Host cost of enqueueing kernels with hipLaunchKernelGGL (no synchronisation inside the timed loop): what bounds the host side of a
// train step once Python is out of the loop (ader_step_enqueue walks ~30 launches + 8 event operations per step).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_probe tools/launch_probe.hip && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
struct Big { float x[24]; int n[8]; void* p[16]; };
__global__ void k_small(float* p, int n) { if (threadIdx.x == 0 && n < 0) p[0] = 1.0f; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && b.n[0] < 0) ((float*)b.p[0])[0] = b.x[0]; }
int main() {
    float* d; hipMalloc(&d, 4);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    hipEvent_t ev[8]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    Big b = {}; b.p[0] = d;
    for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        const int steps = 200;
        auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) {
            for (int i = 0; i < 22; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s0, d, i);
            for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, (i & 1) ? s1 : s0, b);
            for (int i = 0; i < 4; ++i) { hipEventRecord(ev[i], (i & 1) ? s0 : s1); hipStreamWaitEvent((i & 1) ? s1 : s0, ev[i], 0); }
        }
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("30 launches + 4 record/wait pairs per step: host enqueue %.1f us/step (%.2f us per call), with the GPU drained %.1f us/step\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / steps,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / steps / 38,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / steps);
    }
    return 0;
}
