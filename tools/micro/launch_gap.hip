// Cost of a kernel boundary on one stream (diagnostic): M dependent launches of a kernel whose every workgroup
// spins for `cycles` shader clocks, against the same total spin in ONE launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void spin(long cycles, float* sink, int touch) {
    const long t0 = clock64();
    float acc = 0.f;
    if (touch) acc = sink[(blockIdx.x * 512 + threadIdx.x) & 0xffff];       // a dependent global read, as a real layer has
    while (clock64() - t0 < cycles) acc += 1e-9f;
    if (acc == 123.f) sink[0] = acc;
    if (touch) sink[(blockIdx.x * 512 + threadIdx.x) & 0xffff] = acc;       // and a write the next launch reads
}

int main() {
    float* sink;
    hipMalloc(&sink, 1 << 20);
    hipMemset(sink, 0, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st;
    hipStreamCreate(&st);
    const int M = 40;
    for (int touch = 0; touch < 2; ++touch)
        for (int wgs : {256, 512, 2048})
            for (long cyc : {2000L, 20000L, 60000L}) {
                auto run = [&](int launches, long c) {
                    for (int w = 0; w < 2; ++w) {
                        if (w == 1) hipEventRecord(e0, st);
                        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), 0, st, c, sink, touch);
                    }
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms = 0.f;
                    hipEventElapsedTime(&ms, e0, e1);
                    return ms * 1e3f;
                };
                const float many = run(M, cyc), one = run(1, cyc * M);
                std::printf("touch %d  wgs %4d  spin %6ld cyc: %d launches %8.1f us, one launch of the same spin %8.1f us -> %.2f us per boundary\n",
                            touch, wgs, cyc, M, many, one, (many - one) / (M - 1));
            }
    return 0;
}
