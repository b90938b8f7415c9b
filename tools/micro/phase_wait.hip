// Cost of an in-kernel phase boundary against a kernel boundary (diagnostic for the persistent per-block launches):
// P dependent phases of G workgroup-jobs each, every job writing a line the next phase reads,
//   (a) as P launches of G workgroups,
//   (b) as ONE launch whose workgroups pull jobs from an ORDERED ticket counter (phase 0's jobs first) and, before the
//       first job of phase k, wait until phase k-1's completion counter is full - device-scope release (L2 write-back)
//       + atomic on completion, acquire (L2 invalidate) after the wait.  Deadlock-free for ANY number of resident
//       workgroups: a waiting workgroup only waits for jobs that running workgroups hold.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/phase_wait.hip -o /tmp/phase_wait && /tmp/phase_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void job(long cycles, float* buf, int phase, int j, int G) {
    // read what the previous phase's job (j + 1) % G wrote (another workgroup, most likely another XCD), spin, write
    const int tid = threadIdx.x;
    float v = phase > 0 ? buf[((phase - 1) * G + (j + 1) % G) * 512 + tid] : 1.f;
    const long t0 = clock64();
    while (clock64() - t0 < cycles) v += 1e-9f;
    buf[(phase * G + j) * 512 + tid] = v + 1.f;
}

__global__ __launch_bounds__(512) void phase_kernel(long cycles, float* buf, int phase, int G) {
    job(cycles, buf, phase, blockIdx.x, G);
}

__global__ __launch_bounds__(512) void persistent_kernel(long cycles, float* buf, int P, int G, unsigned* ctr /* [0] tickets, [1 + p] done */,
                                                         unsigned* err) {
    __shared__ unsigned s_t;
    int seen_phase = 0;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&ctr[0], 1u);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= (unsigned)(P * G)) return;
        const int phase = t / G, j = t - phase * G;
        if (phase > seen_phase) {
            if (threadIdx.x == 0) {
                long spins = 0;
                while (__hip_atomic_load(&ctr[phase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1L << 24)) { *err = 1u; break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            seen_phase = phase;
        }
        job(cycles, buf, phase, j, G);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&ctr[1 + phase], 1u);
    }
}

int main() {
    const int P = 5;
    float* buf;
    unsigned *ctr, *err;
    hipMalloc(&buf, (size_t)P * 4096 * 512 * sizeof(float));
    hipMalloc(&ctr, 64 * sizeof(unsigned));
    hipMalloc(&err, sizeof(unsigned));
    hipMemset(err, 0, sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st;
    hipStreamCreate(&st);
    const int R = 20;
    for (int G : {64, 224, 256, 1024})
        for (long cyc : {2000L, 10000L, 40000L}) {
            float t_launches = 0.f, t_persist = 0.f;
            for (int w = 0; w < 2; ++w) {
                if (w == 1) hipEventRecord(e0, st);
                for (int r = 0; r < R; ++r)
                    for (int p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(512), 0, st, cyc, buf, p, G);
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&t_launches, e0, e1);
            for (int grid : {256, 512}) {
                for (int w = 0; w < 2; ++w) {
                    if (w == 1) hipEventRecord(e0, st);
                    for (int r = 0; r < R; ++r) {
                        hipMemsetAsync(ctr, 0, 64 * sizeof(unsigned), st);
                        hipLaunchKernelGGL(persistent_kernel, dim3(grid < G ? grid : G), dim3(512), 0, st, cyc, buf, P, G, ctr, err);
                    }
                }
                hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&t_persist, e0, e1);
                unsigned herr = 0;
                hipMemcpy(&herr, err, sizeof(herr), hipMemcpyDeviceToHost);
                std::printf("G %4d jobs/phase, job %6ld cyc (%.1f us): %d launches %7.1f us | one persistent launch (grid %3d, + its memset) %7.1f us%s\n",
                            G, cyc, cyc / 2400.0, P, t_launches * 1e3f / R, grid < G ? grid : G, t_persist * 1e3f / R, herr ? "  [WAIT TIMED OUT]" : "");
            }
        }
    return 0;
}
