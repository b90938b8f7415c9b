// Micro-benchmark: HBM store throughput of the MFMA-epilogue store shape (16 rows x 64 B per wave
// instruction) against row-contiguous shapes, same grid and bytes.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows = Z * 24 channel rows of T floats; a workgroup of 4 waves owns a 256-column tile of one z.
template <int PATTERN>
__global__ __launch_bounds__(256) void store_kernel(float* y, int T, int tpw, float v) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* yb = y + (long)blockIdx.z * 24 * T;
    for (int k = 0; k < tpw; ++k) {
        const int t0 = (blockIdx.x * tpw + k) * 256;
        if (t0 >= T) return;
        const f32x4 val = {v, v + 1.f, v + 2.f, v + (float)k};
        if (PATTERN == 0) {            // MFMA C layout: 16 rows x 64 B per instruction
            for (int m = 0; m < 2; ++m)
                for (int n = 0; n < 4; ++n) {
                    const int row = m * 16 + (lane & 15), t = t0 + w * 64 + n * 16 + (lane >> 4) * 4;
                    if (row < 24) *reinterpret_cast<f32x4*>(yb + (long)row * T + t) = val;
                }
        } else if (PATTERN == 1) {     // 8 rows x 128 B
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane;               // 512 float4 = 32 rows x 16 float4 (64 t)
                const int row = (idx >> 3) & 31, q = (idx & 7) + 8 * (idx >> 8);
                const int t = t0 + w * 64 + q * 4;
                if (row < 24) *reinterpret_cast<f32x4*>(yb + (long)row * T + t) = val;
            }
        } else if (PATTERN == 2) {     // 4 rows x 256 B (wave-private transpose)
            for (int i = 0; i < 8; ++i) {
                const int row = i * 4 + (lane >> 4), t = t0 + w * 64 + (lane & 15) * 4;
                if (row < 24) *reinterpret_cast<f32x4*>(yb + (long)row * T + t) = val;
            }
        } else {                       // 1 row x 1 KB (workgroup transpose)
            for (int i = 0; i < 6; ++i) {
                const int row = w * 6 + i, t = t0 + lane * 4;
                *reinterpret_cast<f32x4*>(yb + (long)row * T + t) = val;
            }
        }
    }
}

template <int P> static float run(float* y, int T, int Z, int tpw, int wgs_mult) {
    dim3 grid(((T + 255) / 256 + tpw - 1) / tpw, 1, Z);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<P>, grid, dim3(256), 0, 0, y, T, tpw, 1.f);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<P>, grid, dim3(256), 0, 0, y, T, tpw, (float)i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10.f;
}

int main() {
    const int T = 96000, Z = 16;
    float* y; hipMalloc(&y, (size_t)Z * 24 * T * 4);
    const double mb = (double)Z * 24 * T * 4 / 1e6;
    for (int tpw : {1, 4, 12}) {
        const float t0 = run<0>(y, T, Z, tpw, 1), t1 = run<1>(y, T, Z, tpw, 1), t2 = run<2>(y, T, Z, tpw, 1), t3 = run<3>(y, T, Z, tpw, 1);
        std::printf("tpw %2d  %.0f MB: 16x64B %.1f us (%.2f TB/s) | 8x128B %.1f us (%.2f) | 4x256B %.1f us (%.2f) | 1x1KB %.1f us (%.2f)\n",
                    tpw, mb, t0 * 1e3, mb / t0 / 1e3 / 1e3 * 1e3 / 1e3 * 1e3, t1 * 1e3, mb / t1 / 1e6 * 1e3, t2 * 1e3, mb / t2 / 1e6 * 1e3, t3 * 1e3, mb / t3 / 1e6 * 1e3);
    }
    return 0;
}
