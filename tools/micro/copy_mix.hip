// Micro-benchmark: read + write mix at the size of one C = 24 full-rate layer (147 MB in, 147 MB out,
// 384 rows of 96000 floats): what the HBM system sustains when both directions stream at once.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4, int reads, int writes) {
    const long stride = (long)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 v = {1.f, 2.f, 3.f, 4.f};
        if (reads) v = x[i];
        if (writes) y[i] = v * 1.5f; else acc += v;
    }
    if (!writes && acc.x == 1.2345e30f) y[0] = acc;
}
int main() {
    const long n4 = 16L * 24 * 96000 / 4;
    f32x4 *x, *y, *x2, *y2;
    (void)hipMalloc(&x, n4 * 16); (void)hipMalloc(&y, n4 * 16); (void)hipMalloc(&x2, n4 * 16); (void)hipMalloc(&y2, n4 * 16);
    (void)hipMemset(x, 0, n4 * 16); (void)hipMemset(x2, 0, n4 * 16);
    const double mb = n4 * 16 / 1e6;
    for (int mode = 0; mode < 3; ++mode) {
        const int reads = mode != 1, writes = mode != 0;
        for (int grid : {2048, 8192, 32768}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, (i & 1) ? x2 : x, (i & 1) ? y2 : y, n4, reads, writes);
            (void)hipEventRecord(a);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, (i & 1) ? x2 : x, (i & 1) ? y2 : y, n4, reads, writes);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 10.f;
            const double bytes = mb * (reads + writes);
            std::printf("%s grid %5d: %6.1f us  %.2f TB/s\n", mode == 0 ? "read  147 MB      " : mode == 1 ? "write 147 MB      " : "copy  147 + 147 MB", grid, ms * 1e3, bytes / ms / 1e6 * 1e3);
        }
    }
    return 0;
}
