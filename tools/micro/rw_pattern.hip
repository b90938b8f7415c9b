// rw_pattern.hip - what an HBM-bound conv layer can reach with a given STORE shape: stream a (Z, 32, T) tensor
// through registers (row-contiguous loads: 4 rows x 256 B per wave instruction, like the hx producers) and write
// a (Z, 24|32, T) tensor with (0) the MFMA D-fragment shape - 16 rows x (4 lanes x 16 B | 8 B) per instruction -
// or (1) whole rows - 64 lanes x 16 B contiguous.  ESZ 4: float32 elements, 2: bfloat16 (8-byte lane accesses).
// Sizes well beyond the 256 MB Infinity Cache.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ESZ> struct Vec;
template <> struct Vec<4> { typedef f32x4 type; };
template <> struct Vec<2> { typedef f32x2 type; };       // 4 two-byte elements

template <int PATTERN, int ESZ, int ROWS>
__global__ __launch_bounds__(256) void rw_kernel(const char* x, char* y, long T, int tpw) {
    typedef typename Vec<ESZ>::type V;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* xb = x + (long)blockIdx.z * ROWS * T * ESZ;
    char* yb = y + (long)blockIdx.z * ROWS * T * ESZ;
    for (int k = 0; k < tpw; ++k) {
        const long t0 = ((long)blockIdx.x * tpw + k) * 256;           // 256 columns per workgroup tile, 64 per wave
        if (t0 >= T) return;
        V v[8];
        // loads: 8 instructions, each 4 rows x 16 lanes x 4 columns
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 4 + (lane >> 4);
            const long t = t0 + w * 64 + (lane & 15) * 4;
            v[i] = row < ROWS ? *reinterpret_cast<const V*>(xb + ((long)row * T + t) * ESZ) : V(0);
        }
        if (PATTERN == 0) {            // D-fragment shape: lane -> row = 16 m + (lane & 15), columns 16 n + 4 (lane >> 4)
            #pragma unroll
            for (int m = 0; m < 2; ++m)
                #pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int row = m * 16 + (lane & 15);
                    const long t = t0 + w * 64 + n * 16 + (lane >> 4) * 4;
                    if (row < ROWS) *reinterpret_cast<V*>(yb + ((long)row * T + t) * ESZ) = v[m * 4 + n] + V(1.f);
                }
        } else {                       // row shape: 4 rows x 256 B (or 128 B) per instruction, as loaded
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = i * 4 + (lane >> 4);
                const long t = t0 + w * 64 + (lane & 15) * 4;
                if (row < ROWS) *reinterpret_cast<V*>(yb + ((long)row * T + t) * ESZ) = v[i] + V(1.f);
            }
        }
    }
}

template <int P, int ESZ, int ROWS> static float run(const char* x, char* y, long T, int Z, int tpw) {
    dim3 grid((unsigned)(((T + 255) / 256 + tpw - 1) / tpw), 1, Z);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((rw_kernel<P, ESZ, ROWS>), grid, dim3(256), 0, 0, x, y, T, tpw);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rw_kernel<P, ESZ, ROWS>), grid, dim3(256), 0, 0, x, y, T, tpw);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5.f;
}

int main() {
    const long T = 240000; const int Z = 128;            // cfg3's down.0 geometry: 2 x 64 rows of 24 channels
    char *x, *y;
    hipMalloc(&x, (size_t)Z * 32 * T * 4); hipMalloc(&y, (size_t)Z * 32 * T * 4);
    hipMemset(x, 0, (size_t)Z * 32 * T * 4);
    for (int tpw : {1, 4, 16}) {
        const double gb4 = 2.0 * Z * 24 * T * 4 / 1e9, gb2 = gb4 / 2;
        const float a = run<0, 4, 24>(x, y, T, Z, tpw), b = run<1, 4, 24>(x, y, T, Z, tpw);
        const float c = run<0, 2, 24>(x, y, T, Z, tpw), d = run<1, 2, 24>(x, y, T, Z, tpw);
        std::printf("tpw %2d  f32 %.2f GB: D-shape %.0f us (%.2f TB/s) | rows %.0f us (%.2f TB/s)   bf16 %.2f GB: D-shape %.0f us (%.2f TB/s) | rows %.0f us (%.2f TB/s)\n",
                    tpw, gb4, a * 1e3, gb4 / a, b * 1e3, gb4 / b, gb2, c * 1e3, gb2 / c, d * 1e3, gb2 / d);
    }
    return 0;
}
