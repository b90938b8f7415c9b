// Micro-benchmark: store throughput when only a few waves per CU issue the stores, in bursts separated by
// compute (the shape of the pipelined conv kernel's epilogue: 4 consumer waves per workgroup, 2 workgroups
// per CU, 8 x 1 KB stores per wave per tile, then an MFMA loop).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 512-thread workgroups; waves [0, storing) store the 24 x 256 tile in the MFMA C layout, everyone
// then idles `sleep64` x 64 cycles (stand-in for the MFMA loop) and meets at a barrier.
__global__ __launch_bounds__(512) void k(float* y, int T, int tiles_per_wg, int storing, int sleep64, float v,
                                          unsigned long long* cyc) {
    extern __shared__ unsigned char smem[];      // sized by the host to set the workgroups per CU
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* yb = y + (long)blockIdx.z * 24 * T;
    unsigned long long spent = 0;
    for (int k = 0; k < tiles_per_wg; ++k) {
        const int t0 = (blockIdx.x * tiles_per_wg + k) * 256;
        if (t0 >= T) break;
        if (w < storing) {
            const int per = 16 / storing;                      // 16 (m, n) items of 16 x 16 shared by the storing waves
            const unsigned long long a = __builtin_readcyclecounter();
            for (int i = 0; i < per; ++i) {
                const int item = w * per + i, m = item & 1, n = item >> 1;        // n: 0..7 -> 8 x 32 columns
                const int row = m * 16 + (lane & 15), t = t0 + n * 32 + (lane >> 4) * 4;
                const f32x4 val = {v, v + 1.f, v + 2.f, v + (float)k};
                if (row < 24) {
                    *reinterpret_cast<f32x4*>(yb + (long)row * T + t) = val;
                    *reinterpret_cast<f32x4*>(yb + (long)row * T + t + 16) = val;
                }
            }
            spent += __builtin_readcyclecounter() - a;
        }
        for (int s = 0; s < sleep64; ++s) __builtin_amdgcn_s_sleep(1);   // 64 cycles each
        __syncthreads();
    }
    if (lane == 0 && w == 0 && cyc) atomicAdd(cyc, spent);
}

int main() {
    const int T = 96000, Z = 16, tiles = 375;
    float* y; (void)hipMalloc(&y, (size_t)Z * 24 * T * 4 * 2);
    unsigned long long* cyc; (void)hipMalloc(&cyc, 8);
    const double mb = (double)Z * 24 * T * 4 / 1e6;
    std::printf("%.0f MB per launch; 512-thread workgroups, stores in the 16 x 64 B MFMA layout\n", mb);
    for (int wgs_per_cu : {1, 2, 4})
        for (int storing : {4, 8})
            for (int sleep64 : {0, 40, 80}) {
                const int tpw = 12;
                const size_t smem = wgs_per_cu == 1 ? 100 * 1024 : wgs_per_cu == 2 ? 70 * 1024 : 36 * 1024;
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                dim3 grid((tiles + tpw - 1) / tpw, 1, Z);
                hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, grid, dim3(512), smem, 0, y + (i & 1) * (size_t)Z * 24 * T, T, tpw, storing, sleep64, 1.f, nullptr);
                (void)hipMemset(cyc, 0, 8);
                (void)hipEventRecord(a);
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, grid, dim3(512), smem, 0, y + (i & 1) * (size_t)Z * 24 * T, T, tpw, storing, sleep64, (float)i, cyc);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 10.f;
                unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double bursts = 10.0 * grid.x * Z * tpw;
                std::printf("wg/CU %d storing waves %d sleep %4d cyc: %6.1f us  %.2f TB/s  store burst %.0f cyc per tile (wave 0)\n",
                            wgs_per_cu, storing, sleep64 * 64, ms * 1e3, mb / ms / 1e6 * 1e3, (double)c / bursts);
            }
    return 0;
}
