// mfma_probe.hip - facts about v_mfma_f32_16x16x32_{f16,bf16} on gfx950 that the split-half conv
// kernels rely on (run once on the GPU box; prints a small report):
//   1. operand / result lane layout (A: row = lane & 15, k = 8 * (lane >> 4) + e;  B: col = lane & 15,
//      same k;  D: col = lane & 15, rows 4 * (lane >> 4) + r) against a host GEMM on random data;
//   2. whether f16 DENORMAL inputs are preserved by the matrix pipe (decides whether the low halves
//      of the split operands need a power-of-two scale and a second accumulator set);
//   3. issue rate of back-to-back MFMAs (one wave per SIMD and two), f16 vs the f32-input MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_probe.hip -o tools/micro/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void layout_kernel(const float* A, const float* B, float* D) {   // A[16][32], B[32][16], D[16][16]
    const int lane = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (KIND == 0) {
        h8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)A[(lane & 15) * 32 + 8 * (lane >> 4) + e];
            b[e] = (_Float16)B[(8 * (lane >> 4) + e) * 16 + (lane & 15)];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    } else {
        b8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)A[(lane & 15) * 32 + 8 * (lane >> 4) + e];
            b[e] = (__bf16)B[(8 * (lane >> 4) + e) * 16 + (lane & 15)];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

// denormal probe: A[0][0] = a0 (f16 bits given), B[0][0] = b0, everything else 0 -> D[0][0] = a0 * b0
__global__ void denorm_kernel(unsigned short abits, unsigned short bbits, float* out) {
    const int lane = threadIdx.x;
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == 0) {
        a[0] = __builtin_bit_cast(_Float16, abits);
        b[0] = __builtin_bit_cast(_Float16, bbits);
    }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    // conversion behaviour of the producer's split: hi = f16(x), lo = f16(x - hi)
    if (lane == 1) {
        const float x = 1.0e-5f;                       // below the f16 normal range (6.1e-5)
        const _Float16 h = (_Float16)x;
        out[1] = (float)h;
        const float y = 0.3f;
        const _Float16 hy = (_Float16)y;
        const _Float16 ly = (_Float16)(y - (float)hy); // ~ 2^-13: subnormal in f16
        out[2] = (float)hy + (float)ly;
        out[3] = (float)ly;
    }
}

template <int KIND, int NACC>
__global__ void rate_kernel(float* out, int iters) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float seed = (float)threadIdx.x * 1e-3f;
    h8 ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(seed + e); bh[e] = (_Float16)(1.f - seed * e); }
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed + 1.f, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e33f) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), Dref(256);
    srand(7);
    for (auto& v : A) v = (float)((rand() % 17) - 8) / 8.f;
    for (auto& v : B) v = (float)((rand() % 17) - 8) / 4.f;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double s = 0;
            for (int k = 0; k < 32; ++k) s += (double)A[m * 32 + k] * B[k * 16 + n];
            Dref[m * 16 + n] = (float)s;
        }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 2; ++kind) {
        if (kind == 0) hipLaunchKernelGGL(layout_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        else hipLaunchKernelGGL(layout_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int i = 0; i < 256; ++i) worst = std::fmax(worst, std::fabs(D[i] - Dref[i]));
        std::printf("layout %s: max |D - ref| = %g  (%s)\n", kind ? "bf16" : "f16", worst, worst < 1e-5 ? "OK" : "MISMATCH");
    }
    // f16 denormal 2^-20 (bits 0x0010) times 2^10 (bits 0x6400) = 2^-10 if preserved
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, (unsigned short)0x0010, (unsigned short)0x6400, dD);
    CK(hipMemcpy(D.data(), dD, 16, hipMemcpyDeviceToHost));
    std::printf("denormal input 2^-20 * 2^10 -> %g (2^-10 = %g): %s\n", D[0], std::ldexp(1.0, -10),
                D[0] == (float)std::ldexp(1.0, -10) ? "PRESERVED" : "FLUSHED");
    std::printf("cvt f16(1e-5) -> %g ; 0.3 = hi + lo -> %.9g (lo = %g)\n", D[1], D[2], D[3]);
    // rate: 256 CUs x 4 SIMDs, 1 or 2 waves per SIMD
    for (int kind = 0; kind < 2; ++kind)
        for (int wps = 1; wps <= 2; ++wps) {
            const int iters = 20000, nacc = 8;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL((rate_kernel<0, 8>), dim3(256), dim3(256 * wps), 0, 0, dD, iters);
                else hipLaunchKernelGGL((rate_kernel<1, 8>), dim3(256), dim3(256 * wps), 0, 0, dD, iters);
            };
            launch();
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop_per = kind == 0 ? 16.0 * 16 * 32 * 2 : 16.0 * 16 * 4 * 2;
            const double total = (double)iters * nacc * flop_per * 256 * 4 * wps;
            std::printf("rate %s, %d wave(s)/SIMD: %.1f TFLOP/s\n", kind ? "f32 16x16x4" : "f16 16x16x32", wps, total / (ms * 1e-3) / 1e12);
        }
    return 0;
}
