// cu_pull.hip - how many bytes per cycle ONE CU can pull through its vector-memory pipe, by request shape, requests in flight,
// waves issuing and where the data lives (round 6: what paces the wide layers is the CU's request queue, not HBM or the matrix
// pipe - DESIGN.md "what a request costs").  One 256-thread (or 64 * W) workgroup per CU streams a (Z, C, T) bfloat16 tensor
// the way the conv kernels' staging does: per step a wave requests 8 instructions = one 32-channel chunk of its 64-column
// slice, keeps NSETS such steps in flight in registers and folds what arrives into a checksum.
//   SHAPE 0: 8 B per lane, 4 rows x 128 B per instruction   (conv_hx / conv_wx window requests, bfloat16)
//   SHAPE 1: 16 B per lane, 4 rows x 256 B per instruction  (8 time steps per lane)
//   SHAPE 2: 16 B per lane, 8 rows x 128 B per instruction
//   SHAPE 3: 16 B per lane, 1 KB contiguous per instruction (weight fragments)
//   SHAPE 4: 8 B per lane, 16 rows x 32 B per instruction   (epilogue operands in the MFMA result layout)
// hipcc --offload-arch=gfx950 -O3 tools/micro/cu_pull.hip -o /tmp/cu_pull && /tmp/cu_pull
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE> struct Req { typedef u32x4 type; static constexpr int bytes = 16; };
template <> struct Req<0> { typedef u32x2 type; static constexpr int bytes = 8; };
template <> struct Req<4> { typedef u32x2 type; static constexpr int bytes = 8; };

__device__ __forceinline__ unsigned fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ unsigned fold(u32x2 v) { return v.x ^ v.y; }

// byte offset of instruction i (0..7) of a step for this lane; row pitch `pitch` bytes, the step's first row r0, first byte c0
template <int SHAPE>
__device__ __forceinline__ long req_off(int i, int lane, long pitch, long r0, long c0) {
    if (SHAPE == 0) return (r0 + 4 * i + (lane & 3)) * pitch + c0 + (lane >> 2) * 8;            // 4 rows x 16 lanes x 8 B
    if (SHAPE == 1) return (r0 + 4 * i + (lane >> 4)) * pitch + c0 + (lane & 15) * 16;         // 4 rows x 16 lanes x 16 B
    if (SHAPE == 2) return (r0 + 8 * (i & 3) + (lane >> 3)) * pitch + c0 + (i >> 2) * 128 + (lane & 7) * 16;   // 8 rows x 128 B
    if (SHAPE == 3) return (r0 * pitch) + c0 + (long)i * 1024 + lane * 16;                      // 1 KB contiguous
    return (r0 + 16 * (i & 1) + (lane & 15)) * pitch + c0 + (i >> 1) * 32 + (lane >> 4) * 8;    // 16 rows x 32 B
}
// bytes one step (8 instructions) of one wave covers per row, rows per step
template <int SHAPE> constexpr int step_cols() { return SHAPE == 0 ? 128 : SHAPE == 1 ? 256 : SHAPE == 2 ? 256 : SHAPE == 3 ? 8192 : 128; }
template <int SHAPE> constexpr int step_rows() { return SHAPE == 3 ? 1 : 32; }

template <int SHAPE, int NSETS>
__global__ __launch_bounds__(512) void pull_kernel(const char* x, unsigned* out, long pitch, int C, long T_bytes, int Zd, int steps) {
    typedef typename Req<SHAPE>::type V;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* xb = x + (long)(blockIdx.x % Zd) * C * pitch;
    const int nchunk = C / step_rows<SHAPE>();
    const long cols_per_tile = (long)step_cols<SHAPE>() * nw;
    const long ntile = SHAPE == 3 ? 1 : T_bytes / cols_per_tile;
    V v[NSETS][8];
    unsigned acc = 0;
    long s_issue = 0;
    auto issue = [&](int set) {
        long r0, c0;
        if (SHAPE == 3) {                                  // a flat walk over the utterance's bytes
            r0 = 0; c0 = ((s_issue * nw + w) * 8192 + (long)(blockIdx.x / Zd) * 65536) % ((long)C * pitch - 8192);
        } else {
            const long tile = (s_issue / nchunk + blockIdx.x / Zd) % ntile, ch = s_issue % nchunk;
            r0 = ch * step_rows<SHAPE>(); c0 = tile * cols_per_tile + (long)w * step_cols<SHAPE>();
        }
        ++s_issue;
        #pragma unroll
        for (int i = 0; i < 8; ++i)
            v[set][i] = *reinterpret_cast<const V*>(xb + req_off<SHAPE>(i, lane, pitch, r0, c0));
    };
    #pragma unroll
    for (int s = 0; s < NSETS; ++s) issue(s);
    for (int k = 0; k < steps; k += NSETS) {
        #pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= fold(v[s][i]);
            issue(s);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int SHAPE, int NSETS>
static void run(const char* name, const char* x, unsigned* out, long pitch, int C, long T_bytes, int Zd, int waves, const char* where) {
    const int steps = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((pull_kernel<SHAPE, NSETS>), dim3(256), dim3(64 * waves), 0, 0, x, out, pitch, C, T_bytes, Zd, 64);
    hipEventRecord(a);
    hipLaunchKernelGGL((pull_kernel<SHAPE, NSETS>), dim3(256), dim3(64 * waves), 0, 0, x, out, pitch, C, T_bytes, Zd, steps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * waves * (steps + NSETS) * 8.0 * 64.0 * Req<SHAPE>::bytes;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s %-4s waves %d  in flight/wave %2d req = %5.1f KB/CU  %6.2f TB/s  = %5.1f B/clk/CU @2.1GHz\n", name, where, waves, NSETS * 8,
           waves * NSETS * 8 * 64.0 * Req<SHAPE>::bytes / 1024.0, tbs, tbs * 1e12 / 256.0 / 2.1e9);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    // (Z, C, T) bfloat16: rows of 24000 B (T = 12000, the 8F-rate tensors at 10 s), 192 channels; Z = 128 -> 590 MB (HBM);
    // Zd = distinct utterances the 256 workgroups walk: 128 (HBM) or 1 (4.6 MB: L2 / Infinity Cache)
    const long pitch = 24000, C = 192, Z = 128;
    char* x; unsigned* out;
    hipMalloc(&x, Z * C * pitch + (1 << 20)); hipMalloc(&out, 4096);
    hipMemset(x, 1, Z * C * pitch + (1 << 20));
    for (int where = 0; where < 2; ++where) {
        const int Zd = where ? 1 : (int)Z;
        const char* wn = where ? "L2" : "HBM";
        for (int waves : {1, 2, 4, 8}) {
            run<0, 2>("8 B/lane, 4 rows x 128 B", x, out, pitch, C, pitch, Zd, waves, wn);
            run<0, 4>("8 B/lane, 4 rows x 128 B", x, out, pitch, C, pitch, Zd, waves, wn);
            run<1, 2>("16 B/lane, 4 rows x 256 B", x, out, pitch, C, pitch, Zd, waves, wn);
            run<2, 2>("16 B/lane, 8 rows x 128 B", x, out, pitch, C, pitch, Zd, waves, wn);
            run<3, 2>("16 B/lane, 1 KB contiguous", x, out, pitch, C, pitch, Zd, waves, wn);
            run<4, 2>("8 B/lane, 16 rows x 32 B", x, out, pitch, C, pitch, Zd, waves, wn);
        }
    }
    return 0;
}
