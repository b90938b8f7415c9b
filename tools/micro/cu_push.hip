// cu_push.hip - the store side of cu_pull.hip: bytes per cycle ONE CU can push through its vector-memory pipe, by store shape, waves
// issuing and footprint (round 6: the layers that write more than they read - decimating pairs, stretched convs - sit at 2.5-3 TB/s
// of stores; timelines show the staging waves' next window requests queued behind a tile's epilogue stores).  One workgroup per CU
// streams over a (Z, C, T) bfloat16 tensor the way the epilogues write it; optionally one load request per K stores rides along
// (MIX > 0) to see what a read stream gets beside the stores.
//   SHAPE 0: 8 B per lane, 16 rows x 32 B per instruction   (MFMA result layout, bfloat16)
//   SHAPE 1: 16 B per lane, 4 rows x 256 B                  (pair / row-run epilogues)
//   SHAPE 2: 16 B per lane, 8 rows x 128 B
//   SHAPE 3: 16 B per lane, 1 KB contiguous
// hipcc --offload-arch=gfx950 -O3 tools/micro/cu_push.hip -o /tmp/cu_push && /tmp/cu_push
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE>
__device__ __forceinline__ long st_off(int i, int lane, long pitch, long r0, long c0) {
    if (SHAPE == 0) return (r0 + 16 * (i & 1) + (lane & 15)) * pitch + c0 + (i >> 1) * 32 + (lane >> 4) * 8;    // 16 rows x 32 B
    if (SHAPE == 1) return (r0 + 4 * i + (lane >> 4)) * pitch + c0 + (lane & 15) * 16;                            // 4 rows x 256 B
    if (SHAPE == 2) return (r0 + 8 * (i & 3) + (lane >> 3)) * pitch + c0 + (i >> 2) * 128 + (lane & 7) * 16;      // 8 rows x 128 B
    return r0 * pitch + c0 + (long)i * 1024 + lane * 16;                                                          // 1 KB contiguous
}
template <int SHAPE> constexpr int step_cols() { return SHAPE == 0 ? 128 : SHAPE == 1 ? 256 : SHAPE == 2 ? 256 : 8192; }
template <int SHAPE> constexpr int lane_bytes() { return SHAPE == 0 ? 8 : 16; }

template <int SHAPE, int MIX>
__global__ __launch_bounds__(512) void push_kernel(char* y, const char* x, unsigned* out, long pitch, int C, long T_bytes, int Zd, int steps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* yb = y + (long)(blockIdx.x % Zd) * C * pitch;
    const char* xb = x + (long)(blockIdx.x % Zd) * C * pitch;
    const int nchunk = SHAPE == 3 ? 1 : C / 32;
    const long cols_per_tile = (long)step_cols<SHAPE>() * nw;
    const long ntile = SHAPE == 3 ? 1 : T_bytes / cols_per_tile;
    unsigned acc = 0;
    u32x4 v = {(unsigned)lane, (unsigned)w, 3u, 4u};
    for (int s = 0; s < steps; ++s) {
        long r0, c0;
        if (SHAPE == 3) { r0 = 0; c0 = (((long)s * nw + w) * 8192 + (long)(blockIdx.x / Zd) * 65536) % ((long)C * pitch - 8192); }
        else { const long tile = (s / nchunk + blockIdx.x / Zd) % ntile, ch = s % nchunk; r0 = ch * 32; c0 = tile * cols_per_tile + (long)w * step_cols<SHAPE>(); }
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long o = st_off<SHAPE>(i, lane, pitch, r0, c0);
            if (lane_bytes<SHAPE>() == 8) *reinterpret_cast<u32x2*>(yb + o) = u32x2{v.x + i, v.y};
            else *reinterpret_cast<u32x4*>(yb + o) = u32x4{v.x + i, v.y, v.z, v.w};
            if (MIX > 0 && (i % MIX) == MIX - 1) {                     // one 16-byte-per-lane read request per MIX stores
                const u32x4 l = *reinterpret_cast<const u32x4*>(xb + st_off<1>(i, lane, pitch, r0 % 160, (c0 + 4096) % (T_bytes - 4096)));
                acc ^= l.x ^ l.w;
            }
        }
        v.x += 8;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int SHAPE, int MIX>
static void run(const char* name, char* y, const char* x, unsigned* out, long pitch, int C, int Zd, int waves, const char* where) {
    const int steps = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((push_kernel<SHAPE, MIX>), dim3(256), dim3(64 * waves), 0, 0, y, x, out, pitch, C, pitch, Zd, 64);
    hipEventRecord(a);
    hipLaunchKernelGGL((push_kernel<SHAPE, MIX>), dim3(256), dim3(64 * waves), 0, 0, y, x, out, pitch, C, pitch, Zd, steps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double st = 256.0 * waves * steps * 8.0 * 64.0 * lane_bytes<SHAPE>();
    const double ld = MIX > 0 ? 256.0 * waves * steps * (8.0 / MIX) * 64.0 * 16.0 : 0.0;
    printf("%-30s %-4s waves %d  loads 1 per %d stores: stores %6.2f TB/s = %5.1f B/clk/CU   loads %5.2f TB/s   total %5.2f TB/s\n", name, where, waves, MIX,
           st / (ms * 1e-3) / 1e12, st / (ms * 1e-3) / 256.0 / 2.1e9, ld / (ms * 1e-3) / 1e12, (st + ld) / (ms * 1e-3) / 1e12);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    const long pitch = 24000, C = 192, Z = 128;
    char *x, *y; unsigned* out;
    hipMalloc(&x, Z * C * pitch + (1 << 20)); hipMalloc(&y, Z * C * pitch + (1 << 20)); hipMalloc(&out, 4096);
    hipMemset(x, 1, Z * C * pitch + (1 << 20));
    for (int where = 0; where < 2; ++where) {
        const int Zd = where ? 1 : (int)Z;
        const char* wn = where ? "L2" : "HBM";
        for (int waves : {2, 4, 8}) {
            run<0, 0>("8 B/lane, 16 rows x 32 B", y, x, out, pitch, C, Zd, waves, wn);
            run<1, 0>("16 B/lane, 4 rows x 256 B", y, x, out, pitch, C, Zd, waves, wn);
            run<2, 0>("16 B/lane, 8 rows x 128 B", y, x, out, pitch, C, Zd, waves, wn);
            run<3, 0>("16 B/lane, 1 KB contiguous", y, x, out, pitch, C, Zd, waves, wn);
            run<1, 1>("16 B/lane, 4 rows x 256 B", y, x, out, pitch, C, Zd, waves, wn);
            run<1, 2>("16 B/lane, 4 rows x 256 B", y, x, out, pitch, C, Zd, waves, wn);
            run<1, 4>("16 B/lane, 4 rows x 256 B", y, x, out, pitch, C, Zd, waves, wn);
        }
    }
    return 0;
}
