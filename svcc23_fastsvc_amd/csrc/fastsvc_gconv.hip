// fastsvc_gconv.hip - gfx950: the grouped, strided k = 41 convolutions of the recipe's discriminator, forward, backward data
// and backward weight (BASELINE config 5).
//
// Reference: MelGANDiscriminator (harana/models/fastsvc.py:386-520; yaml egs/svcc23/fastsvc1/conf/fastsvc.yaml:34-52) stacks,
// per scale, `Conv1d(c, min(4 c, 512), kernel_size = 10 s + 1, stride = s, padding = 5 s, groups = c // 4)` + LeakyReLU(0.2)
// for s = 4: every group maps 4 input channels to 16 (8 at the 512-channel cap) output channels with 41 taps at stride 4.
// PyTorch-ROCm runs them as im2col per sample (512 `Im2d2Col_v2` launches per step) + small Tensile GEMMs + layout
// transposes + CK grouped kernels: ~24 of the 47 ms of kernel time of a training step (profiles/r4_cfg5_step_breakdown.txt)
// for 0.1 TFLOP of arithmetic.  Here each is ONE direct launch, float32 on the VALU as packed FMAs (two MACs per lane and
// instruction; the work is small: what counts is one pass over the tensors and no launches in between):
//
//   forward   thread = 4 consecutive outputs x all OG output channels of a group; its input window (53 samples per input
//             channel, 16-byte aligned: padding 20 = 5 s) sits in registers, the group's weights in LDS as [i][k][o] so that
//             16 bytes = 4 output channels are one wave-uniform read feeding 8 packed FMAs; bias and LeakyReLU fused
//             (the activated tensor is what the next layer and the feature-map list hold: the backward recovers the
//             derivative's sign from it).
//   backward data   thread = 16 consecutive input samples x 4 input channels; tap k meets exactly the four of them with
//             u + 20 = k (mod 4), so a weight read [o][k][4 i] feeds 8 packed FMAs as well; the LeakyReLU derivative is
//             applied to dy while its 20-sample window is loaded.
//   backward weight   thread = one (i, k) of a group (164 + one all-ones column for the bias gradient), 16 accumulators; dy of a
//             64-step tile staged in LDS as [t][o] (derivative applied); every (utterance, slab of 512 steps) writes its
//             own partial sums and a second launch adds them in a fixed order: bit-reproducible, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GC_IG = 4;          // input channels per group
constexpr int GC_K = 41;          // taps
constexpr int GC_S = 4;           // stride
constexpr int GC_PAD = 20;        // zero padding (a multiple of 4: windows start 16-byte aligned)
constexpr int GC_COLS = GC_IG * GC_K + 1;      // weight columns of a group + the all-ones column (bias gradient)
constexpr int GC_SLAB = 512;      // output steps per workgroup of the backward-weight kernel

// x[idx .. idx + 3] of a row of length T, zeros outside (idx may be negative)
__device__ __forceinline__ f32x4 gc_load4(const float* __restrict__ row, int idx, int T) {
    if (idx >= 0 && idx + 3 < T) {
        f32x4 v;
        __builtin_memcpy(&v, row + idx, 16);               // (rows of odd length: 4-byte aligned only)
        return v;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
    for (int e = 0; e < 4; ++e)
        if ((unsigned)(idx + e) < (unsigned)T) v[e] = row[idx + e];
    return v;
}

template <int OG>
__global__ __launch_bounds__(64)
void gconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                      int Cin, int Cout, int T, int Tout, float slope) {
    __shared__ f32x4 wl[GC_IG * GC_K * OG / 4];            // [i][k][o]
    const int g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    float* wls = reinterpret_cast<float*>(wl);
    for (int e = tid; e < OG * GC_IG * GC_K; e += 64) {
        const int o = e / (GC_IG * GC_K), r = e - o * (GC_IG * GC_K);
        wls[r * OG + o] = w[(long)(g * OG + o) * (GC_IG * GC_K) + r];
    }
    __syncthreads();
    const int t0 = (blockIdx.x * 64 + tid) * 4;
    if (t0 >= Tout) return;
    f32x2 acc[4][OG / 2];
    #pragma unroll
    for (int oq = 0; oq < OG / 2; ++oq) {
        const f32x2 bv = bias ? f32x2{bias[g * OG + 2 * oq], bias[g * OG + 2 * oq + 1]} : f32x2{0.f, 0.f};
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt][oq] = bv;
    }
    const int s0 = GC_S * t0 - GC_PAD;
    #pragma unroll 1
    for (int i = 0; i < GC_IG; ++i) {
        const float* row = x + ((long)b * Cin + g * GC_IG + i) * T;
        float xw[56];
        #pragma unroll
        for (int q = 0; q < 14; ++q) {
            const f32x4 v = gc_load4(row, s0 + 4 * q, T);
            xw[4 * q] = v.x; xw[4 * q + 1] = v.y; xw[4 * q + 2] = v.z; xw[4 * q + 3] = v.w;
        }
        const f32x4* wi = wl + i * GC_K * (OG / 4);
        #pragma unroll
        for (int k = 0; k < GC_K; ++k) {
            #pragma unroll
            for (int oq = 0; oq < OG / 4; ++oq) {
                const f32x4 w4 = wi[k * (OG / 4) + oq];    // (wave-uniform address: one broadcast read)
                const f32x2 wa = {w4.x, w4.y}, wb = {w4.z, w4.w};
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const float xv = xw[GC_S * tt + k];
                    const f32x2 x2 = {xv, xv};
                    acc[tt][2 * oq] = __builtin_elementwise_fma(wa, x2, acc[tt][2 * oq]);
                    acc[tt][2 * oq + 1] = __builtin_elementwise_fma(wb, x2, acc[tt][2 * oq + 1]);
                }
            }
        }
    }
    #pragma unroll
    for (int o = 0; o < OG; ++o) {
        float* yrow = y + ((long)b * Cout + g * OG + o) * Tout;
        float v[4];
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float a = acc[tt][o >> 1][o & 1];
            v[tt] = a > 0.f ? a : a * slope;
        }
        if (t0 + 3 < Tout) {
            const f32x4 v4 = {v[0], v[1], v[2], v[3]};
            __builtin_memcpy(yrow + t0, &v4, 16);
        } else {
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                if (t0 + tt < Tout) yrow[t0 + tt] = v[tt];
        }
    }
}

template <int OG>
__global__ __launch_bounds__(64)
void gconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yact, const float* __restrict__ w,
                        float* __restrict__ dx, int Cin, int Cout, int T, int Tout, float slope) {
    __shared__ f32x4 wl[OG * GC_K];                        // [o][k] -> the 4 input channels
    const int g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    for (int e = tid; e < OG * GC_K; e += 64) {
        const int o = e / GC_K, k = e - o * GC_K;
        const float* wp = w + (long)(g * OG + o) * (GC_IG * GC_K) + k;
        wl[e] = f32x4{wp[0], wp[GC_K], wp[2 * GC_K], wp[3 * GC_K]};
    }
    __syncthreads();
    const int u0 = (blockIdx.x * 64 + tid) * 16;
    if (u0 >= T) return;
    f32x2 acc[16][2];
    #pragma unroll
    for (int uu = 0; uu < 16; ++uu) { acc[uu][0] = f32x2{0.f, 0.f}; acc[uu][1] = f32x2{0.f, 0.f}; }
    const int tb = u0 / GC_S - 8;                          // first step of the 20-step dy window (a multiple of 4)
    #pragma unroll 1
    for (int o = 0; o < OG; ++o) {
        const long ro = ((long)b * Cout + g * OG + o) * Tout;
        float dv[20];
        #pragma unroll
        for (int q = 0; q < 5; ++q) {
            f32x4 d = gc_load4(dy + ro, tb + 4 * q, Tout);
            if (yact) {
                const f32x4 a = gc_load4(yact + ro, tb + 4 * q, Tout);
                #pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : d[e] * slope;
            }
            dv[4 * q] = d.x; dv[4 * q + 1] = d.y; dv[4 * q + 2] = d.z; dv[4 * q + 3] = d.w;
        }
        #pragma unroll
        for (int k = 0; k < GC_K; ++k) {
            const f32x4 w4 = wl[o * GC_K + k];
            const f32x2 wa = {w4.x, w4.y}, wb = {w4.z, w4.w};
            #pragma unroll
            for (int uu = 0; uu < 16; ++uu) {
                if (((uu + GC_PAD - k) & 3) == 0) {        // u + pad = S t + k  <=>  this tap meets this sample
                    const float d = dv[((uu + GC_PAD - k) >> 2) + 8];
                    const f32x2 d2 = {d, d};
                    acc[uu][0] = __builtin_elementwise_fma(wa, d2, acc[uu][0]);
                    acc[uu][1] = __builtin_elementwise_fma(wb, d2, acc[uu][1]);
                }
            }
        }
    }
    #pragma unroll
    for (int i = 0; i < GC_IG; ++i) {
        float* xrow = dx + ((long)b * Cin + g * GC_IG + i) * T;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v4 = {acc[4 * q][i >> 1][i & 1], acc[4 * q + 1][i >> 1][i & 1], acc[4 * q + 2][i >> 1][i & 1], acc[4 * q + 3][i >> 1][i & 1]};
            const int u = u0 + 4 * q;
            if (u + 3 < T) __builtin_memcpy(xrow + u, &v4, 16);
            else {
                #pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (u + e < T) xrow[u + e] = v4[e];
            }
        }
    }
}

template <int OG>
__global__ __launch_bounds__(192)
void gconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ yact,
                        float* __restrict__ part, int Cin, int Cout, int T, int Tout, float slope) {
    __shared__ f32x4 dt[64 * OG / 4];                      // [t][o] of a 64-step tile, LeakyReLU derivative applied
    const int slab = blockIdx.x, g = blockIdx.y, b = blockIdx.z, j = threadIdx.x;
    const int nslab = gridDim.x, G = gridDim.y;
    const bool col = j < GC_COLS;                          // 164 weight columns + the ones column
    const bool ones = j == GC_COLS - 1;
    const int i = ones ? 0 : j / GC_K, k = j - i * GC_K;
    const float* xrow = x + ((long)b * Cin + g * GC_IG + (col ? i : 0)) * T;
    f32x2 acc[OG / 2];
    #pragma unroll
    for (int oq = 0; oq < OG / 2; ++oq) acc[oq] = f32x2{0.f, 0.f};
    float* dts = reinterpret_cast<float*>(dt);
    const int t_end = min(Tout, (slab + 1) * GC_SLAB);
    for (int tt0 = slab * GC_SLAB; tt0 < t_end; tt0 += 64) {
        __syncthreads();
        for (int e = j; e < 64 * OG; e += 192) {
            const int o = e >> 6, tl = e & 63, t = tt0 + tl;
            float v = 0.f;
            if (t < t_end) {
                const long a = ((long)b * Cout + g * OG + o) * Tout + t;
                v = dy[a];
                if (yact && !(yact[a] > 0.f)) v *= slope;
            }
            dts[tl * OG + o] = v;
        }
        __syncthreads();
        if (col) {
            #pragma unroll 1
            for (int tl = 0; tl < 64; tl += 8) {
                float xv[8];
                #pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = GC_S * (tt0 + tl + e) + k - GC_PAD;
                    xv[e] = ones ? 1.f : ((unsigned)idx < (unsigned)T ? xrow[idx] : 0.f);
                }
                #pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const f32x2 x2 = {xv[e], xv[e]};
                    #pragma unroll
                    for (int oq = 0; oq < OG / 4; ++oq) {
                        const f32x4 d4 = dt[(tl + e) * (OG / 4) + oq];
                        acc[2 * oq] = __builtin_elementwise_fma(f32x2{d4.x, d4.y}, x2, acc[2 * oq]);
                        acc[2 * oq + 1] = __builtin_elementwise_fma(f32x2{d4.z, d4.w}, x2, acc[2 * oq + 1]);
                    }
                }
            }
        }
    }
    if (col) {
        float* p = part + (((long)(b * nslab + slab) * G + g) * OG) * GC_COLS + j;
        #pragma unroll
        for (int o = 0; o < OG; ++o) p[o * GC_COLS] = acc[o >> 1][o & 1];
    }
}

// dw[row][0 .. 163], db[row] = sum over the P partial sets, in a fixed order (row = output channel)
__global__ __launch_bounds__(256)
void gconv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int P, int rows) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)rows * GC_COLS;
    if (e >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < P; p += 4) {
        s0 += part[(long)p * n + e]; s1 += part[(long)(p + 1) * n + e];
        s2 += part[(long)(p + 2) * n + e]; s3 += part[(long)(p + 3) * n + e];
    }
    for (; p < P; ++p) s0 += part[(long)p * n + e];
    const float s = (s0 + s1) + (s2 + s3);
    const int row = (int)(e / GC_COLS), c = (int)(e - (long)row * GC_COLS);
    if (c == GC_COLS - 1) { if (db) db[row] = s; }
    else dw[(long)row * (GC_COLS - 1) + c] = s;
}

bool gc_supported(int Cin, int Cout, int groups, int K, int stride, int pad) {
    if (groups < 1 || Cin % groups || Cout % groups) return false;
    const int ig = Cin / groups, og = Cout / groups;
    return ig == GC_IG && (og == 8 || og == 16) && K == GC_K && stride == GC_S && pad == GC_PAD;
}
int gc_out_len(int T) { return (T + 2 * GC_PAD - GC_K) / GC_S + 1; }

}  // namespace

extern "C" {

int fastsvc_gconv1d_supported(int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad) {
    return gc_supported(Cin, Cout, groups, K, stride, pad) ? 1 : 0;
}

int fastsvc_gconv1d_forward(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t Cout,
                            int32_t groups, int32_t T, int32_t K, int32_t stride, int32_t pad, float slope, void* stream_) {
    if (!x || !w || !y || B < 1 || T < 1) return FASTSVC_E_INVALID;
    if (!gc_supported(Cin, Cout, groups, K, stride, pad) || B > 65535 || groups > 65535) return FASTSVC_E_UNSUPPORTED;
    const int Tout = gc_out_len(T);
    if (Tout < 1) return FASTSVC_E_INVALID;
    const dim3 grid((Tout + 255) / 256, groups, B);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (Cout / groups == 16) hipLaunchKernelGGL(gconv_fwd_kernel<16>, grid, dim3(64), 0, st, x, w, bias, y, (int)Cin, (int)Cout, (int)T, Tout, slope);
    else hipLaunchKernelGGL(gconv_fwd_kernel<8>, grid, dim3(64), 0, st, x, w, bias, y, (int)Cin, (int)Cout, (int)T, Tout, slope);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

int fastsvc_gconv1d_backward_data(const float* dy, const float* y_act, const float* w, float* dx, int32_t B, int32_t Cin,
                                  int32_t Cout, int32_t groups, int32_t T, int32_t K, int32_t stride, int32_t pad, float slope,
                                  void* stream_) {
    if (!dy || !w || !dx || B < 1 || T < 1) return FASTSVC_E_INVALID;
    if (!gc_supported(Cin, Cout, groups, K, stride, pad) || B > 65535 || groups > 65535) return FASTSVC_E_UNSUPPORTED;
    const int Tout = gc_out_len(T);
    if (Tout < 1) return FASTSVC_E_INVALID;
    const dim3 grid((T + 1023) / 1024, groups, B);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (Cout / groups == 16) hipLaunchKernelGGL(gconv_dgrad_kernel<16>, grid, dim3(64), 0, st, dy, y_act, w, dx, (int)Cin, (int)Cout, (int)T, Tout, slope);
    else hipLaunchKernelGGL(gconv_dgrad_kernel<8>, grid, dim3(64), 0, st, dy, y_act, w, dx, (int)Cin, (int)Cout, (int)T, Tout, slope);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

size_t fastsvc_gconv1d_backward_weight_scratch_bytes(int32_t B, int32_t Cout, int32_t T) {
    if (B < 1 || Cout < 1 || T < 1) return 0;
    const int Tout = gc_out_len(T);
    const long nslab = (Tout + GC_SLAB - 1) / GC_SLAB;
    return (size_t)B * nslab * Cout * GC_COLS * sizeof(float);
}

int fastsvc_gconv1d_backward_weight(const float* x, const float* dy, const float* y_act, float* dw, float* dbias, void* scratch,
                                    int32_t B, int32_t Cin, int32_t Cout, int32_t groups, int32_t T, int32_t K, int32_t stride,
                                    int32_t pad, float slope, void* stream_) {
    if (!x || !dy || !dw || !scratch || B < 1 || T < 1) return FASTSVC_E_INVALID;
    if (!gc_supported(Cin, Cout, groups, K, stride, pad) || B > 65535 || groups > 65535) return FASTSVC_E_UNSUPPORTED;
    const int Tout = gc_out_len(T);
    if (Tout < 1) return FASTSVC_E_INVALID;
    const int nslab = (Tout + GC_SLAB - 1) / GC_SLAB;
    float* part = static_cast<float*>(scratch);
    const dim3 grid(nslab, groups, B);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (Cout / groups == 16) hipLaunchKernelGGL(gconv_wgrad_kernel<16>, grid, dim3(192), 0, st, x, dy, y_act, part, (int)Cin, (int)Cout, (int)T, Tout, slope);
    else hipLaunchKernelGGL(gconv_wgrad_kernel<8>, grid, dim3(192), 0, st, x, dy, y_act, part, (int)Cin, (int)Cout, (int)T, Tout, slope);
    if (hipGetLastError() != hipSuccess) return FASTSVC_E_HIP;
    const long n = (long)Cout * GC_COLS;
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, dw, dbias, B * nslab, (int)Cout);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
