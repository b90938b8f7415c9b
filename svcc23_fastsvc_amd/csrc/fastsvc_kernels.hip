// fastsvc_kernels.hip - gfx950 (CDNA4 / MI355X) kernels of the FastSVC generator forward pass.
//
// Hot op: k=3 dilated "same" convolution over (B, C, T) float32 tensors
// (reference: Conv1d1x3 / Conv2d1x3, harana/layers/upsample.py:76-83,99-106, used by
// harana/models/fastsvc.py:56-75,164-178,209-218) with everything the generator wraps around it
// fused into the same launch:
//   prologue (while staging the input window into LDS):
//       nearest decimation / nearest stretch as an index map (Squeeze2d / Stretch2d),
//       FiLM affine + InstanceNorm-apply + speaker bias (fastsvc.py:115-140), LeakyReLU(0.2),
//       zero "same" padding at the utterance edges (never bleeding across batch items);
//   main loop: implicit GEMM  D[t][co] += X[t + (tap-1)*d][ci] * W[co][ci][tap]  on the f32-input
//       MFMA v_mfma_f32_16x16x4_f32 (exact f32, bitwise an fmaf chain) - M = 16 time columns,
//       N = 16 output channels, K = 4 input channels per instruction; activations come from the
//       LDS window (ds_read_b32, conflict-free because the row stride is 16 mod 32), weights are
//       streamed from L2 in pre-packed fragment order straight into VGPRs;
//   epilogue: bias, LeakyReLU, residual add (tensor or rank-1), float4 stores, and the
//       per-(b, c) sum / sum-of-squares of the NEXT FiLM-affine's output for InstanceNorm
//       (wave shuffle -> LDS f64 -> one f64 atomic per channel per workgroup).
//
// Wavefront = 64 lanes; one workgroup = 4 waves (one per SIMD), each wave owns NW 16-column time
// tiles x MW 16-channel tiles of accumulators.  No CUDA-isms, no dual paths: gfx950 only.
#include "fastsvc_kernels.h"

namespace fastsvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v) { return v >= 0.f ? v : LRELU_SLOPE * v; }

__device__ __forceinline__ int div_small(int t, int s) {
    // nearest-stretch source index; s in {2,4,5} for the yaml config - keep those divisions cheap
    switch (s) {
        case 2: return t >> 1;
        case 4: return t >> 2;
        case 5: return t / 5;
        case 1: return t;
        default: return t / s;
    }
}

template <int MW>
__device__ __forceinline__ void load_wfrag(float (&w)[MW], const float* p) {
    if constexpr (MW == 1) {
        w[0] = p[0];
    } else if constexpr (MW == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        w[0] = v.x; w[1] = v.y;
    } else if constexpr (MW == 3) {
        // 12-byte fragment (global_load_dwordx3); base is 4-byte aligned only
        w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
    } else {
        #pragma unroll
        for (int m = 0; m < MW; ++m) w[m] = p[m];
    }
}

// ---------------------------------------------------------------------------------------------
// Generic convolution.  gridDim = (ceil(T / NT), ceil(ngroups / WM), nsig * B)
// ---------------------------------------------------------------------------------------------
template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_mfma_kernel(const ConvParams p) {
    constexpr int NWAVES = WM * WN;
    constexpr int NTHREADS = 64 * NWAVES;
    constexpr int NT = 16 * NW * WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN;
    const int wave_n = wave - wave_m * WN;
    const int z = blockIdx.z;
    const int sig = z / p.B;
    const int b = z - sig * p.B;
    const int t0 = blockIdx.x * NT;
    const int mg = blockIdx.y * WM + wave_m;
    const bool active = mg < p.ngroups;
    const int halo = (p.ntaps == 3) ? p.dil : 0;
    const int W = NT + 2 * halo;
    const int XS = p.xs;
    const int CINp = p.nchunks * p.KC;

    // LDS carve-up (all offsets multiples of 16 bytes)
    double* sstat = reinterpret_cast<double*>(smem_raw);                       // [WM*MW*16][2]
    float* nmean = reinterpret_cast<float*>(smem_raw + sizeof(double) * 2 * 16 * MW * WM);
    float* nrstd = nmean + CINp;
    float* nspk = nrstd + CINp;
    float* Xs = nspk + CINp;                                                   // [KC][XS]
    // (CINp is a multiple of 4, so Xs stays 16-byte aligned)

    if (p.flags & F_STATS) {
        for (int i = tid; i < 2 * 16 * MW * WM; i += NTHREADS) sstat[i] = 0.0;
    }
    if (p.flags & F_PRE_NORM) {
        const double inv_len = 1.0 / (double)p.x_T;
        for (int c = tid; c < CINp; c += NTHREADS) {
            float m = 0.f, r = 0.f, s = 0.f;
            if (c < p.CIN) {
                const double s1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
                const double s2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
                const double mean = s1 * inv_len;
                double var = s2 * inv_len - mean * mean;      // biased variance (InstanceNorm2d)
                var = var > 0.0 ? var : 0.0;
                m = (float)mean;
                r = (float)(1.0 / sqrt(var + IN_EPS));
                s = p.spk[(long)b * p.CIN + c];
            }
            nmean[c] = m; nrstd[c] = r; nspk[c] = s;
        }
    }

    f32x4 acc[NW][MW];
    #pragma unroll
    for (int n = 0; n < NW; ++n)
        #pragma unroll
        for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* xbase = p.x + (long)sig * p.x_sig + (long)b * p.x_b;
    const float* ssbase = (p.flags & F_PRE_AFFINE) ? p.ss_in + (long)b * p.ss_in_b : nullptr;
    const int kg = p.KC >> 2;
    const int steps = p.ntaps * kg;
    const float* wchunk = p.w + (long)sig * p.w_sig + ((long)mg * p.Q * 64 + lane) * MW;
    const int flags = p.flags;
    const int mode = p.mode;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        __syncthreads();   // previous chunk's LDS reads are done (first pass: norm constants visible)
        // ---- stage KC input rows [t0 - halo, t0 + NT + halo) with the prologue applied ----
        for (int r = wave; r < p.KC; r += NWAVES) {
            const int ci = ch * p.KC + r;
            float* row = Xs + r * XS;
            if (ci >= p.CIN) {
                for (int j = lane; j < W; j += 64) row[j] = 0.f;
                continue;
            }
            const float* xrow = xbase + (long)ci * p.x_T;
            const float* scrow = nullptr;
            const float* shrow = nullptr;
            float mean = 0.f, rstd = 1.f, pb = 0.f;
            if (flags & F_PRE_AFFINE) {
                scrow = ssbase + (long)ci * p.x_T;
                shrow = ssbase + (long)(p.CIN + ci) * p.x_T;
                if (flags & F_PRE_NORM) { mean = nmean[ci]; rstd = nrstd[ci]; pb = nspk[ci]; }
            }
            for (int j = lane; j < W; j += 64) {
                const int t = t0 - halo + j;
                float v = 0.f;
                if (t >= 0 && t < p.T) {
                    const int src = (mode == MODE_DIRECT) ? t
                                  : (mode == MODE_DECIMATE) ? t * p.s : div_small(t, p.s);
                    v = xrow[src];
                    if (flags & F_PRE_AFFINE) {
                        v = scrow[src] * v + shrow[src];
                        if (flags & F_PRE_NORM) v = (v - mean) * rstd + pb;
                    }
                    if (flags & F_PRE_LRELU) v = lrelu(v);
                }
                row[j] = v;
            }
        }
        __syncthreads();
        // ---- implicit GEMM over this chunk: steps = ntaps * KC/4 MFMA k-steps ----
        if (active) {
            const float* wp = wchunk + (long)ch * steps * 64 * MW;
            const float* xa0 = Xs + (lane >> 4) * XS + (lane & 15) + wave_n * (NW * 16);
            float wcur[MW];
            load_wfrag<MW>(wcur, wp);
            int g = 0, tap = 0;
            const float* xa = xa0;
            for (int st = 0; st < steps; ++st) {
                float wnext[MW];
                #pragma unroll
                for (int m = 0; m < MW; ++m) wnext[m] = 0.f;
                if (st + 1 < steps) load_wfrag<MW>(wnext, wp + (long)(st + 1) * 64 * MW);
                float av[NW];
                #pragma unroll
                for (int n = 0; n < NW; ++n) av[n] = xa[n * 16];
                #pragma unroll
                for (int n = 0; n < NW; ++n)
                    #pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n], wcur[m], acc[n][m], 0, 0, 0);
                #pragma unroll
                for (int m = 0; m < MW; ++m) wcur[m] = wnext[m];
                ++g;
                xa += 4 * XS;
                if (g == kg) { g = 0; ++tap; xa = xa0 + tap * p.dil; }
            }
        }
    }

    // ---- epilogue ----
    // D layout (16x16x4 f32): lane holds column j = lane & 15 (output channel) and rows
    // i = (lane >> 4) * 4 + r (time), r = 0..3  ->  four consecutive time steps per lane.
    float s1[MW], s2[MW];
    #pragma unroll
    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }

    if (active) {
        const float* biasp = p.bias + (long)sig * p.bias_sig;
        float* ybase = p.y + (long)sig * p.y_sig + (long)b * p.y_b;
        const float* resbase = p.res ? p.res + (long)sig * p.res_sig + (long)b * p.res_b : nullptr;
        const float* r1x = p.r1x ? p.r1x + (long)sig * p.r1x_sig + (long)b * p.r1x_b : nullptr;
        const float* ssob = (flags & F_STATS) ? p.ss_out + (long)b * p.ss_out_b : nullptr;
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int co = (mg * MW + m) * 16 + (lane & 15);
            if (co >= p.COUT) continue;
            const float bias = biasp[co];
            float r1w = 0.f, r1b = 0.f;
            if (r1x) { r1w = p.r1w[(long)sig * p.r1_sig + co]; r1b = p.r1b[(long)sig * p.r1_sig + co]; }
            float* yrow = ybase + (long)co * p.T;
            const float* rrow = resbase ? resbase + (long)co * p.T : nullptr;
            const float* scrow = ssob ? ssob + (long)co * p.T : nullptr;
            const float* shrow = ssob ? ssob + (long)(p.COUT + co) * p.T : nullptr;
            #pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int t = t0 + wave_n * (NW * 16) + n * 16 + (lane >> 4) * 4;
                if (t >= p.T) continue;
                f32x4 v = acc[n][m];
                v += bias;
                if (flags & F_POST_LRELU) {
                    v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w);
                }
                if (p.vec) {
                    if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + t);
                    if (r1x) v += *reinterpret_cast<const f32x4*>(r1x + t) * r1w + r1b;
                    *reinterpret_cast<f32x4*>(yrow + t) = v;
                    if (scrow) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(scrow + t) * v
                                      + *reinterpret_cast<const f32x4*>(shrow + t);
                        s1[m] += (u.x + u.y) + (u.z + u.w);
                        s2[m] += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
                    }
                } else {
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (t + r >= p.T) break;
                        float e = v[r];
                        if (rrow) e += rrow[t + r];
                        if (r1x) e += r1x[t + r] * r1w + r1b;
                        yrow[t + r] = e;
                        if (scrow) {
                            const float u = scrow[t + r] * e + shrow[t + r];
                            s1[m] += u; s2[m] += u * u;
                        }
                    }
                }
            }
        }
    }

    if (flags & F_STATS) {
        // reduce over the 4 lane groups that share a channel, then over the WN waves via LDS f64
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            double d1 = (double)s1[m], d2 = (double)s2[m];
            d1 += __shfl_xor(d1, 16); d2 += __shfl_xor(d2, 16);
            d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
            if (active && lane < 16) {
                const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                atomicAdd(&sstat[slot + 0], d1);
                atomicAdd(&sstat[slot + 1], d2);
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += NTHREADS) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}

template <int MW, int NW>
static hipError_t launch_conv_t(const ConvParams& p, int nsig, hipStream_t stream) {
    constexpr int WM = 1, WN = 4;
    constexpr int NT = 16 * NW * WN;
    dim3 grid((p.T + NT - 1) / NT, (p.ngroups + WM - 1) / WM, nsig * p.B);
    dim3 block(64 * WM * WN);
    const int CINp = p.nchunks * p.KC;
    const size_t smem = sizeof(double) * 2 * 16 * MW * WM + sizeof(float) * (3 * (size_t)CINp + (size_t)p.KC * p.xs);
    hipLaunchKernelGGL((conv_mfma_kernel<MW, NW, WM, WN>), grid, block, smem, stream, p);
    return hipGetLastError();
}

int conv_tile_columns(int NW) { return 16 * NW * 4; }

hipError_t launch_conv(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
#define FASTSVC_CASE(mw, nw) if (cfg.MW == mw && cfg.NW == nw) return launch_conv_t<mw, nw>(p, cfg.nsig, stream);
    FASTSVC_CASE(1, 1) FASTSVC_CASE(1, 2) FASTSVC_CASE(1, 4)
    FASTSVC_CASE(2, 1) FASTSVC_CASE(2, 2) FASTSVC_CASE(2, 4)
    FASTSVC_CASE(3, 1) FASTSVC_CASE(3, 2) FASTSVC_CASE(3, 4)
#undef FASTSVC_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Down-sampling stage 0, first conv: C_in = 1 (fastsvc.py:173, downsample_block.2 of net 0).
// K = 3 only: a VALU kernel; every thread produces 4 consecutive samples for all C channels.
// HBM-write bound (C floats written per float read).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void in1_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                     const float* __restrict__ bias, long w_sig, long b_sig, float* __restrict__ y,
                     int B, int C, int T) {
    const int z = blockIdx.z;
    const int sig = z / B;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const float* xr = x + (long)z * T;
    float xv[6];
    #pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int tt = t - 1 + i;
        xv[i] = (tt >= 0 && tt < T) ? lrelu(xr[tt]) : 0.f;
    }
    const float* ws = w + sig * w_sig;
    const float* bs = bias + sig * b_sig;
    float* yb = y + (long)z * C * T;
    const bool full = (t + 3 < T) && ((T & 3) == 0);
    for (int co = 0; co < C; ++co) {
        const float w0 = ws[co * 3 + 0], w1 = ws[co * 3 + 1], w2 = ws[co * 3 + 2], bb = bs[co];
        float o[4];
        #pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = bb + (w0 * xv[i] + w1 * xv[i + 1]) + w2 * xv[i + 2];
        float* yr = yb + (long)co * T + t;
        if (full) {
            *reinterpret_cast<f32x4*>(yr) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
            for (int i = 0; i < 4 && t + i < T; ++i) yr[i] = o[i];
        }
    }
}

hipError_t launch_in1_conv(const float* x, const float* w, const float* bias, long w_sig, long b_sig,
                           float* y, int nsig, int B, int C, int T, hipStream_t stream) {
    dim3 grid((T + 1023) / 1024, 1, nsig * B);
    hipLaunchKernelGGL(in1_conv_kernel, grid, dim3(256), 0, stream, x, w, bias, w_sig, b_sig, y, B, C, T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv_last: 1x1 conv C -> O (fastsvc.py:301,330), HBM-read bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void pointwise_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                          const float* __restrict__ bias, float* __restrict__ y,
                          int C, int O, int T) {
    const int b = blockIdx.z;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const float* xb = x + (long)b * C * T;
    const bool full = (t + 3 < T) && ((T & 3) == 0);
    for (int o = 0; o < O; ++o) {
        f32x4 acc = f32x4{bias[o], bias[o], bias[o], bias[o]};
        if (full) {
            for (int c = 0; c < C; ++c)
                acc += *reinterpret_cast<const f32x4*>(xb + (long)c * T + t) * w[o * C + c];
            *reinterpret_cast<f32x4*>(y + ((long)b * O + o) * T + t) = acc;
        } else {
            for (int i = 0; i < 4 && t + i < T; ++i) {
                float a = bias[o];
                for (int c = 0; c < C; ++c) a += xb[(long)c * T + t + i] * w[o * C + c];
                y[((long)b * O + o) * T + t + i] = a;
            }
        }
    }
}

hipError_t launch_pointwise_out(const float* x, const float* w, const float* bias, float* y,
                                int B, int C, int O, int T, hipStream_t stream) {
    dim3 grid((T + 1023) / 1024, 1, B);
    hipLaunchKernelGGL(pointwise_out_kernel, grid, dim3(256), 0, stream, x, w, bias, y, C, O, T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Speaker bias p = Linear(F.normalize(spk_emb)) for every up block (fastsvc.py:135-137).
// grid = (B, nblocks); one wave per output channel, lanes stride over the embedding.
// ---------------------------------------------------------------------------------------------
struct SpkArgs {
    SpkBlock blk[8];
};

__global__ __launch_bounds__(256)
void spk_proj_kernel(const float* __restrict__ emb, const SpkArgs args, int E) {
    extern __shared__ __attribute__((aligned(16))) float e_s[];   // [E] normalised embedding
    __shared__ float red[4];
    const int b = blockIdx.x;
    const SpkBlock blk = args.blk[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* e = emb + (long)b * E;
    float ss = 0.f;
    for (int i = tid; i < E; i += 256) { const float v = e[i]; ss += v * v; }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);
    for (int i = tid; i < E; i += 256) e_s[i] = e[i] / nrm;
    __syncthreads();
    for (int c = wave; c < blk.C; c += 4) {
        const float* wr = blk.w + (long)c * E;
        float a = 0.f;
        for (int i = lane; i < E; i += 64) a += wr[i] * e_s[i];
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) blk.out[(long)b * blk.C + c] = a + blk.bias[c];
    }
}

hipError_t launch_spk_proj(const float* emb, const SpkBlock* blocks, int nblocks, int B, int E,
                           hipStream_t stream) {
    if (nblocks > 8) return hipErrorInvalidValue;
    SpkArgs args;
    for (int i = 0; i < nblocks; ++i) args.blk[i] = blocks[i];
    for (int i = nblocks; i < 8; ++i) args.blk[i] = blocks[0];
    hipLaunchKernelGGL(spk_proj_kernel, dim3(B, nblocks), dim3(256), sizeof(float) * E, stream,
                       emb, args, E);
    return hipGetLastError();
}

}  // namespace fastsvc
