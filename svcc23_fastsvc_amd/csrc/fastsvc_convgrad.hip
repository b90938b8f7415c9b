// fastsvc_convgrad.hip - the convolution kernels of the generator's BACKWARD pass on gfx950.
//
// The reference trains the generator under autograd (harana/bin/train_fastsvc.py:157-205: y_ = generator(*x), losses,
// gen_loss.backward()); every convolution of the generator is a stride-1 "same" Conv1d / Conv2d(1 x k) with k = 1 or 3
// and dilation 1 ... 27 (harana/layers/residual_block.py:27-48, harana/models/fastsvc.py:34-232, the decimation is a
// slice in front of the conv).  For y = conv(x, W) + bias autograd needs
//   dx[b, ci, t]   = sum_{co, k} W[co, ci, k] dy[b, co, t - (k - h) d]          "backward data": the SAME convolution with
//                                                                                the weight transposed and its taps flipped
//   dW[co, ci, k]  = sum_{b, t} dy[b, co, t] x[b, ci, t + (k - h) d]            "backward weight"
//   dbias[co]      = sum_{b, t} dy[b, co, t]
// in float32 (master weights and gradients are float32).  Both run on the matrix cores with float32 operands
// (v_mfma_f32_16x16x4_f32: exact float32 products, float32 accumulation - the arithmetic of an fmaf chain):
//
//  conv1d_fwd_kernel   one workgroup = 128 time steps x 16 ... 48 output channels of one utterance.  The input window
//      (16 channels at a time, halo included) and the matching weight slice sit in LDS; a wave owns 32 time steps and
//      all the workgroup's channel tiles: per 4-deep reduction step one LDS read per operand row, 2 x (up to 4) MFMAs.
//      `transposed` reads the weight as [in][out][k] with flipped taps: backward data without a transposed copy.
//      Also the forward convolution of the recomputed dataflow (the generator's own forward is the fused HIP path).
//  conv1d_wgrad_kernel one workgroup = a slab of time steps of one utterance x 32 output x 32 input channels; wave =
//      one 16 x 16 (co, ci) tile, all k taps: the reduction runs over time, 4 steps per MFMA, both operands read from
//      LDS tiles with a row pitch = 2 (mod 32) floats (conflict-free column reads).  Every slab writes its own
//      partial dW / dbias; a second launch adds the slabs in order: no atomics (thousands of same-address float atomics
//      were 5x the kernel's own time), bit-reproducible gradients.
//  Both kernels fetch the next chunk / tile into registers while the matrix cores work on the current one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CG_THREADS = 256;
constexpr int CG_TT = 128;             // time steps per workgroup tile
// weight LDS row pitch for NCT 16-channel tiles: = 17 (mod 32) - odd (staging writes run down a column), and the 2 x 16 lanes
// of an operand read (rows r, r + 1) land on 32 different banks but one
constexpr int cg_ws(int nct) { return nct <= 1 ? 17 : 49; }
constexpr int CG_MAX_HALO = 27;        // (k / 2) * dilation
constexpr int CG_XS = CG_TT + 2 * CG_MAX_HALO + 2;     // 184

// CC = input channels per staged chunk: 16, or 24 for the 24-channel layers (no zero rows), or 8 for fewer than 16
// (four waves per SIMD = four workgroups per CU, with buffer loads whose row offsets are scalars: left alone the compiler hoists every LDS read of the unrolled reduction
// and takes 192 ... 260 registers, i.e. one or two workgroups per CU with nothing to hide a workgroup's fetch behind)
template <int K, int CC, int NCT>
__global__ __launch_bounds__(CG_THREADS, 4)
void conv1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ y, int Cin, int Cout, int T, int dil, int transposed, int tpw) {
    constexpr int RC = CC * K;                          // reduction rows per chunk
    constexpr int XR = CC / 4;                          // input rows per wave
    constexpr int CT = NCT * 16;                        // output channels per workgroup
    constexpr int WQ = (CT * RC + CG_THREADS - 1) / CG_THREADS;
    __shared__ float xl[CC][CG_XS];
    __shared__ float wl[RC][cg_ws(NCT)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2) in launch order: remap so that one XCD
    // gets a CONTIGUOUS run of (time tile, channel group) items of an utterance - the channel groups of a tile re-read the same
    // input window and neighbouring tiles share their halos, which then stay in that XCD's L2
    // (items are numbered channel group fastest; the launch-order index d of the first 8 * (n / 8) workgroups is permuted onto
    // them, the remaining n % 8 keep theirs - one bijection over all n items)
    int bx, by;
    {
        const int nxy = gridDim.x * gridDim.y, d = blockIdx.x + gridDim.x * blockIdx.y, per = nxy >> 3;
        const int w = d < (per << 3) ? (d & 7) * per + (d >> 3) : d;
        bx = w / (int)gridDim.y;
        by = w - bx * (int)gridDim.y;
    }
    const int co0 = by * CT, b = blockIdx.z;
    const int halo = (K / 2) * dil, xw = CG_TT + 2 * halo;
    const float* xb = x + (long)b * Cin * T;
    float* yb = y + (long)b * Cout * T;
    const int n = lane & 15, kk = lane >> 4;
    const int tiles = (T + CG_TT - 1) / CG_TT;
    const int tile0 = bx * tpw, ntile = min(tpw, tiles - tile0);
    const int nchunk = (Cin + CC - 1) / CC;
    f32x4 acc[NCT][2];
    #pragma unroll
    for (int i = 0; i < NCT; ++i)
        #pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the next (tile, chunk) travels global -> registers while the matrix cores work on the current one
    // (buffer loads: one lane offset per 64-column group for ALL rows - the row is a scalar offset - and one 32-bit offset
    // per weight value; lanes outside the utterance carry an offset beyond the descriptor's range and read 0)
    float xr[XR][3], wr[WQ];
    constexpr int OOB = 0x7fffff00;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, Cin * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, Cin * Cout * K * 4, 0x00020000);
    auto fetch = [&](int tile, int ci0) {
        const int t0 = tile * CG_TT;
        int voff[3];
        #pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int col = lane + 64 * c, t = t0 - halo + col;
            voff[c] = (col < xw && t >= 0 && t < T) ? t * 4 : OOB;
        }
        #pragma unroll
        for (int rr = 0; rr < XR; ++rr) {
            const int ci = __builtin_amdgcn_readfirstlane(ci0 + wave + 4 * rr);
            #pragma unroll
            for (int c = 0; c < 3; ++c)
                xr[rr][c] = ci < Cin ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, voff[c], ci * T * 4, 0)) : 0.f;
        }
        #pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = tid + CG_THREADS * q;
            const int co = idx / RC, j = idx - co * RC;
            const int cil = j / K, k = j - cil * K;
            const int ci = ci0 + cil, o = co0 + co;
            const int off = transposed ? ((ci * Cout + o) * K + (K - 1 - k)) * 4 : ((o * Cin + ci) * K + k) * 4;
            wr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wres, (idx < CT * RC && ci < Cin && o < Cout) ? off : OOB, 0, 0));
        }
    };
    fetch(tile0, 0);
    int tile = tile0, chunk = 0;
    for (int it = 0; it < ntile * nchunk; ++it) {
        __syncthreads();                                // the previous chunk's readers are done
        #pragma unroll
        for (int rr = 0; rr < XR; ++rr)
            #pragma unroll
            for (int c = 0; c < 3; ++c)
                if (lane + 64 * c < CG_XS) xl[wave + 4 * rr][lane + 64 * c] = xr[rr][c];
        #pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = tid + CG_THREADS * q;
            const int co = idx / RC, j = idx - co * RC;
            if (idx < CT * RC) wl[j][co] = wr[q];
        }
        __syncthreads();
        const bool last = chunk + 1 == nchunk;
        if (it + 1 < ntile * nchunk) fetch(last ? tile + 1 : tile, last ? 0 : (chunk + 1) * CC);
        // the channel-tile count is a template argument (a wave-uniform branch per tile kept every LDS read's latency in
        // front of its two MFMAs); three steps in flight (fully unrolled the LDS reads of all steps are hoisted and spill)
        #pragma unroll 3
        for (int r0 = 0; r0 < RC; r0 += 4) {
            const int r = r0 + kk;
            const int cil = r / K, k = r - cil * K;
            const float* xrow = &xl[cil][wave * 32 + n + k * dil];
            const float b0 = xrow[0], b1 = xrow[16];
            #pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const float a = wl[r][i * 16 + n];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[i][1], 0, 0, 0);
            }
        }
        if (last) {
            // D[m = 4 (lane / 16) + e][n = lane % 16]: channel co0 + 16 i + m, time t0 + 32 wave + 16 j + n
            const int t0 = tile * CG_TT;
            #pragma unroll
            for (int i = 0; i < NCT; ++i) {
                #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = co0 + i * 16 + 4 * kk + e;
                    const float bv = (bias && o < Cout) ? bias[o] : 0.f;
                    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int t = t0 + wave * 32 + j * 16 + n;
                        if (o < Cout && t < T) yb[(long)o * T + t] = acc[i][j][e] + bv;
                    }
                }
                acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            ++tile;
            chunk = 0;
        } else {
            ++chunk;
        }
    }
}

constexpr int CW_DS = CG_TT + 2;                       // 130 = 2 (mod 32): lanes (row n, column kk) of a ds_read_b32 half
constexpr int CW_XS = 194;                             // >= 128 + 54, = 2 (mod 32)   hit 32 different banks

// partial[slab][Cout * Cin * K + Cout]: every (slab, co, ci, k) and (slab, co) entry is written by exactly one wave
template <int K>
__global__ __launch_bounds__(CG_THREADS)
void conv1d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                         int Cin, int Cout, int T, int dil, int slab_tiles, int slabs_per_b, int ci_groups) {
    __shared__ float dyl[32][CW_DS];
    __shared__ float xl[32][CW_XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / slabs_per_b, slab = blockIdx.x - b * slabs_per_b;
    const int cog = blockIdx.y / ci_groups, cig = blockIdx.y - cog * ci_groups;
    const int co0 = cog * 32, ci0 = cig * 32;
    const int cot = wave & 1, cit = wave >> 1;
    const int n = lane & 15, kk = lane >> 4;
    const int halo = (K / 2) * dil, xw = CG_TT + 2 * halo;
    const float* xb = x + (long)b * Cin * T;
    const float* dyb = dy + (long)b * Cout * T;
    f32x4 acc[K];
    #pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int tiles = (T + CG_TT - 1) / CG_TT;
    const int tile_end = min(tiles, (slab + 1) * slab_tiles);
    float dr[8][2], xr[8][3];
    auto fetch = [&](int tile) {
        const int t0 = tile * CG_TT;
        #pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave + 4 * rr;
            const int o = co0 + row, ci = ci0 + row;
            #pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int t = t0 + lane + 64 * c;
                dr[rr][c] = (o < Cout && t < T) ? dyb[(long)o * T + t] : 0.f;
            }
            #pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int col = lane + 64 * c, t = t0 - halo + col;
                xr[rr][c] = (col < xw && ci < Cin && t >= 0 && t < T) ? xb[(long)ci * T + t] : 0.f;
            }
        }
    };
    int tile = slab * slab_tiles;
    if (tile < tile_end) fetch(tile);
    for (; tile < tile_end; ++tile) {
        __syncthreads();
        #pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave + 4 * rr;
            #pragma unroll
            for (int c = 0; c < 2; ++c) dyl[row][lane + 64 * c] = dr[rr][c];
            #pragma unroll
            for (int c = 0; c < 3; ++c)
                if (lane + 64 * c < CW_XS) xl[row][lane + 64 * c] = xr[rr][c];
        }
        __syncthreads();
        if (tile + 1 < tile_end) fetch(tile + 1);
        const float* arow = &dyl[cot * 16 + n][kk];
        const float* brow = &xl[cit * 16 + n][kk];
        #pragma unroll 8
        for (int tt = 0; tt < CG_TT; tt += 4) {
            const float a = arow[tt];
            bsum += a;
            #pragma unroll
            for (int k = 0; k < K; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[tt + k * dil], acc[k], 0, 0, 0);
        }
    }
    // D[m = 4 (lane / 16) + e][n]: dW[co0 + 16 cot + m][ci0 + 16 cit + n][k]
    float* out = partial + (size_t)blockIdx.x * ((size_t)Cout * Cin * K + Cout);
    const int ci = ci0 + cit * 16 + n;
    #pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = co0 + cot * 16 + 4 * kk + e;
        if (o < Cout && ci < Cin) {
            #pragma unroll
            for (int k = 0; k < K; ++k) out[((long)o * Cin + ci) * K + k] = acc[k][e];
        }
    }
    if (cig == 0 && cit == 0) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        const int o = co0 + cot * 16 + n;
        if (kk == 0 && o < Cout) out[(size_t)Cout * Cin * K + o] = bsum;
    }
}

// dw / dbias = the slabs' partial results added in a fixed order (bit-reproducible): a workgroup owns 64 consecutive
// elements, its 16 waves take every 16th slab each (four loads in flight per lane: up to 800 slabs of the 24-channel
// layers were 80 us as one sequential chain per element), LDS adds the 16 partial sums in wave order
constexpr int CR_WAVES = 16;
__global__ __launch_bounds__(64 * CR_WAVES)
void conv1d_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ dbias,
                                int nw, int nb, int nslabs) {
    __shared__ float part[CR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const size_t pitch = (size_t)nw + nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < nw + nb) {
        int i = wave;
        for (; i + 3 * CR_WAVES < nslabs; i += 4 * CR_WAVES) {
            s0 += partial[(size_t)i * pitch + e];
            s1 += partial[(size_t)(i + CR_WAVES) * pitch + e];
            s2 += partial[(size_t)(i + 2 * CR_WAVES) * pitch + e];
            s3 += partial[(size_t)(i + 3 * CR_WAVES) * pitch + e];
        }
        for (; i < nslabs; i += CR_WAVES) s0 += partial[(size_t)i * pitch + e];
    }
    part[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < nw + nb) {
        float s = 0.f;
        #pragma unroll
        for (int w = 0; w < CR_WAVES; ++w) s += part[w][lane];
        if (e < nw) dw[e] = s;
        else if (dbias) dbias[e - nw] = s;
    }
}

bool cg_args_ok(int B, int Cin, int Cout, int T, int K, int dil) {
    return B >= 1 && Cin >= 1 && Cout >= 1 && T >= 1 && dil >= 1 && (K == 1 || K == 3);
}

struct WgradShape {
    int tiles, co_groups, ci_groups, slab_tiles, slabs_per_b, nslabs;
};

WgradShape wgrad_shape(int B, int Cin, int Cout, int T) {
    WgradShape s;
    s.tiles = (T + CG_TT - 1) / CG_TT;
    s.co_groups = (Cout + 31) / 32;
    s.ci_groups = (Cin + 31) / 32;
    // about a thousand workgroups (three per CU) where the layer has that much work
    const long want = ((long)B * s.tiles * s.co_groups * s.ci_groups + 1023) / 1024;
    s.slab_tiles = (int)(want < 1 ? 1 : (want > s.tiles ? s.tiles : want));
    s.slabs_per_b = (s.tiles + s.slab_tiles - 1) / s.slab_tiles;
    s.nslabs = B * s.slabs_per_b;
    return s;
}

template <int K, int CC, int NCT>
void launch_fwd3(hipStream_t stream, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int T,
                 int dil, int transposed, int groups) {
    const int tiles = (T + CG_TT - 1) / CG_TT;
    const dim3 grid(tiles, groups, B);
    hipLaunchKernelGGL((conv1d_fwd_kernel<K, CC, NCT>), grid, dim3(CG_THREADS), 0, stream, x, w, bias, y, Cin, Cout, T, dil, transposed, 1);
}

template <int K, int CC>
void launch_fwd2(hipStream_t stream, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int T,
                 int dil, int transposed) {
    // output channels per workgroup: the 16-channel tiles split evenly over the fewest groups of at most three (four
    // tiles per wave do not fit 168 registers next to the prefetched chunk: three workgroups per CU matter more)
    const int tiles16 = (Cout + 15) / 16, groups = (tiles16 + 2) / 3, nct = (tiles16 + groups - 1) / groups;
    switch (nct) {
        case 1: launch_fwd3<K, CC, 1>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed, groups); break;
        case 2: launch_fwd3<K, CC, 2>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed, groups); break;
        default: launch_fwd3<K, CC, 3>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed, groups); break;
    }
}

template <int K>
void launch_fwd(hipStream_t stream, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int T,
                int dil, int transposed) {
    // 24-channel inputs: one 24-row chunk (no zero rows) where the wave holds at most two channel tiles - with three the
    // prefetched chunk no longer fits 128 registers and three 8-row chunks do better
    const int tiles16 = (Cout + 15) / 16, groups = (tiles16 + 2) / 3, nct = (tiles16 + groups - 1) / groups;
    if (Cin % 16 != 0 && Cin % 24 == 0 && nct <= 2) launch_fwd2<K, 24>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed);
    else if (Cin <= 8 || Cin % 16 != 0) launch_fwd2<K, 8>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed);
    else launch_fwd2<K, 16>(stream, x, w, bias, y, B, Cin, Cout, T, dil, transposed);
}

}  // namespace

extern "C" {

int fastsvc_conv1d_forward(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t Cout,
                           int32_t T, int32_t K, int32_t dilation, int32_t transposed, void* stream_) {
    if (!x || !w || !y) return FASTSVC_E_INVALID;
    if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || dilation < 1 || K < 1 || (K & 1) == 0) return FASTSVC_E_INVALID;
    if (!cg_args_ok(B, Cin, Cout, T, K, dilation) || (K / 2) * dilation > CG_MAX_HALO || B > 65535) return FASTSVC_E_UNSUPPORTED;
    // (one utterance's input is addressed through a buffer descriptor with 32-bit byte offsets)
    if ((long)Cin * T * 4 >= 0x7fffff00L || (long)Cin * Cout * K * 4 >= 0x7fffff00L) return FASTSVC_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (K == 1) launch_fwd<1>(stream, x, w, bias, y, B, Cin, Cout, T, dilation, transposed);
    else launch_fwd<3>(stream, x, w, bias, y, B, Cin, Cout, T, dilation, transposed);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

size_t fastsvc_conv1d_backward_weight_scratch_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t T, int32_t K) {
    if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || K < 1) return 0;
    const WgradShape s = wgrad_shape(B, Cin, Cout, T);
    return (size_t)s.nslabs * ((size_t)Cout * Cin * K + Cout) * sizeof(float);
}

int fastsvc_conv1d_backward_weight(const float* x, const float* dy, float* dw, float* dbias, void* scratch, int32_t B, int32_t Cin,
                                   int32_t Cout, int32_t T, int32_t K, int32_t dilation, void* stream_) {
    if (!x || !dy || !dw || !scratch) return FASTSVC_E_INVALID;
    if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || dilation < 1 || K < 1 || (K & 1) == 0) return FASTSVC_E_INVALID;
    if (!cg_args_ok(B, Cin, Cout, T, K, dilation) || (K / 2) * dilation > CG_MAX_HALO) return FASTSVC_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const WgradShape s = wgrad_shape(B, Cin, Cout, T);
    float* partial = static_cast<float*>(scratch);
    const dim3 grid(s.nslabs, s.co_groups * s.ci_groups);
    if (K == 1)
        hipLaunchKernelGGL(conv1d_wgrad_kernel<1>, grid, dim3(CG_THREADS), 0, stream, x, dy, partial, (int)Cin, (int)Cout, (int)T,
                           (int)dilation, s.slab_tiles, s.slabs_per_b, s.ci_groups);
    else
        hipLaunchKernelGGL(conv1d_wgrad_kernel<3>, grid, dim3(CG_THREADS), 0, stream, x, dy, partial, (int)Cin, (int)Cout, (int)T,
                           (int)dilation, s.slab_tiles, s.slabs_per_b, s.ci_groups);
    const int nw = Cout * Cin * K, nb = Cout;
    hipLaunchKernelGGL(conv1d_wgrad_reduce_kernel, dim3((nw + nb + 63) / 64), dim3(64 * CR_WAVES), 0, stream,
                       partial, dw, dbias, nw, nb, s.nslabs);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
