// fastsvc_filmnorm.hip - the FiLM affine + InstanceNorm + speaker bias + LeakyReLU node of the generator's training
// graph on gfx950, forward and backward.
//
// The reference's up block applies `_feature_affine` three times (harana/models/fastsvc.py:96-110, 115-139):
//   u = scale * x + shift;  z = InstanceNorm2d(u) + emb_projector(normalize(spk_emb))   (eps 1e-5, no affine, biased
//   variance over the time axis of every (utterance, channel) row)
// and every one of them is followed by the LeakyReLU(0.2) that opens the next conv block (fastsvc.py:60-83).  In
// PyTorch that is six elementwise / reduction launches forward and a dozen backward per application, each a full pass
// over (B, C, T).  Here one workgroup owns one (utterance, channel) row:
//   forward   mean, then the centred sum of squares (two passes over the row: no E[u^2] - mean^2 cancellation),
//             then out = lrelu((u - mean) rstd + bias); the row is read from L2 the second and third time
//   backward  dz = dout * lrelu'(z); the two row sums (sum dz, sum dz u^) in one pass, then
//             du = rstd (dz - mean(dz) - u^ mean(dz u^)),  dx = du scale, dscale = du x, dshift = du, dbias = sum dz
// with u^, z recomputed from x / scale / shift and the saved (mean, rstd): nothing of size (B, C, T) is kept between
// the passes but the node's inputs.  Sums: float per thread (at most T / 256 terms), double across the workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

constexpr int FN_THREADS = 256;

__device__ __forceinline__ double fn_block_sum(double v, double* red) {
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(FN_THREADS)
void film_norm_forward_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                              const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ mean_out,
                              float* __restrict__ rstd_out, int T, float eps, float slope) {
    __shared__ double red[4];
    const long row = blockIdx.x;
    x += row * T; sc += row * T; sh += row * T; out += row * T;
    float s = 0.f;
    for (int i = threadIdx.x; i < T; i += FN_THREADS) s += fmaf(sc[i], x[i], sh[i]);
    const float mean = (float)(fn_block_sum((double)s, red) / T);
    float q = 0.f;
    for (int i = threadIdx.x; i < T; i += FN_THREADS) {
        const float d = fmaf(sc[i], x[i], sh[i]) - mean;
        q = fmaf(d, d, q);
    }
    const float rstd = (float)(1.0 / sqrt(fn_block_sum((double)q, red) / T + (double)eps));
    const float bv = bias[row];
    for (int i = threadIdx.x; i < T; i += FN_THREADS) {
        const float z = (fmaf(sc[i], x[i], sh[i]) - mean) * rstd + bv;
        out[i] = z > 0.f ? z : slope * z;
    }
    if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

__global__ __launch_bounds__(FN_THREADS)
void film_norm_backward_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ sc,
                               const float* __restrict__ sh, const float* __restrict__ bias, const float* __restrict__ mean_in,
                               const float* __restrict__ rstd_in, float* __restrict__ dx, float* __restrict__ dsc,
                               float* __restrict__ dsh, float* __restrict__ dbias, int T, float slope) {
    __shared__ double red[4];
    const long row = blockIdx.x;
    dout += row * T; x += row * T; sc += row * T; sh += row * T; dx += row * T; dsc += row * T; dsh += row * T;
    const float mean = mean_in[row], rstd = rstd_in[row], bv = bias[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < T; i += FN_THREADS) {
        const float uh = (fmaf(sc[i], x[i], sh[i]) - mean) * rstd;
        const float dz = (uh + bv > 0.f) ? dout[i] : slope * dout[i];
        s1 += dz;
        s2 = fmaf(dz, uh, s2);
    }
    const double t1 = fn_block_sum((double)s1, red), t2 = fn_block_sum((double)s2, red);
    const float m1 = (float)(t1 / T), m2 = (float)(t2 / T);
    for (int i = threadIdx.x; i < T; i += FN_THREADS) {
        const float xv = x[i], sv = sc[i];
        const float uh = (fmaf(sv, xv, sh[i]) - mean) * rstd;
        const float dz = (uh + bv > 0.f) ? dout[i] : slope * dout[i];
        const float du = rstd * (dz - m1 - uh * m2);
        dx[i] = du * sv;
        dsc[i] = du * xv;
        dsh[i] = du;
    }
    if (threadIdx.x == 0) dbias[row] = (float)t1;
}

}  // namespace

extern "C" {

int fastsvc_film_norm_forward(const float* x, const float* scale, const float* shift, const float* bias, float* out, float* mean,
                              float* rstd, int32_t rows, int32_t T, float eps, float slope, void* stream_) {
    if (!x || !scale || !shift || !bias || !out || !mean || !rstd || rows < 1 || T < 1) return FASTSVC_E_INVALID;
    hipLaunchKernelGGL(film_norm_forward_kernel, dim3(rows), dim3(FN_THREADS), 0, static_cast<hipStream_t>(stream_), x, scale, shift,
                       bias, out, mean, rstd, (int)T, eps, slope);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

int fastsvc_film_norm_backward(const float* dout, const float* x, const float* scale, const float* shift, const float* bias,
                               const float* mean, const float* rstd, float* dx, float* dscale, float* dshift, float* dbias,
                               int32_t rows, int32_t T, float slope, void* stream_) {
    if (!dout || !x || !scale || !shift || !bias || !mean || !rstd || !dx || !dscale || !dshift || !dbias || rows < 1 || T < 1)
        return FASTSVC_E_INVALID;
    hipLaunchKernelGGL(film_norm_backward_kernel, dim3(rows), dim3(FN_THREADS), 0, static_cast<hipStream_t>(stream_), dout, x, scale,
                       shift, bias, mean, rstd, dx, dscale, dshift, dbias, (int)T, slope);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
