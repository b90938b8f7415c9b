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

// ---- weight normalisation of ALL layers in one launch each way ----
// The reference parametrises every conv as w = g * v / ||v|| (torch.nn.utils.weight_norm, norm over all dims but 0:
// harana/models/fastsvc.py:354-362) and autograd differentiates it layer by layer: for the generator's 54 layers that is
// ~430 launches forward and as many backward per step (norm, div, mul and their backward nodes), each a few hundred
// values.  Here: one wave per output-channel row of any layer (a table of up to 56 layers travels in the kernel arguments),
//   forward   n = ||v_r||,  w_r = (g_r / n) v_r                      (n saved)
//   backward  dg_r = <dw_r, v_r> / n,   dv_r = (g_r / n) (dw_r - v_r <dw_r, v_r> / n^2)
namespace {

constexpr int WN_MAX_LAYERS = 56;       // 56 x 64 bytes of table: inside the 4 KB of kernel arguments

struct WnLayer {
    const float* v;      // (rows, cols)
    const float* g;      // (rows)
    float* w;            // forward: out (rows, cols); backward: dv
    float* aux;          // forward: norm out (rows); backward: dg
    const float* dw;     // backward only
    const float* norm;   // backward only
    int rows, cols, row0;
    int pad;
};

struct WnTable {
    WnLayer l[WN_MAX_LAYERS];
    int n, total_rows;
};

template <bool BACKWARD>
__global__ __launch_bounds__(256)
void weight_norm_kernel(WnTable tab) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= tab.total_rows) return;
    int li = 0;
    while (li + 1 < tab.n && tab.l[li + 1].row0 <= row) ++li;          // wave-uniform
    const WnLayer& L = tab.l[li];
    const int r = row - L.row0, cols = L.cols;
    const float* v = L.v + (long)r * cols;
    if (!BACKWARD) {
        float ss = 0.f;
        for (int c = lane; c < cols; c += 64) ss = fmaf(v[c], v[c], ss);
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float n = sqrtf(ss), sc = L.g[r] / n;
        float* w = L.w + (long)r * cols;
        for (int c = lane; c < cols; c += 64) w[c] = sc * v[c];
        if (lane == 0) L.aux[r] = n;
    } else {
        const float* dw = L.dw + (long)r * cols;
        float dot = 0.f;
        for (int c = lane; c < cols; c += 64) dot = fmaf(dw[c], v[c], dot);
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const float n = L.norm[r], sc = L.g[r] / n, k = dot / (n * n);
        float* dv = L.w + (long)r * cols;
        for (int c = lane; c < cols; c += 64) dv[c] = sc * (dw[c] - v[c] * k);
        if (lane == 0) L.aux[r] = dot / n;
    }
}

int wn_fill(WnTable& tab, int32_t n, const float* const* v, const float* const* g, float* const* w, float* const* aux,
            const float* const* dw, const float* const* norm, const int32_t* rows, const int32_t* cols) {
    if (n < 1 || !v || !g || !w || !aux || !rows || !cols) return FASTSVC_E_INVALID;
    if (n > WN_MAX_LAYERS) return FASTSVC_E_UNSUPPORTED;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!v[i] || !g[i] || !w[i] || !aux[i] || rows[i] < 1 || cols[i] < 1) return FASTSVC_E_INVALID;
        WnLayer& L = tab.l[i];
        L.v = v[i]; L.g = g[i]; L.w = w[i]; L.aux = aux[i];
        L.dw = dw ? dw[i] : nullptr; L.norm = norm ? norm[i] : nullptr;
        L.rows = rows[i]; L.cols = cols[i]; L.row0 = total; L.pad = 0;
        total += rows[i];
    }
    tab.n = n; tab.total_rows = total;
    return FASTSVC_OK;
}

}  // namespace

extern "C" {

int fastsvc_weight_norm_forward(int32_t n, const float* const* v, const float* const* g, float* const* w, float* const* norm,
                                const int32_t* rows, const int32_t* cols, void* stream_) {
    WnTable tab;
    const int rc = wn_fill(tab, n, v, g, w, norm, nullptr, nullptr, rows, cols);
    if (rc != FASTSVC_OK) return rc;
    hipLaunchKernelGGL(weight_norm_kernel<false>, dim3((tab.total_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), tab);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

int fastsvc_weight_norm_backward(int32_t n, const float* const* v, const float* const* g, const float* const* dw,
                                 const float* const* norm, float* const* dv, float* const* dg, const int32_t* rows,
                                 const int32_t* cols, void* stream_) {
    if (!dw || !norm) return FASTSVC_E_INVALID;
    WnTable tab;
    const int rc = wn_fill(tab, n, v, g, dv, dg, dw, norm, rows, cols);
    if (rc != FASTSVC_OK) return rc;
    for (int i = 0; i < n; ++i) if (!dw[i] || !norm[i]) return FASTSVC_E_INVALID;
    hipLaunchKernelGGL(weight_norm_kernel<true>, dim3((tab.total_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), tab);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
