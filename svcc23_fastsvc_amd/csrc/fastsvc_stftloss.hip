// fastsvc_stftloss.hip - multi-resolution STFT loss of the training step on gfx950, forward and backward
// (SURVEY.md 8 f2).
//
// Replaces MultiResolutionSTFTLoss.forward(x, y) of harana/losses/stft_loss.py:131-180 and what autograd derives from
// it for the predicted signal x:
//   per resolution r (fft N, hop, window of win_length samples centred in the N-sample frame, stft_loss.py:21-51):
//     X = |STFT(x)|, Y = |STFT(y)| with |.| = sqrt(clamp(re^2 + im^2, 1e-7)), centred frames, reflect padding,
//     one-sided (N / 2 + 1 bins), frames = 1 + T / hop
//     sc_r  = ||Y - X||_F / ||Y||_F                        (stft_loss.py:54-74, norms over the whole batch)
//     mag_r = mean |log Y - log X|                         (stft_loss.py:77-97)
//   sc = mean_r sc_r, mag = mean_r mag_r                   (stft_loss.py:170-180)
//
// Work decomposition (the whole loss is a few hundred thousand short FFTs: latency- and launch-bound in the stock
// composition - 6 x (2 stft, clamp, sqrt, 2 norms, 2 logs, abs, mean) and as many backward nodes):
//   * one launch per resolution; a workgroup holds 4096 complex values in LDS = 4096 / N "slots" of one N-point
//     complex FFT each.  A slot transforms TWO consecutive frames of ONE signal at once (frame 2p in the real part,
//     frame 2p + 1 in the imaginary part; the two spectra are separated with the Hermitian symmetry) - two frames of
//     the same signal, not x with y, so that a quiet prediction's bins are not rounded at the target's magnitude.
//     Slots alternate x / y of the same frame pair: a bin's X and Y meet in one workgroup.
//   * radix-2 decimation-in-frequency in place (natural order in, bit-reversed out), twiddles from an LDS table
//     (sincospi, computed once per workgroup).  The sums S1 = sum (Y - X)^2, S2 = sum Y^2, S3 = sum |log Y - log X|
//     leave a workgroup as three doubles; one small launch folds them in a FIXED order (bit-reproducible losses).
//   * backward = the same transform again (nothing but the three sums per resolution is kept from the forward), the
//     gradient of both outputs with respect to every bin, and the adjoint transform: for a frame a[n] = w[n] x~[n],
//     d/da[n] = Re sum_{k <= N/2} G[k] e^{+2 pi i k n / N} with G = dL/dRe + i dL/dIm.  The Hermitian extension of G
//     makes that sum a REAL inverse DFT, so the two frames of a slot again share one complex FFT (decimation in
//     time: bit-reversed in - exactly where the forward pass left the bins - natural order out).  The windowed frame
//     gradients go to a scratch buffer and a gather launch adds, for every sample, the (at most N / hop) frames and the
//     reflected positions that cover it: no atomics, gradients bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

constexpr int SL_THREADS = 256;
constexpr int SL_ELEMS = 4096;           // complex values per workgroup
constexpr int SL_MAX_N = 2048;
constexpr int SL_MIN_N = 8;
constexpr int SL_MAX_RES = 16;
constexpr float SL_FLOOR = 1e-7f;        // clamp(re^2 + im^2, min=1e-7)  (stft_loss.py:49-51)

struct SlRes {
    const float* win;      // device, win_length values
    long frames_off;       // float offset of this resolution's (B, frames, N) frame gradients in the scratch
    double nel;            // B * frames * (N / 2 + 1)
    int N, logn, hop, wl, woff;
    int frames, pairs;     // frames = 1 + T / hop, pairs = ceil(frames / 2)
    int ppb, bpu;          // frame pairs per workgroup, workgroups per utterance
    int part_off;          // index of this resolution's first workgroup in the partial sums
    int nblocks;           // B * bpu
};

struct SlAll {
    SlRes r[SL_MAX_RES];
    int n;
};

struct SlLayout {
    size_t stats_off, part_off, frames_off, bytes;
    int total_blocks;
};

__device__ __forceinline__ int sl_reflect(int q, int T) {
    // F.pad(..., mode="reflect") of torch.stft(center=True): x[2] x[1] | x[0] ... x[T-1] | x[T-2] x[T-3]   (N / 2 < T)
    return q < 0 ? -q : (q >= T ? 2 * (T - 1) - q : q);
}

__device__ __forceinline__ float2 sl_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// exp(-2 pi i m / N) for m < N / 2
__device__ __forceinline__ void sl_twiddles(float2* tw, int N) {
    for (int m = threadIdx.x; m < N / 2; m += SL_THREADS) {
        float sn, cs;
        sincospif(-2.0f * (float)m / (float)N, &sn, &cs);
        tw[m] = make_float2(cs, sn);
    }
}

// The windowed frame pairs of this workgroup, natural order: slot 2p = x, slot 2p + 1 = y of pair pair0 + p
__device__ __forceinline__ void sl_load(float2* z, const float* __restrict__ x, const float* __restrict__ y, const SlRes& r,
                                        int T, int pair0, int nslots) {
    const int N = r.N;
    for (int idx = threadIdx.x; idx < nslots * N; idx += SL_THREADS) {
        const int slot = idx >> r.logn, n = idx & (N - 1);
        const float* s = (slot & 1) ? y : x;
        const int f0 = 2 * (pair0 + (slot >> 1));
        const int wi = n - r.woff;
        const float w = (wi >= 0 && wi < r.wl) ? r.win[wi] : 0.f;
        const int q = f0 * r.hop + n - N / 2;
        float2 v = make_float2(0.f, 0.f);
        if (f0 < r.frames) v.x = s[sl_reflect(q, T)] * w;
        if (f0 + 1 < r.frames) v.y = s[sl_reflect(q + r.hop, T)] * w;
        z[idx] = v;
    }
}

// Decimation in frequency, in place: natural order in, bin k at position brev(k) out.  `stride` selects the slots
// (1: all, 2: the x slots only).
__device__ __forceinline__ void sl_fft_dif(float2* z, const float2* tw, int logn, int nslots, int stride) {
    const int N = 1 << logn;
    for (int s = logn; s >= 1; --s) {
        const int half = 1 << (s - 1);
        for (int i = threadIdx.x; i < nslots * (N / 2); i += SL_THREADS) {
            const int slot = (i >> (logn - 1)) * stride, ii = i & (N / 2 - 1);
            const int j = ii & (half - 1);
            const int lo = slot * N + ((ii >> (s - 1)) << s) + j, hi = lo + half;
            const float2 a = z[lo], b = z[hi];
            z[lo] = make_float2(a.x + b.x, a.y + b.y);
            z[hi] = sl_cmul(make_float2(a.x - b.x, a.y - b.y), tw[j << (logn - s)]);
        }
        __syncthreads();
    }
}

// Decimation in time, in place: value k at position brev(k) in, natural order out
__device__ __forceinline__ void sl_fft_dit(float2* z, const float2* tw, int logn, int nslots, int stride) {
    const int N = 1 << logn;
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        for (int i = threadIdx.x; i < nslots * (N / 2); i += SL_THREADS) {
            const int slot = (i >> (logn - 1)) * stride, ii = i & (N / 2 - 1);
            const int j = ii & (half - 1);
            const int lo = slot * N + ((ii >> (s - 1)) << s) + j, hi = lo + half;
            const float2 a = z[lo], b = sl_cmul(z[hi], tw[j << (logn - s)]);
            z[lo] = make_float2(a.x + b.x, a.y + b.y);
            z[hi] = make_float2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int sl_brev(int k, int logn) { return (int)(__brev((unsigned)k) >> (32 - logn)); }

// Spectra of the two real frames packed into one complex transform: Z = FFT(a + i b),
// A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / (2 i)
__device__ __forceinline__ void sl_split(float2 zk, float2 zc, float2& A, float2& Bv) {
    A = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
    Bv = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
}

__device__ __forceinline__ double sl_block_sum(double v, double* red) {
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(SL_THREADS)
void stft_loss_forward_kernel(const float* __restrict__ x, const float* __restrict__ y, int T, SlRes r,
                              double* __restrict__ partials) {
    __shared__ float2 z[SL_ELEMS];
    __shared__ float2 tw[SL_MAX_N / 2];
    __shared__ double red[4];
    const int N = r.N;
    const int b = blockIdx.x / r.bpu, pair0 = (blockIdx.x % r.bpu) * r.ppb;
    const int np = min(r.ppb, r.pairs - pair0);
    x += (long)b * T; y += (long)b * T;
    sl_twiddles(tw, N);
    sl_load(z, x, y, r, T, pair0, 2 * np);
    __syncthreads();
    sl_fft_dif(z, tw, r.logn, 2 * np, 1);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int nb = N / 2 + 1;
    for (int idx = threadIdx.x; idx < np * nb; idx += SL_THREADS) {
        const int p = idx / nb, k = idx - p * nb;
        const int pk = sl_brev(k, r.logn), pc = sl_brev((N - k) & (N - 1), r.logn);
        const float2* zx = z + (2 * p) * N;
        const float2* zy = zx + N;
        float2 X[2], Y[2];
        sl_split(zx[pk], zx[pc], X[0], X[1]);
        sl_split(zy[pk], zy[pc], Y[0], Y[1]);
        const int f0 = 2 * (pair0 + p);
        #pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (f0 + i >= r.frames) continue;
            const float xm = sqrtf(fmaxf(X[i].x * X[i].x + X[i].y * X[i].y, SL_FLOOR));
            const float ym = sqrtf(fmaxf(Y[i].x * Y[i].x + Y[i].y * Y[i].y, SL_FLOOR));
            const float d = ym - xm;
            s1 += d * d;
            s2 += ym * ym;
            s3 += fabsf(logf(ym) - logf(xm));
        }
    }
    const double t1 = sl_block_sum((double)s1, red), t2 = sl_block_sum((double)s2, red), t3 = sl_block_sum((double)s3, red);
    if (threadIdx.x == 0) {
        double* o = partials + (size_t)(r.part_off + blockIdx.x) * 3;
        o[0] = t1; o[1] = t2; o[2] = t3;
    }
}

// One workgroup: stats[r] = (S1, S2, S3) folded in a fixed order; loss = (sc, mag)
__global__ __launch_bounds__(SL_THREADS)
void stft_loss_finalize_kernel(SlAll all, const double* __restrict__ partials, double* __restrict__ stats, float* __restrict__ loss) {
    __shared__ double red[4];
    double sc = 0.0, mag = 0.0;
    for (int i = 0; i < all.n; ++i) {
        const SlRes& r = all.r[i];
        double a[3] = {0.0, 0.0, 0.0};
        for (int j = threadIdx.x; j < r.nblocks; j += SL_THREADS) {
            const double* p = partials + (size_t)(r.part_off + j) * 3;
            a[0] += p[0]; a[1] += p[1]; a[2] += p[2];
        }
        const double S1 = sl_block_sum(a[0], red), S2 = sl_block_sum(a[1], red), S3 = sl_block_sum(a[2], red);
        if (threadIdx.x == 0) { stats[4 * i] = S1; stats[4 * i + 1] = S2; stats[4 * i + 2] = S3; }
        sc += sqrt(S1) / sqrt(S2);
        mag += S3 / r.nel;
    }
    if (threadIdx.x == 0) { loss[0] = (float)(sc / all.n); loss[1] = (float)(mag / all.n); }
}

__global__ __launch_bounds__(SL_THREADS)
void stft_loss_backward_frames_kernel(const float* __restrict__ x, const float* __restrict__ y, int T, SlRes r, int nres,
                                      const double* __restrict__ stats /* of this resolution */,
                                      const float* __restrict__ grad_loss, float* __restrict__ frames_grad) {
    __shared__ float2 z[SL_ELEMS];
    __shared__ float2 tw[SL_MAX_N / 2];
    const int N = r.N;
    const int b = blockIdx.x / r.bpu, pair0 = (blockIdx.x % r.bpu) * r.ppb;
    const int np = min(r.ppb, r.pairs - pair0);
    x += (long)b * T; y += (long)b * T;
    sl_twiddles(tw, N);
    sl_load(z, x, y, r, T, pair0, 2 * np);
    __syncthreads();
    sl_fft_dif(z, tw, r.logn, 2 * np, 1);
    // d sc / d X = (X - Y) / (sqrt(S1) sqrt(S2)) (0 where the spectra coincide: the subgradient torch.norm takes);
    // d mag / d X = sign(log X - log Y) / (X nel); both times the incoming gradient over the number of resolutions
    const double S1 = stats[0], S2 = stats[1];
    const float c_sc = S1 > 0.0 ? (float)((double)grad_loss[0] / ((double)nres * sqrt(S1) * sqrt(S2))) : 0.f;
    const float c_mag = (float)((double)grad_loss[1] / ((double)nres * r.nel));
    const int nb = N / 2 + 1;
    for (int idx = threadIdx.x; idx < np * nb; idx += SL_THREADS) {
        const int p = idx / nb, k = idx - p * nb;
        const int pk = sl_brev(k, r.logn), pc = sl_brev((N - k) & (N - 1), r.logn);
        float2* zx = z + (2 * p) * N;
        const float2* zy = zx + N;
        float2 X[2], Y[2], G[2];
        sl_split(zx[pk], zx[pc], X[0], X[1]);
        sl_split(zy[pk], zy[pc], Y[0], Y[1]);
        const int f0 = 2 * (pair0 + p);
        #pragma unroll
        for (int i = 0; i < 2; ++i) {
            G[i] = make_float2(0.f, 0.f);
            const float pw = X[i].x * X[i].x + X[i].y * X[i].y;
            if (f0 + i >= r.frames || pw < SL_FLOOR) continue;             // clamp: no gradient below the floor
            const float xm = sqrtf(pw);
            const float ym = sqrtf(fmaxf(Y[i].x * Y[i].x + Y[i].y * Y[i].y, SL_FLOOR));
            const float dl = logf(xm) - logf(ym);
            const float sg = dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f);
            const float g = (c_sc * (xm - ym) + c_mag * sg / xm) / xm;     // dL/dX / X: times (re, im)
            G[i] = make_float2(g * X[i].x, g * X[i].y);
        }
        // conj of W = H0 + i H1, H the Hermitian extension of G (see the header), at the bins' own positions
        if (k == 0 || k == N / 2) {
            zx[pk] = make_float2(G[0].x, -G[1].x);
        } else {
            zx[pk] = make_float2(0.5f * (G[0].x - G[1].y), 0.5f * (-G[0].y - G[1].x));
            zx[pc] = make_float2(0.5f * (G[0].x + G[1].y), 0.5f * (G[0].y - G[1].x));
        }
    }
    __syncthreads();
    sl_fft_dit(z, tw, r.logn, np, 2);
    // DFT(conj W) = ga0 - i ga1: the frame gradients, times the window
    float* out = frames_grad + r.frames_off + (long)b * r.frames * N;
    for (int idx = threadIdx.x; idx < np * N; idx += SL_THREADS) {
        const int p = idx >> r.logn, n = idx & (N - 1);
        const float2 v = z[(2 * p) * N + n];
        const int wi = n - r.woff;
        const float w = (wi >= 0 && wi < r.wl) ? r.win[wi] : 0.f;
        const int f0 = 2 * (pair0 + p);
        if (f0 < r.frames) out[(long)f0 * N + n] = w * v.x;
        if (f0 + 1 < r.frames) out[(long)(f0 + 1) * N + n] = -w * v.y;
    }
}

// grad_x[b][t] = sum over resolutions, over the padded positions q that read sample t (itself and its reflections)
// and over the frames that cover q
__global__ __launch_bounds__(SL_THREADS)
void stft_loss_backward_gather_kernel(SlAll all, const float* __restrict__ frames_grad, float* __restrict__ grad_x, int T) {
    const int t = blockIdx.x * SL_THREADS + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    float acc = 0.f;
    for (int i = 0; i < all.n; ++i) {
        const SlRes& r = all.r[i];
        const int N = r.N, pad = N / 2;
        const float* fg = frames_grad + r.frames_off + (long)b * r.frames * N;
        int qs[3];
        int nq = 0;
        qs[nq++] = t + pad;
        if (t >= 1 && t <= pad) qs[nq++] = pad - t;
        if (t <= T - 2 && t >= T - 1 - pad) qs[nq++] = pad + 2 * (T - 1) - t;
        float a = 0.f;
        for (int j = 0; j < nq; ++j) {
            const int q = qs[j];
            const int f_hi = min(r.frames - 1, q / r.hop);
            int f_lo = q - N + 1;
            f_lo = f_lo <= 0 ? 0 : (f_lo + r.hop - 1) / r.hop;
            for (int f = f_lo; f <= f_hi; ++f) a += fg[(long)f * N + (q - f * r.hop)];
        }
        acc += a;
    }
    grad_x[(long)b * T + t] = acc;
}

int sl_describe(SlAll& all, SlLayout& lay, int B, int T, int n_res, const int32_t* fft_sizes, const int32_t* hop_sizes,
                const int32_t* win_lengths, const float* const* windows) {
    if (B < 1 || T < 2 || n_res < 1 || !fft_sizes || !hop_sizes) return FASTSVC_E_INVALID;
    if (n_res > SL_MAX_RES) return FASTSVC_E_UNSUPPORTED;
    all.n = n_res;
    long frames_off = 0;
    int part = 0;
    for (int i = 0; i < n_res; ++i) {
        SlRes& r = all.r[i];
        const int N = fft_sizes[i];
        const int wl = win_lengths ? win_lengths[i] : N;
        if (N < 1 || hop_sizes[i] < 1 || wl < 1 || wl > N) return FASTSVC_E_INVALID;
        if (N / 2 >= T) return FASTSVC_E_INVALID;                    // reflect padding needs N / 2 < T (torch raises too)
        if ((N & (N - 1)) != 0 || N < SL_MIN_N || N > SL_MAX_N) return FASTSVC_E_UNSUPPORTED;
        r.N = N;
        r.logn = 0;
        while ((1 << r.logn) < N) ++r.logn;
        r.hop = hop_sizes[i];
        r.wl = wl;
        r.woff = (N - wl) / 2;                                       // torch.stft centres a short window in the frame
        r.win = windows ? windows[i] : nullptr;
        r.frames = 1 + T / r.hop;
        r.pairs = (r.frames + 1) / 2;
        r.ppb = SL_ELEMS / (2 * N);
        r.bpu = (r.pairs + r.ppb - 1) / r.ppb;
        r.nblocks = B * r.bpu;
        r.part_off = part;
        part += r.nblocks;
        r.frames_off = frames_off;
        frames_off += (long)B * r.frames * N;
        r.nel = (double)B * r.frames * (N / 2 + 1);
    }
    lay.total_blocks = part;
    lay.stats_off = 0;
    lay.part_off = 4 * SL_MAX_RES * sizeof(double);
    lay.frames_off = lay.part_off + (size_t)part * 3 * sizeof(double);
    lay.frames_off = (lay.frames_off + 255) & ~(size_t)255;
    lay.bytes = lay.frames_off + (size_t)frames_off * sizeof(float);
    return FASTSVC_OK;
}

}  // namespace

extern "C" {

size_t fastsvc_stft_loss_scratch_bytes(int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes, const int32_t* hop_sizes) {
    SlAll all;
    SlLayout lay;
    if (sl_describe(all, lay, B, T, n_res, fft_sizes, hop_sizes, nullptr, nullptr) != FASTSVC_OK) return 0;
    return lay.bytes;
}

int fastsvc_stft_loss_forward(const float* x, const float* y, int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes,
                              const int32_t* hop_sizes, const int32_t* win_lengths, const float* const* windows,
                              float* loss, void* scratch, void* stream_) {
    if (!x || !y || !loss || !scratch || !windows || !win_lengths) return FASTSVC_E_INVALID;
    SlAll all;
    SlLayout lay;
    const int rc = sl_describe(all, lay, B, T, n_res, fft_sizes, hop_sizes, win_lengths, windows);
    if (rc != FASTSVC_OK) return rc;
    for (int i = 0; i < n_res; ++i) if (!windows[i]) return FASTSVC_E_INVALID;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(scratch);
    double* stats = reinterpret_cast<double*>(base + lay.stats_off);
    double* partials = reinterpret_cast<double*>(base + lay.part_off);
    for (int i = 0; i < n_res; ++i)
        hipLaunchKernelGGL(stft_loss_forward_kernel, dim3(all.r[i].nblocks), dim3(SL_THREADS), 0, stream, x, y, (int)T, all.r[i], partials);
    hipLaunchKernelGGL(stft_loss_finalize_kernel, dim3(1), dim3(SL_THREADS), 0, stream, all, partials, stats, loss);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

int fastsvc_stft_loss_backward(const float* x, const float* y, int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes,
                               const int32_t* hop_sizes, const int32_t* win_lengths, const float* const* windows,
                               const float* grad_loss, float* grad_x, void* scratch, void* stream_) {
    if (!x || !y || !grad_loss || !grad_x || !scratch || !windows || !win_lengths) return FASTSVC_E_INVALID;
    SlAll all;
    SlLayout lay;
    const int rc = sl_describe(all, lay, B, T, n_res, fft_sizes, hop_sizes, win_lengths, windows);
    if (rc != FASTSVC_OK) return rc;
    for (int i = 0; i < n_res; ++i) if (!windows[i]) return FASTSVC_E_INVALID;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(scratch);
    const double* stats = reinterpret_cast<const double*>(base + lay.stats_off);
    float* frames_grad = reinterpret_cast<float*>(base + lay.frames_off);
    for (int i = 0; i < n_res; ++i)
        hipLaunchKernelGGL(stft_loss_backward_frames_kernel, dim3(all.r[i].nblocks), dim3(SL_THREADS), 0, stream, x, y, (int)T,
                           all.r[i], (int)n_res, stats + 4 * i, grad_loss, frames_grad);
    hipLaunchKernelGGL(stft_loss_backward_gather_kernel, dim3((T + SL_THREADS - 1) / SL_THREADS, B), dim3(SL_THREADS), 0, stream,
                       all, frames_grad, grad_x, (int)T);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
