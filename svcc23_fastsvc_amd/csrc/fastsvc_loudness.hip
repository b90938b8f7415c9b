// fastsvc_loudness.hip - A-weighted log loudness feature on gfx950.
//
// Replaces `loudness_extract(audio, sampling_rate, hop_length)` of harana/bin/preprocess_fastsvc.py:60-75,
// the producer of the generator's `l` input.  The reference composes librosa 0.8.1 (setup.py:30) calls:
//   stft = librosa.stft(audio, hop_length=hop)                   n_fft 2048, periodic Hann, center=True, reflect pad
//   P    = |stft|^2
//   dB   = perceptual_weighting(P, fft_frequencies(sr))          power_to_db(P, ref 1, amin 1e-10, top_db 80: floor
//                                                                 at (max over the WHOLE spectrogram) - 80 dB) + A-weighting
//   amp  = db_to_amplitude(dB) = 10^(dB/20);   loudness[f] = log(mean_bins(amp) + 1e-5)
//   Stretch2d(hop, 1): every frame value repeated hop times      -> (1 + T // hop) * hop samples
// Here: one workgroup per frame runs a 2048-point radix-2 FFT in LDS (f32; bins below max - 80 dB are clamped
// anyway), writes the 1025-bin power spectrum and folds the utterance maximum with an atomic; a second
// kernel applies floor / A-weighting / mean / log and writes the stretched rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

constexpr int NFFT = 2048;
constexpr int NBINS = NFFT / 2 + 1;
constexpr int LOGN = 11;

__device__ __forceinline__ int reflect_index(int i, int T) {
    // np.pad(..., mode="reflect"): ... y[2] y[1] | y[0] y[1] ... y[T-1] | y[T-2] y[T-3] ...
    if (T == 1) return 0;
    const int period = 2 * (T - 1);
    i %= period;
    if (i < 0) i += period;
    return i < T ? i : period - i;
}

__global__ __launch_bounds__(256)
void loudness_power_kernel(const float* __restrict__ audio, float* __restrict__ power, unsigned* __restrict__ pmax_bits,
                           int T, int hop, int frames) {
    __shared__ float re[NFFT], im[NFFT];
    __shared__ float wmax[4];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* y = audio + (long)b * T;
    // windowed frame, written in bit-reversed order (decimation in time)
    for (int k = tid; k < NFFT; k += 256) {
        const int src = reflect_index(f * hop - NFFT / 2 + k, T);
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)NFFT);        // scipy get_window("hann", fftbins=True)
        const int r = (int)(__brev((unsigned)k) >> (32 - LOGN));
        re[r] = y[src] * w;
        im[r] = 0.f;
    }
    __syncthreads();
    for (int s = 1; s <= LOGN; ++s) {
        const int half = 1 << (s - 1);
        for (int i = tid; i < NFFT / 2; i += 256) {
            const int j = i & (half - 1);
            const int lo = ((i >> (s - 1)) << s) + j, hi = lo + half;
            float sn, cs;
            sincospif(-(float)j / (float)half, &sn, &cs);                           // exp(-i pi j / half)
            const float xr = re[hi] * cs - im[hi] * sn, xi = re[hi] * sn + im[hi] * cs;
            const float ar = re[lo], ai = im[lo];
            re[lo] = ar + xr; im[lo] = ai + xi;
            re[hi] = ar - xr; im[hi] = ai - xi;
        }
        __syncthreads();
    }
    float m = 0.f;
    float* prow = power + ((long)b * frames + f) * NBINS;
    for (int k = tid; k < NBINS; k += 256) {
        const float p = re[k] * re[k] + im[k] * im[k];
        prow[k] = p;
        m = fmaxf(m, p);
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) atomicMax(&pmax_bits[b], __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

__device__ __forceinline__ float a_weighting_db(float freq) {
    // librosa.A_weighting (min_db = -80): IEC 61672 A curve
    const float f2 = freq * freq;
    const float c0 = 12200.f * 12200.f, c1 = 20.6f * 20.6f, c2 = 107.7f * 107.7f, c3 = 737.9f * 737.9f;
    if (f2 <= 0.f) return -80.f;
    const float w = 2.0f + 20.0f * (log10f(c0) + 2.0f * log10f(f2) - log10f(f2 + c0) - log10f(f2 + c1)
                                    - 0.5f * log10f(f2 + c2) - 0.5f * log10f(f2 + c3));
    return fmaxf(w, -80.f);
}

__global__ __launch_bounds__(256)
void loudness_reduce_kernel(const float* __restrict__ power, const unsigned* __restrict__ pmax_bits,
                            float* __restrict__ out, int hop, int frames, float sample_rate) {
    __shared__ float part[4];
    __shared__ float result;
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* prow = power + ((long)b * frames + f) * NBINS;
    const float floor_db = 10.f * log10f(fmaxf(1e-10f, __uint_as_float(pmax_bits[b]))) - 80.f;    // top_db = 80
    float sum = 0.f;
    for (int k = tid; k < NBINS; k += 256) {
        float db = fmaxf(10.f * log10f(fmaxf(1e-10f, prow[k])), floor_db);
        db += a_weighting_db((float)k * sample_rate / (float)NFFT);
        sum += exp10f(db * 0.05f);
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) part[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) result = logf(((part[0] + part[1]) + (part[2] + part[3])) / (float)NBINS + 1e-5f);
    __syncthreads();
    float* orow = out + ((long)b * frames + f) * hop;
    for (int j = tid; j < hop; j += 256) orow[j] = result;
}

}  // namespace

extern "C" {

int32_t fastsvc_loudness_frames(int32_t T, int32_t hop) { return (T < 1 || hop < 1) ? 0 : 1 + T / hop; }

size_t fastsvc_loudness_scratch_bytes(int32_t B, int32_t T, int32_t hop) {
    if (B < 1 || T < 1 || hop < 1) return 0;
    return (size_t)B * fastsvc_loudness_frames(T, hop) * NBINS * sizeof(float) + 256 * (size_t)((B + 63) / 64);
}

int fastsvc_loudness_extract(const float* audio, float* out, void* scratch, int32_t B, int32_t T, int32_t hop,
                             float sample_rate, void* stream_) {
    if (!audio || !out || !scratch || B < 1 || T < 2 || hop < 1) return FASTSVC_E_INVALID;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int frames = fastsvc_loudness_frames(T, hop);
    float* power = static_cast<float*>(scratch);
    unsigned* pmax = reinterpret_cast<unsigned*>(power + (size_t)B * frames * NBINS);
    if (hipMemsetAsync(pmax, 0, sizeof(unsigned) * B, stream) != hipSuccess) return FASTSVC_E_HIP;
    hipLaunchKernelGGL(loudness_power_kernel, dim3(frames, B), dim3(256), 0, stream, audio, power, pmax, T, hop, frames);
    hipLaunchKernelGGL(loudness_reduce_kernel, dim3(frames, B), dim3(256), 0, stream, power, pmax, out, hop, frames, sample_rate);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
