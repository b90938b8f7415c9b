// fastsvc_signal.hip - sine-excitation synthesis on gfx950.
//
// Replaces harana/utils/features.py:111-213 `SignalGenerator.__call__` (the step right before the
// generator forward inside `FastSVCGenerator.inference`, fastsvc.py:381):
//   sinusoid(f0):  vuv = (f0 > 0) nearest-upsampled x hop;  rad = (f0_up / sr) % 1;
//                  sine = sine_amp * vuv * sin(2 pi cumsum(rad)) + randn * (noise_amp voiced, noise_amp/3 unvoiced)
//   random_noise:  randn;    vuv_binary:  vuv
// rad is constant inside a frame, so cumsum[f*hop + j] = C_f + (j+1) * rad_f with the frame prefix
// C_f = hop * sum_{g<f} rad_g: a scan over F frames (f64, reduced mod 1 so the sin argument stays
// small - the reference's fp32 cumsum reaches ~2e3 cycles at 10 s) and a closed form per sample.
// Noise: counter-based hash + Box-Muller (the reference's torch.randn stream is not reproducible
// anyway); deterministic for a given seed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

constexpr double TWO_PI = 6.283185307179586476925286766559;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float gauss(uint64_t seed, uint64_t idx) {
    const uint64_t h = mix64(mix64(idx ^ seed) + seed);
    const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// one workgroup per utterance: exclusive scan of hop * rad_f over the frames, kept mod 1 in f64
__global__ __launch_bounds__(256)
void frame_phase_kernel(const float* __restrict__ f0, double* __restrict__ cf, int F, int hop, float sr) {
    __shared__ double part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* f = f0 + (long)b * F;
    double* c = cf + (long)b * F;
    const int per = (F + 255) / 256;
    const int lo = tid * per, hi = min(F, lo + per);
    double s = 0.0;
    for (int i = lo; i < hi; ++i) {
        const float rad = fmodf(f[i] / sr, 1.0f);              // (f0 / sample_rate) % 1 in fp32, as the reference
        s += (double)hop * (double)rad;
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {                                            // 256 partials: serial scan is fine
        double run = 0.0;
        for (int i = 0; i < 256; ++i) { const double v = part[i]; part[i] = run; run += v; run -= floor(run); }
    }
    __syncthreads();
    double run = part[tid];
    for (int i = lo; i < hi; ++i) {
        c[i] = run;
        const float rad = fmodf(f[i] / sr, 1.0f);
        run += (double)hop * (double)rad;
        run -= floor(run);
    }
}

struct SigArgs {
    int types[4];      // 0 noise, 1 sine, 2 uv  (channel order of the output)
    int ntypes;
};

__global__ __launch_bounds__(256)
void synth_kernel(const float* __restrict__ f0, const double* __restrict__ cf, float* __restrict__ out,
                  int F, int hop, float sr, float sine_amp, float noise_amp, SigArgs a, uint64_t seed) {
    const int b = blockIdx.y;
    const long T = (long)F * hop;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int fr = (int)(t / hop);
    const int j = (int)(t - (long)fr * hop);
    const float f0v = f0[(long)b * F + fr];
    const float vuv = f0v > 0.f ? 1.f : 0.f;
    for (int k = 0; k < a.ntypes; ++k) {
        float v;
        const uint64_t ctr = ((uint64_t)(b * 4 + k) << 40) + (uint64_t)t;
        if (a.types[k] == 1) {
            const float rad = fmodf(f0v / sr, 1.0f);
            double ph = cf[(long)b * F + fr] + (double)(j + 1) * (double)rad;
            ph -= floor(ph);
            v = vuv * (float)sin(ph * TWO_PI) * sine_amp;
            if (noise_amp > 0.f) v += gauss(seed, ctr) * (vuv * noise_amp + (1.0f - vuv) * noise_amp / 3.0f);
        } else if (a.types[k] == 0) {
            v = gauss(seed ^ 0x5851F42D4C957F2Dull, ctr);
        } else {
            v = vuv;
        }
        out[((long)b * a.ntypes + k) * T + t] = v;
    }
}

thread_local char g_sig_err[160];

}  // namespace

extern "C" {

size_t fastsvc_signal_scratch_bytes(int32_t B, int32_t F) { return (size_t)B * F * sizeof(double); }

int fastsvc_signal_generate(const float* f0, float* out, void* scratch, int32_t B, int32_t F, int32_t hop,
                            float sample_rate, float sine_amp, float noise_amp,
                            const int32_t* types, int32_t ntypes, uint64_t seed, void* stream_) {
    if (!f0 || !out || !scratch || !types || B < 1 || F < 1 || hop < 1 || ntypes < 1 || ntypes > 4 || sample_rate <= 0.f)
        return FASTSVC_E_INVALID;
    SigArgs a;
    a.ntypes = ntypes;
    for (int i = 0; i < 4; ++i) a.types[i] = i < ntypes ? types[i] : 0;
    for (int i = 0; i < ntypes; ++i)
        if (types[i] < 0 || types[i] > 2) return FASTSVC_E_INVALID;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* cf = static_cast<double*>(scratch);
    hipLaunchKernelGGL(frame_phase_kernel, dim3(B), dim3(256), 0, stream, f0, cf, F, hop, sample_rate);
    const long T = (long)F * hop;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((T + 255) / 256), B), dim3(256), 0, stream,
                       f0, cf, out, F, hop, sample_rate, sine_amp, noise_amp, a, seed);
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
