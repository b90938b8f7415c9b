// fastsvc_stage.hip - batch assembly on gfx950: B utterances of different lengths, each a (C, len_b) float32 block
// somewhere in device memory, into ONE zero-padded (B, C, width) batch - what the decode harness and the
// utterance-parallel driver hand to fastsvc_forward with per-utterance `lengths`.
//
// The reference decodes one utterance at a time (decode_fastsvc.py:160-200), so it has no counterpart; the batched
// path needs it because features that are already on the device (an upstream PPG / F0 model's outputs) would
// otherwise be assembled by B separate copies per tensor: queued behind a busy stream they cost ~15 us apiece on
// MI355X, 42 ms of a 185 ms pass over 512 utterances (tools/ragged_check.py).  Here: one launch per tensor, the
// source pointers / lengths / row pitches travel IN the kernel arguments (no table upload, nothing to keep alive).
// Pure data movement: coalesced element-wise reads (the sources have arbitrary alignment), 16-byte stores where the
// destination row allows them; HBM-bound at (bytes read + bytes written) / 8 TB/s, and three orders of magnitude
// below the forward it feeds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastsvc_hip.h"

namespace {

constexpr int GATHER_MAX = 64;                      // utterances per launch (the arguments hold their descriptors)

struct GatherArgs {
    const float* src[GATHER_MAX];
    int len[GATHER_MAX];                            // valid columns of every row of utterance b
    int pitch[GATHER_MAX];                          // elements between consecutive rows of utterance b
};

__global__ __launch_bounds__(256)
void gather_padded_kernel(GatherArgs a, float* __restrict__ dst, int C, int width) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t4 >= width) return;
    const float* s = a.src[b] + (long)c * a.pitch[b];
    const int len = a.len[b];
    float v[4];
    #pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (t4 + e < len) ? s[t4 + e] : 0.f;
    float* d = dst + ((long)b * C + c) * width + t4;
    if (t4 + 3 < width && ((reinterpret_cast<uintptr_t>(d) & 15) == 0)) {
        *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        #pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t4 + e < width) d[e] = v[e];
    }
}

}  // namespace

extern "C" {

int fastsvc_gather_padded(const float* const* src, const int32_t* lens, const int32_t* pitches, float* dst,
                          int32_t B, int32_t C, int32_t width, void* stream_) {
    if (!src || !lens || !pitches || !dst || B < 1 || C < 1 || width < 1 || C > 65535) return FASTSVC_E_INVALID;
    for (int b = 0; b < B; ++b)
        if ((!src[b] && lens[b] > 0) || lens[b] < 0 || lens[b] > width || pitches[b] < lens[b]) return FASTSVC_E_INVALID;   // (an empty utterance may have no storage)
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int b0 = 0; b0 < B; b0 += GATHER_MAX) {
        const int nb = B - b0 < GATHER_MAX ? B - b0 : GATHER_MAX;
        GatherArgs a;
        for (int i = 0; i < GATHER_MAX; ++i) {
            a.src[i] = i < nb ? src[b0 + i] : nullptr;
            a.len[i] = i < nb ? lens[b0 + i] : 0;
            a.pitch[i] = i < nb ? pitches[b0 + i] : 0;
        }
        hipLaunchKernelGGL(gather_padded_kernel, dim3((unsigned)((width + 1023) / 1024), (unsigned)C, (unsigned)nb), dim3(256), 0,
                           stream, a, dst + (long)b0 * C * width, C, width);
    }
    return hipGetLastError() == hipSuccess ? FASTSVC_OK : FASTSVC_E_HIP;
}

}  // extern "C"
