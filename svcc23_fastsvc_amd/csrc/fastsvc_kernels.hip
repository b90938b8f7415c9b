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
#ifdef FASTSVC_ACT_BF16
namespace bf16 {          // second compilation of this file: bfloat16 activation storage (see act_t below)
#endif

#include "fastsvc_device.inc"

// One packed weight fragment = MW consecutive floats per lane.  It is kept as a VECTOR value so
// that it lives in consecutive VGPRs: a global_load_dwordx{2,3} can then land directly in the
// ring slot and stay in flight (an array of scalars makes hipcc load into a temporary tuple and
// v_mov it out, which costs an s_waitcnt vmcnt(0) per fragment).
template <int MW> struct WFrag;
template <> struct WFrag<1> { typedef float type; };
template <> struct WFrag<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct WFrag<3> { typedef float type __attribute__((ext_vector_type(3))); };

struct __attribute__((packed, aligned(4))) PackedF3 { float a, b, c; };

template <int MW>
__device__ __forceinline__ typename WFrag<MW>::type load_wfrag(const float* p) {
    if constexpr (MW == 1) {
        return p[0];
    } else if constexpr (MW == 2) {
        return *reinterpret_cast<const typename WFrag<2>::type*>(p);      // 8-byte aligned by layout
    } else {
        const PackedF3 v = *reinterpret_cast<const PackedF3*>(p);         // 12-byte fragment
        typename WFrag<3>::type r;
        r.x = v.a; r.y = v.b; r.z = v.c;
        return r;
    }
}

template <int MW>
__device__ __forceinline__ float wfrag_get(const typename WFrag<MW>::type& w, int m) {
    if constexpr (MW == 1) return w;
    else return w[m];
}

// ---------------------------------------------------------------------------------------------
// Shared pieces of the two convolution kernels
// ---------------------------------------------------------------------------------------------

// Weight fragments are streamed from L2 straight into VGPRs in packed fragment order.  A "group"
// is the kg = KC/4 <= 6 k-steps of one tap of one chunk; the register ring `wr` always holds the
// group being multiplied, and slot j is re-requested with fragment j of the NEXT group right after
// its last MFMA, so every fragment is in flight for a whole group (kg * NW * MW MFMAs) before it is
// needed.  Slots are indexed statically (no register shifting: moving an in-flight load would
// force a wait).  The group sequence is cyclic over the layer, so the stream continues seamlessly
// across chunks and across the consecutive time tiles a workgroup walks.
constexpr int WG_MAX = 6;

template <int MW>
struct WeightStream {
    typename WFrag<MW>::type wr[WG_MAX];
    const float* next;     // fragment 0 of the group after the one held in wr
    const float* base;     // first fragment of the layer (this wave's channel group, this lane)
    const float* end;      // one past the last

    __device__ __forceinline__ void init(const float* b, int q_total, int kg) {
        base = b;
        end = b + (long)q_total * 64 * MW;
        #pragma unroll
        for (int j = 0; j < WG_MAX; ++j) {
            wr[j] = typename WFrag<MW>::type(0.f);
            if (j < kg) wr[j] = load_wfrag<MW>(b + (long)j * 64 * MW);
        }
        next = b + (long)kg * 64 * MW;
        if (next >= end) next = base;
    }
};

// All k-steps of one staged chunk.  xa0: this lane's LDS address for tap 0, k-group 0.
// KG > 0: compile-time group size (straight-line steps, so hipcc can emit counted vmcnt waits
// and keep the other fragments in flight); KG == 0: run-time group size (generic kernel).
template <int MW, int NW, int KG>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[NW][MW], const float* xa0, int XS,
                                           WeightStream<MW>& ws, int ntaps, int kg, int dil) {
    for (int tap = 0; tap < ntaps; ++tap) {
        const float* xa = xa0 + tap * dil;
        if constexpr (KG > 0) {
            #pragma unroll
            for (int j = 0; j < KG; ++j) {
                float av[NW];
                #pragma unroll
                for (int n = 0; n < NW; ++n) av[n] = xa[j * 4 * XS + n * 16];
                #pragma unroll
                for (int n = 0; n < NW; ++n)
                    #pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n], wfrag_get<MW>(ws.wr[j], m), acc[n][m], 0, 0, 0);
                ws.wr[j] = load_wfrag<MW>(ws.next + (long)j * 64 * MW);
                // pin the re-request right behind its step: hipcc otherwise sinks all six loads to
                // the end of the tap and waits for them at the top of the next one
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            #pragma unroll
            for (int j = 0; j < WG_MAX; ++j) {
                if (j < kg) {
                    float av[NW];
                    #pragma unroll
                    for (int n = 0; n < NW; ++n) av[n] = xa[n * 16];
                    #pragma unroll
                    for (int n = 0; n < NW; ++n)
                        #pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n], wfrag_get<MW>(ws.wr[j], m), acc[n][m], 0, 0, 0);
                    ws.wr[j] = load_wfrag<MW>(ws.next + (long)j * 64 * MW);
                    xa += 4 * XS;
                }
            }
        }
        ws.next += (long)kg * 64 * MW;
        if (ws.next >= ws.end) ws.next = ws.base;
    }
}

// Pipelined kernel: unit-deep weight stream.  One unit = one K chunk of one tile = 3 taps x 6
// k-steps = 18 fragments, all slots static.  Slot s is re-requested with fragment s of the NEXT
// unit right after its MFMAs, so every fragment flies for a whole unit, and because the staging
// loads of the next unit are issued at the top of the unit (younger than every fragment consumed
// in it) the counted waits never drain them: both streams overlap the matrix work completely.
// For single-chunk layers (C_in = 24) the "next unit" is the same chunk: the layer's weights
// simply stay in registers.
constexpr int UNIT_STEPS = 18;      // k = 3: 3 taps x 6 k-steps; the 1x1 variant uses 6

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// Fragment load through a buffer descriptor: address = SRD base (SGPRs, wave-uniform: this wave's
// channel group) + voffset (VGPR: lane * MW * 4, constant) + soffset (SGPR: unit and step offset).
// No per-step VGPR address arithmetic, no 64-bit address registers.
template <int MW>
__device__ __forceinline__ typename WFrag<MW>::type load_wfrag_buf(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    if constexpr (MW == 1) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    } else if constexpr (MW == 2) {
        return __builtin_bit_cast(typename WFrag<2>::type, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
    } else {
        return __builtin_bit_cast(typename WFrag<3>::type, __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff, 0));
    }
}

template <int MW, int NSTEPS = UNIT_STEPS>
struct UnitWeightStream {
    static constexpr int STEP_BYTES = 64 * MW * 4;
    static constexpr int UNIT_BYTES = NSTEPS * STEP_BYTES;
    typename WFrag<MW>::type wr[NSTEPS];
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;              // lane * MW * 4
    int next;              // byte offset of the unit after the one held in wr
    int total;             // bytes of this channel group's packed weights

    __device__ __forceinline__ void init(const float* group_base, int q_total, int lane) {
        total = q_total * STEP_BYTES;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(group_base), 0, total, 0x00020000);
        voff = lane * MW * 4;
        #pragma unroll
        for (int s = 0; s < NSTEPS; ++s) wr[s] = load_wfrag_buf<MW>(rsrc, voff, s * STEP_BYTES);
        next = UNIT_BYTES;
        if (next >= total) next = 0;
    }
};

// RELOAD false: the whole layer is ONE unit of weights (C_in <= 24), the ring just stays resident.  Apart
// from the L2 traffic saved, the consumer's MFMA loop then has no vmcnt wait at all: on gfx9 loads
// and stores share that counter in order, so every wait on a re-requested slot also drained the
// previous tile's epilogue stores (and the LDS-DMA pieces of this one) in the middle of the loop.
template <int MW, int NW, int NSTEPS, bool RELOAD = true>
__device__ __forceinline__ void mfma_unit(f32x4 (&acc)[NW][MW], const float* xa0, int XS,
                                          UnitWeightStream<MW, NSTEPS>& ws, int dil) {
    // The LDS reads of step s+LA are issued BEFORE the MFMAs of step s (register ring by full
    // unrolling), so their latency hides under the matrix work even with one wave per SIMD.
    // Look-ahead LA = 2 steps when a step is short (few MFMAs), else 1.
    constexpr int LA = (NW * MW <= 4) ? 2 : 1;
    float av[LA + 1][NW];
    #pragma unroll
    for (int q = 0; q < LA; ++q) {
        if (q < NSTEPS) {
            const int tap = q / 6, j = q % 6;
            #pragma unroll
            for (int n = 0; n < NW; ++n) av[q][n] = xa0[tap * dil + j * 4 * XS + n * 16];
        }
    }
    #pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        if (s + LA < NSTEPS) {
            const int tap = (s + LA) / 6, j = (s + LA) % 6;
            #pragma unroll
            for (int n = 0; n < NW; ++n) av[(s + LA) % (LA + 1)][n] = xa0[tap * dil + j * 4 * XS + n * 16];
            __builtin_amdgcn_sched_barrier(0);         // ... and stay before them (hipcc sinks them behind otherwise)
        }
        #pragma unroll
        for (int n = 0; n < NW; ++n)
            #pragma unroll
            for (int m = 0; m < MW; ++m)
                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s % (LA + 1)][n], wfrag_get<MW>(ws.wr[s], m), acc[n][m], 0, 0, 0);
        if constexpr (RELOAD)
            ws.wr[s] = load_wfrag_buf<MW>(ws.rsrc, ws.voff, ws.next + s * UnitWeightStream<MW, NSTEPS>::STEP_BYTES);
        // pin the re-request right behind its step (hipcc otherwise sinks the loads to the end
        // of the unit and waits for all of them at the top of the next one)
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (RELOAD) {
        ws.next += UnitWeightStream<MW, NSTEPS>::UNIT_BYTES;
        if (ws.next >= ws.total) ws.next = 0;
    }
}

// Polyphase unit (MODE_POLY): per 4-channel k-group read x[j-1], x[j], x[j+1] from the window
// (the same three LDS reads a 3-tap conv makes), form the two differences on the VALU and feed three
// accumulator sets: acc[0] += (x[j-1]-x[j]) W0,  acc[1] += x[j] (W0+W1+W2),  acc[2] += (x[j+1]-x[j]) W2.
// Weight slots: tap-major like every unit (slot = tap * 6 + k-group), so the stream is unchanged.
template <int MW, int NW, bool RELOAD = true>
__device__ __forceinline__ void mfma_unit_poly(f32x4 (&acc)[3][NW][MW], const float* xa0, int XS,
                                               UnitWeightStream<MW, UNIT_STEPS>& ws) {
    float av[2][3][NW];                                     // k-group ring, one group of look-ahead
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap)
        #pragma unroll
        for (int n = 0; n < NW; ++n) av[0][tap][n] = xa0[tap + n * 16];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (j + 1 < 6) {
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap)
                #pragma unroll
                for (int n = 0; n < NW; ++n) av[(j + 1) & 1][tap][n] = xa0[tap + (j + 1) * 4 * XS + n * 16];
            __builtin_amdgcn_sched_barrier(0);
        }
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            #pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float a = (tap == 1) ? av[j & 1][1][n] : av[j & 1][tap][n] - av[j & 1][1][n];
                #pragma unroll
                for (int m = 0; m < MW; ++m)
                    acc[tap][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wfrag_get<MW>(ws.wr[tap * 6 + j], m), acc[tap][n][m], 0, 0, 0);
            }
            if constexpr (RELOAD)
                ws.wr[tap * 6 + j] = load_wfrag_buf<MW>(ws.rsrc, ws.voff, ws.next + (tap * 6 + j) * UnitWeightStream<MW, UNIT_STEPS>::STEP_BYTES);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (RELOAD) {
        ws.next += UnitWeightStream<MW, UNIT_STEPS>::UNIT_BYTES;
        if (ws.next >= ws.total) ws.next = 0;
    }
}

// Winograd F(2,3) unit (MODE_WINO).  Weight steps of a unit are packed HALF-major:
// step = (half * 4 + component) * 3 + k-group-in-half, so the same stream feeds either a unit-deep
// ring (RING = 24 slots) or a half-unit ring (RING = 12: the NW = 1 variant must fit 128 VGPRs next
// to its four accumulator sets; a slot then flies for half a unit = 12 steps).
// xr / xrd: this lane's LDS addresses of d1 = x[t] (plane r) and d2 = x[t+d] (plane r+d) for
// k-group 0 of M-tile 0; d3 = x[t+2d] is the next entry of plane r, d0 = x[t-d] the previous entry
// of plane r+d.  M-tile n starts 16 pairs = 16/D plane positions further on.
template <int MW, int NW, int D, int RING, bool RELOAD = true>
__device__ __forceinline__ void mfma_unit_wino(f32x4 (&acc)[4][NW][MW], const float* xr, const float* xrd, int XS,
                                               UnitWeightStream<MW, RING>& ws) {
    constexpr int TSTEP = 16 / D;
    constexpr int STEP_BYTES = UnitWeightStream<MW, RING>::STEP_BYTES;
    float av[2][4][NW];                                     // [ring][d0,d1,d2,d3][n]
    auto fetch = [&](int slot, int j) {
        #pragma unroll
        for (int n = 0; n < NW; ++n) {
            av[slot][1][n] = xr[j * 4 * XS + n * TSTEP];
            av[slot][3][n] = xr[j * 4 * XS + n * TSTEP + 1];
            av[slot][2][n] = xrd[j * 4 * XS + n * TSTEP];
            av[slot][0][n] = xrd[j * 4 * XS + n * TSTEP - 1];
        }
    };
    fetch(0, 0);
    #pragma unroll
    for (int h = 0; h < 2; ++h) {
        #pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int j = 3 * h + jj;
            if (j + 1 < 6) { fetch((j + 1) & 1, j + 1); __builtin_amdgcn_sched_barrier(0); }   // reads stay ahead of the MFMAs
            #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = (RING == 24 ? h * 12 : 0) + c * 3 + jj;
                #pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const float d0 = av[j & 1][0][n], d1 = av[j & 1][1][n], d2 = av[j & 1][2][n], d3 = av[j & 1][3][n];
                    const float a = c == 0 ? d0 - d2 : c == 1 ? d1 + d2 : c == 2 ? d2 - d1 : d1 - d3;
                    #pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[c][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wfrag_get<MW>(ws.wr[slot], m), acc[c][n][m], 0, 0, 0);
                }
                if constexpr (RELOAD || RING == 12)
                    ws.wr[slot] = load_wfrag_buf<MW>(ws.rsrc, ws.voff, ws.next + slot * STEP_BYTES);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (RING == 12) {                                   // next half-unit
            ws.next += UnitWeightStream<MW, RING>::UNIT_BYTES;
            if (ws.next >= ws.total) ws.next = 0;
        }
    }
    if (RING == 24 && RELOAD) {
        ws.next += UnitWeightStream<MW, RING>::UNIT_BYTES;
        if (ws.next >= ws.total) ws.next = 0;
    }
}

// MODE_DEC2 unit: acc[0] += lrelu(x[t-1]) w0 + lrelu(x[t]) w1 + lrelu(x[t+1]) w2,  acc[1] += x[t] w1x1.
// Weight steps are packed half-major like the Winograd ones (component 3 = the 1x1 weights).
template <int MW, int NW, bool RELOAD = true>
__device__ __forceinline__ void mfma_unit_dec2(f32x4 (&acc)[2][NW][MW], const float* xa0, int XS,
                                               UnitWeightStream<MW, 24>& ws) {
    constexpr int STEP_BYTES = UnitWeightStream<MW, 24>::STEP_BYTES;
    float av[2][3][NW];
    auto fetch = [&](int slot, int j) {
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            #pragma unroll
            for (int n = 0; n < NW; ++n) av[slot][tap][n] = xa0[tap + j * 4 * XS + n * 16];
    };
    fetch(0, 0);
    #pragma unroll
    for (int h = 0; h < 2; ++h) {
        #pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int j = 3 * h + jj;
            if (j + 1 < 6) { fetch((j + 1) & 1, j + 1); __builtin_amdgcn_sched_barrier(0); }
            #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = h * 12 + c * 3 + jj;
                #pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const float raw = av[j & 1][c == 3 ? 1 : c][n];
                    const float a = c == 3 ? raw : fmaxf(raw, raw * LRELU_SLOPE);
                    #pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[c == 3][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wfrag_get<MW>(ws.wr[slot], m), acc[c == 3][n][m], 0, 0, 0);
                }
                if constexpr (RELOAD)
                    ws.wr[slot] = load_wfrag_buf<MW>(ws.rsrc, ws.voff, ws.next + slot * STEP_BYTES);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (RELOAD) {
        ws.next += UnitWeightStream<MW, 24>::UNIT_BYTES;
        if (ws.next >= ws.total) ws.next = 0;
    }
}

// Epilogue of one time tile.  D layout (16x16x4 f32): lane holds column j = lane & 15 (output
// channel) and rows i = (lane >> 4) * 4 + r (time), r = 0..3 -> four consecutive time steps per
// lane.  s1/s2 accumulate the InstanceNorm partial sums across the tiles a workgroup walks.
template <int MW, int NW>
__device__ __forceinline__ void conv_epilogue_tile(const ConvParams& p, f32x4 (&acc)[NW][MW],
                                                   float (&s1)[MW], float (&s2)[MW],
                                                   int sig, int b, int mg, int tcol0, bool active, int lane, float& amx) {
    const int flags = p.flags;
    if (FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE)) {
        // keep the accumulators live (never true for finite data), then leave
        float keep = 0.f;
        #pragma unroll
        for (int n = 0; n < NW; ++n)
            #pragma unroll
            for (int m = 0; m < MW; ++m) keep += acc[n][m].x + acc[n][m].y + acc[n][m].z + acc[n][m].w;
        if (keep == 1.2345678e33f) (p.y ? p.y : p.y2)[0] = keep;
        return;
    }
    if (!active) return;
    const float* biasp = p.bias + (long)sig * p.bias_sig;
    float* ybase = p.y ? p.y + (long)sig * p.y_sig + (long)b * p.y_b : nullptr;
    float* y2base = (flags & F_AFF_OUT) ? p.y2 + (long)b * p.y2_b : nullptr;
    const float* resbase = p.res ? p.res + (long)sig * p.res_sig + (long)b * p.res_b : nullptr;
    const float* r1x = p.r1x ? p.r1x + (long)sig * p.r1x_sig + (long)b * p.r1x_b : nullptr;
    const float* ssob = (flags & (F_STATS | F_AFF_OUT)) ? p.ss_out + (long)b * p.ss_out_b : nullptr;
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int co = (mg * MW + m) * 16 + (lane & 15);
        if (co >= p.COUT) continue;
        const float bias = biasp[co];
        float r1w = 0.f, r1b = 0.f;
        if (r1x) { r1w = p.r1w[(long)sig * p.r1_sig + co]; r1b = p.r1b[(long)sig * p.r1_sig + co]; }
        float* yrow = ybase ? ybase + (long)co * p.ldy : nullptr;
        float* y2row = y2base ? y2base + (long)co * p.ldy : nullptr;
        const float* rrow = resbase ? resbase + (long)co * p.ldy : nullptr;
        const float* scrow = ssob ? ssob + (long)co * p.ldy : nullptr;
        const float* shrow = ssob ? ssob + (long)(p.COUT + co) * p.ldy : nullptr;
        #pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int t = tcol0 + n * 16 + (lane >> 4) * 4;
            if (t >= p.T) continue;
            f32x4 v = acc[n][m];
            v += bias;
            if (flags & F_POST_LRELU) {
                v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w);
            }
            if (p.vec) {
                if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + t);
                if (r1x) v += *reinterpret_cast<const f32x4*>(r1x + t) * r1w + r1b;
                if (yrow) *reinterpret_cast<f32x4*>(yrow + t) = v;
                if (scrow) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(scrow + t) * v
                                  + *reinterpret_cast<const f32x4*>(shrow + t);
                    if (y2row) *reinterpret_cast<f32x4*>(y2row + t) = u;
                    s1[m] += (u.x + u.y) + (u.z + u.w);
                    s2[m] += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
                } else amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            } else {
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (t + r >= p.T) break;
                    float e = v[r];
                    if (rrow) e += rrow[t + r];
                    if (r1x) e += r1x[t + r] * r1w + r1b;
                    if (yrow) yrow[t + r] = e;
                    if (scrow) {
                        const float u = scrow[t + r] * e + shrow[t + r];
                        if (y2row) y2row[t + r] = u;
                        s1[m] += u; s2[m] += u * u;
                    } else amx = fmaxf(amx, fabsf(e));
                }
            }
        }
    }
}

// Flush the InstanceNorm partial sums: 4 lane groups share a channel -> shuffle; the waves that
// share a channel -> LDS f64 atomics; one f64 global atomic per channel per workgroup.
template <int MW, int WM, int NTHREADS>
__device__ __forceinline__ void stats_flush(const ConvParams& p, double (&d1a)[MW], double (&d2a)[MW],
                                            double* sstat, int b, int wave_m, bool active, int tid, int lane) {
    if (!(p.flags & F_STATS) || (FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) return;
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        double d1 = d1a[m], d2 = d2a[m];
        d1 += __shfl_xor(d1, 16); d2 += __shfl_xor(d2, 16);
        d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
        if (active && lane < 16) {      // waves that hold no sums (producer waves) pass active = false
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

// InstanceNorm constants of the input channels from the producer's f64 sums (fastsvc.py:138)
template <int NTHREADS>
__device__ __forceinline__ void norm_constants(const ConvParams& p, int b, int CINp, int tid,
                                               float* nmean, float* nrstd, float* nspk) {
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

// ---------------------------------------------------------------------------------------------
// Generic convolution (any length, any index mode; scalar staging).  Used for the decimating
// layers and whenever T is not a multiple of 4.  gridDim = (ceil(T/NT), ceil(ngroups/WM), nsig*B)
// ---------------------------------------------------------------------------------------------
template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_mfma_kernel(const ConvParams p0) {
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
    const int sig = z / p0.B;
    const int b = z - sig * p0.B;
    ConvParams p = p0;                                  // ragged batch: this utterance's row lengths
    if (p0.lens) {
        const int frames = p0.lens[b];
        p.T = frames * p0.len_mul;
        p.x_T = frames * p0.xlen_mul;
        // the float4 epilogue needs THIS utterance's rows to be a multiple of 4 long, not the padded maximum: the
        // straddling float4 would store, and sum into the InstanceNorm statistics, what lies past the row end
        if (p.T & 3) p.vec = 0;
    }
    const int t0 = blockIdx.x * NT;
    if (t0 >= p.T) return;
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

    __shared__ unsigned s_amax, s_cnt;                 // (zeroed here; the chunk loop's barriers separate it from the flush)
    if (tid == 0) { s_amax = 0u; s_cnt = 0u; }
    if (p.flags & F_STATS) {
        for (int i = tid; i < 2 * 16 * MW * WM; i += NTHREADS) sstat[i] = 0.0;
    }
    if (p.flags & F_PRE_NORM) norm_constants<NTHREADS>(p, b, CINp, tid, nmean, nrstd, nspk);

    f32x4 acc[NW][MW];
    #pragma unroll
    for (int n = 0; n < NW; ++n)
        #pragma unroll
        for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* xbase = p.x + (long)sig * p.x_sig + (long)b * p.x_b;
    const float* ssbase = (p.flags & F_PRE_AFFINE) ? p.ss_in + (long)b * p.ss_in_b : nullptr;
    const int kg = p.KC >> 2;
    WeightStream<MW> wst;
    wst.init(p.w + (long)sig * p.w_sig + ((long)(active ? mg : 0) * p.Q * 64 + lane) * MW, p.Q, kg);
    const int flags = p.flags;
    const int mode = p.mode;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        __syncthreads();   // previous chunk's LDS reads are done (first pass: norm constants visible)
        for (int r = wave; r < p.KC; r += NWAVES) {
            const int ci = ch * p.KC + r;
            float* row = Xs + r * XS;
            if (ci >= p.CIN) {
                for (int j = lane; j < W; j += 64) row[j] = 0.f;
                continue;
            }
            const float* xrow = xbase + (long)ci * p.ldx;
            const float* scrow = nullptr;
            const float* shrow = nullptr;
            float mean = 0.f, rstd = 1.f, pb = 0.f;
            if (flags & F_PRE_AFFINE) {
                scrow = ssbase + (long)ci * p.ldx;
                shrow = ssbase + (long)(p.CIN + ci) * p.ldx;
            }
            if (flags & F_PRE_NORM) { mean = nmean[ci]; rstd = nrstd[ci]; pb = nspk[ci]; }
            for (int j = lane; j < W; j += 64) {
                const int t = t0 - halo + j;
                float v = 0.f;
                if (t >= 0 && t < p.T) {
                    const int src = (mode == MODE_DIRECT) ? t
                                  : (mode == MODE_DECIMATE) ? t * p.s : div_small(t, p.s);
                    v = xrow[src];
                    if (flags & F_PRE_AFFINE) v = scrow[src] * v + shrow[src];
                    if (flags & F_PRE_NORM) v = (v - mean) * rstd + pb;
                    if (flags & F_PRE_LRELU) v = lrelu(v);
                }
                row[j] = v;
            }
        }
        __syncthreads();
        if (active) {
            const float* xa0 = Xs + (lane >> 4) * XS + (lane & 15) + wave_n * (NW * 16);
            mfma_chunk<MW, NW, 0>(acc, xa0, XS, wst, p.ntaps, kg, p.dil);
        }
    }
    float s1[MW], s2[MW];
    double d1[MW], d2[MW];
    #pragma unroll
    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
    {
        EpiRsrc R;                                     // (only its running max is used by this kernel)
        conv_epilogue_tile<MW, NW>(p, acc, s1, s2, sig, b, mg, t0 + wave_n * (NW * 16), active, lane, R.amx);
        amax_flush(p, R, &s_amax, &s_cnt, NWAVES, sig, b, lane, blockIdx.x);
    }
    #pragma unroll
    for (int m = 0; m < MW; ++m) { d1[m] = (double)s1[m]; d2[m] = (double)s2[m]; }
    stats_flush<MW, WM, NTHREADS>(p, d1, d2, sstat, b, wave_m, active, tid, lane);
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised pipelined convolution: the hot kernel.  Requires T % 4 == 0 (float4
// everywhere), k = 3, KC = 24, index mode DIRECT or STRETCH.
//
//   workgroup = 8 waves = 4 CONSUMER waves + 4 PRODUCER waves (one of each per SIMD):
//   * consumer waves issue nothing but LDS reads, MFMAs and the packed-weight stream (unit-deep
//     register ring, counted waits), and write the tile epilogue;
//   * producer waves own all global staging: float4 loads of the input window (aligned window:
//     halo rounded up to 4 columns, fixed per-thread slots, clamped unconditional addresses),
//     two register sets deep (the loads of unit u+2 are in flight while unit u+1 is transformed:
//     InstanceNorm-apply + speaker bias as one FMA, LeakyReLU, zero padding) and written to the
//     LDS buffer the consumers will read next.  Each wave has its own vmcnt, so staging latency
//     never stalls the matrix pipe, and the two roles need max(), not sum(), of their registers;
//   * one s_barrier per unit = (time tile, K chunk); a workgroup walks p.tpw consecutive tiles of
//     one (signal, batch item, channel block): set-up, InstanceNorm coefficients and the weight
//     stream are paid once, the InstanceNorm partial sums are flushed once.
// ---------------------------------------------------------------------------------------------
// Register budget: 128 VGPRs (two workgroups per CU) where that fits without spills, else 256.
// Polyphase with MW == 3 carries three accumulator sets next to the 54-register weight ring and
// a 2*S-load epilogue: 256 (those layers launch about one workgroup per CU anyway).
// MW == 3, NW == 2 fits 128 only with the plain / residual epilogues (no extra epilogue operands).
constexpr bool NTAPS_IS_3_DIRECT(int mode) { return mode == MODE_DIRECT; }
// variants that stage their epilogue operands in LDS (ws_epilogue_stage)
template <int MW, int NW, int MODE, int EPI>
constexpr bool ws_estage() {
    return MODE == MODE_DIRECT &&
           ((EPI == EPI_AFF && MW * NW <= 4) || (EPI == EPI_RES && MW * NW <= 6));
}
template <int MW, int NW, int MODE, int EPI>
constexpr int ws_min_waves() {
#ifdef FASTSVC_ACT_BF16
    // bfloat16 storage: the pack / unpack temporaries push these variants over 128 registers
    if (MODE == MODE_POLY && EPI == EPI_AFF) return 2;
    if (MODE == MODE_DEC2 && NW == 2) return 2;
    if (MW == 2 && NW == 4 && (EPI == EPI_AFF || EPI == EPI_RANK1)) return 2;
    if (MW == 3 && NW == 2) return 2;
#endif
    if (MODE == MODE_POLY) return (MW <= 2 && NW == 1) ? 4 : 2;
    if (MODE == MODE_WINO) return (MW == 2 && NW == 1) ? 4 : 2;   // four accumulator sets + a 24-slot weight ring
    if (MODE == MODE_DEC2) return MW <= 2 ? 4 : 2;                // two accumulator sets + a 24-slot weight ring
    if (MW <= 2 || NW == 1) return 4;
    if (NW == 2 && NTAPS_IS_3_DIRECT(MODE) && (EPI == EPI_PLAIN || EPI == EPI_RES)) return 4;
    return 2;
}

// WSTATIC (MW == 2 instances only, chosen by launch_ws): the layer has a single unit of weights per
// channel group (Q == NSTEPS), see mfma_unit's RELOAD.
template <int MW, int NW, int WM, int WN, int MODE, int NTAPS, int EPI = EPI_GENERIC, int S = 1, bool WSTATIC = false>
__global__ __launch_bounds__(512, (ws_min_waves<MW, NW, MODE, EPI>()))
void conv_mfma_ws_kernel(const ConvParams p0) {
    constexpr bool WINO = (MODE == MODE_WINO);                         // S carries the dilation D
    constexpr bool DEC2 = (MODE == MODE_DEC2);
    constexpr int NSTEPS = (WINO || DEC2) ? 24 : 6 * NTAPS;            // weight ring slots
    constexpr int NT = (WINO ? 32 : 16) * NW * WN;                     // output columns per workgroup tile
    constexpr int NPROD = 256;                                         // producer threads
    constexpr int ITEMS = StageGeom<NT, NPROD>::ITEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // 0..3 consumers, 4..7 producers
    const bool producer = wave >= 4;
    const int cw = wave & 3;
    const int wave_m = cw / WN;
    const int wave_n = cw - wave_m * WN;
    const int z = blockIdx.z;
    const int sig = z / p0.B;
    const int b = z - sig * p0.B;
    // ragged batch: this utterance's own row lengths (tiles, masks, InstanceNorm length); the row
    // pitches ldx / ldy stay those of the longest utterance
    ConvParams p = p0;
    if (p0.lens) {
        const int frames = p0.lens[b];
        p.T = frames * p0.len_mul;
        p.x_T = frames * p0.xlen_mul;
    }
    const int mg = blockIdx.y * WM + wave_m;
    const bool active = !producer && mg < p.ngroups;
    const int halo = (NTAPS == 3) ? p.dil : 0;
    const int halo_al = (halo + 3) & ~3;
    const int W4 = (NT + 2 * halo_al) >> 2;
    const int XS = p.xs;
    const int CINp = p.nchunks * p.KC;
    const int flags = p.flags;
    const int ntx = (p.T + NT - 1) / NT;
    const int tile0 = blockIdx.x * p.tpw;
    const int ntiles = min(p.tpw, ntx - tile0);
    if (ntiles <= 0) return;                                           // ragged batch: nothing of this utterance here
    const int nunits = ntiles * p.nchunks;
#ifdef FASTSVC_TIMELINE
    // diagnostic build: lane 0 of every wave stamps s_memtime at its phase boundaries
    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long* tlw = (p.tl && wg_lin < p.tl_wgs) ? p.tl + ((long)wg_lin * 8 + wave) * 64 : nullptr;
    int tli = 0;
    auto stamp = [&](int tag) {
        if (tlw && lane == 0 && tli < 62) { tlw[tli++] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); }
    };
    stamp(1);                                               // kernel entry
    if (tlw && lane == 0) {
        tlw[62] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID: wave / simd / cu / sh / se
        tlw[63] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // XCC_ID
    }
#else
    auto stamp = [](int) {};
#endif

    double* sstat = reinterpret_cast<double*>(smem_raw);                       // [WM*MW*16][2]
    float2* ncoef = reinterpret_cast<float2*>(smem_raw + sizeof(double) * 2 * 16 * MW * WM);  // [CINp]
    float* Xs0 = reinterpret_cast<float*>(ncoef + CINp);                       // [2][KC][XS]
    const int bufsz = p.KC * XS;
    __shared__ unsigned s_amax, s_cnt;                 // workgroup's largest |value| written (ConvParams::amax_out), waves done
    // this instance's epilogue measures what it writes: residual kinds (direct / Winograd) and the generic epilogue
    constexpr bool TRACKS = AMAX_TRACK && (MODE == MODE_DIRECT || MODE == MODE_WINO || MODE == MODE_STRETCH || MODE == MODE_DECIMATE) &&
                            (EPI == EPI_RES || EPI == EPI_RANK1 || EPI == EPI_GENERIC);

    // InstanceNorm coefficients and zeroed sums: run by ALL threads, but only after the producers
    // have their first two window loads and the consumers their weight stream in flight, so the
    // st_in round trip overlaps them instead of preceding them.
    auto setup_shared = [&]() {
        if constexpr (TRACKS) { if (tid == 0) { s_amax = 0u; s_cnt = 0u; } }
        if (flags & F_STATS) {
            for (int i = tid; i < 2 * 16 * MW * WM; i += 512) sstat[i] = 0.0;
        }
        if (flags & F_PRE_NORM) {
            // (u - mean) * rstd + p  ==  u * A + Bc  with A = rstd, Bc = p - mean * rstd
            const double inv_len = 1.0 / (double)p.x_T;
            for (int c = tid; c < CINp; c += 512) {
                float2 ab = make_float2(0.f, 0.f);
                if (c < p.CIN) {
                    const double q1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
                    const double q2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
                    const double mean = q1 * inv_len;
                    double var = q2 * inv_len - mean * mean;      // biased variance (InstanceNorm2d)
                    var = var > 0.0 ? var : 0.0;
                    const double rstd = 1.0 / sqrt(var + IN_EPS);
                    ab.x = (float)rstd;
                    ab.y = (float)((double)p.spk[(long)b * p.CIN + c] - mean * rstd);
                }
                ncoef[c] = ab;
            }
        }
        __syncthreads();                                   // coefficients / zeroed sums visible
    };

    if (producer) {
        // ================================ PRODUCER WAVES ================================
        const int ptid = tid - 256;
        int loff[ITEMS];      // LDS float offset inside a buffer, -1: slot outside the window
        int rq[ITEMS];        // (row inside the chunk) << 16 | first column relative to the window start
        const float inv_w4 = 1.0f / (float)W4;
        #pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int idx = i * NPROD + ptid;
            const int r = (int)(((float)idx + 0.5f) * inv_w4);
            const int q = idx - r * W4;
            loff[i] = (r < p.KC) ? r * XS + 4 * q : -1;
            rq[i] = (r << 16) | (4 * q);
        }
        // source rows of this (signal, batch item) through a buffer descriptor: offsets outside the
        // tensor (columns before the first / after the last row, channel padding) read as 0 in
        // hardware, everything else is real memory and is masked by `okmask` where it is padding.
        const __amdgpu_buffer_rsrc_t xr =
            act_rsrc(p.x, (long)sig * p.x_sig + (long)b * p.x_b, (long)p.CIN * p.ldx);

        // unconditional loads of unit `un` into a register set; validity in the mask
        auto pload = [&](int un, f32x4 (&px)[ITEMS], unsigned& okmask) {
            const int tl = un / p.nchunks;
            const int ch = un - tl * p.nchunks;
            const int t_start = (tile0 + tl) * NT - halo_al;
            const int soff = ch * p.KC * p.ldx * 4;
            const int rows_left = p.CIN - ch * p.KC;          // rows >= this are channel padding
            okmask = 0;
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int r = rq[i] >> 16;
                const int t = t_start + (rq[i] & 0xffff);
                const int ro = r * p.ldx;
                const bool ok = loff[i] >= 0 && (unsigned)t < (unsigned)p.T && r < rows_left;
                okmask |= (ok ? 1u : 0u) << i;
                if (MODE == MODE_STRETCH) {
                    const int tc = max(t, 0);
                    const unsigned src0 = udiv_small((unsigned)tc, p.s);
                    const int ph = tc - (int)src0 * p.s;
                    const int o = (ro + (int)src0) * 4;
                    px[i].x = act_load1(xr, o, soff);
                    px[i].y = act_load1(xr, o + 4 * (int)udiv_small(ph + 1, p.s), soff);
                    px[i].z = act_load1(xr, o + 4 * (int)udiv_small(ph + 2, p.s), soff);
                    px[i].w = act_load1(xr, o + 4 * (int)udiv_small(ph + 3, p.s), soff);
                } else if (MODE == MODE_DECIMATE || MODE == MODE_DEC2) {
                    const int o = (ro + t * p.s) * 4;           // x[..., ::s]; negative t -> out of range -> 0
                    px[i].x = act_load1(xr, o, soff);
                    px[i].y = act_load1(xr, o + 4 * p.s, soff);
                    px[i].z = act_load1(xr, o + 8 * p.s, soff);
                    px[i].w = act_load1(xr, o + 12 * p.s, soff);
                } else {
                    px[i] = act_load4(xr, (ro + t) * 4, soff);
                }
            }
            if (FASTSVC_DBG_ON(p, DBG_NO_LOAD)) okmask = 0;
        };
        // prologue transform + LDS write of a register set
        auto pcommit = [&](int un, const f32x4 (&px)[ITEMS], unsigned okmask, float* Xs) {
            if (FASTSVC_DBG_ON(p, DBG_NO_COMMIT)) return;
            const int ch = un % p.nchunks;
            const int t_start = (tile0 + un / p.nchunks) * NT - halo_al;
            const bool tailmode = (p.T & 3) != 0;              // rows are not a multiple of 4 long
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                if (loff[i] < 0) continue;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};          // zero "same" padding / channel padding
                if (okmask & (1u << i)) {
                    v = px[i];

                    if (flags & F_PRE_NORM) {
                        const float2 ab = ncoef[ch * p.KC + (rq[i] >> 16)];
                        v = v * ab.x + ab.y;
                    }
                    if (flags & F_PRE_LRELU) {
                        v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w);
                    }
                    // the float4 that straddles the row end also holds the head of the next row: the
                    // conv's zero padding starts there (after the prologue transforms, like every pad)
                    if (tailmode) v = keep_first(v, p.T - (t_start + (rq[i] & 0xffff)));
                }
                if constexpr (WINO) {
                    // de-interleave into the 2*D phase planes: x[t] -> plane t % 2D, position t / 2D
                    // (relative to tile start - 8; the window starts at tile start - 4)
                    float* row = Xs + (rq[i] >> 16) * XS;
                    const int u = (rq[i] & 0xffff) + 4;
                    const int PS = p.ps;
                    if (S == 1) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<f32x2*>(row + (u >> 1)) = f32x2{v.x, v.z};
                        *reinterpret_cast<f32x2*>(row + PS + (u >> 1)) = f32x2{v.y, v.w};
                    } else if (S == 2) {
                        float* d = row + (u >> 2);
                        d[0] = v.x; d[PS] = v.y; d[2 * PS] = v.z; d[3 * PS] = v.w;
                    } else {
                        float* d = row + (u & 7) * PS + (u >> 3);
                        d[0] = v.x; d[PS] = v.y; d[2 * PS] = v.z; d[3 * PS] = v.w;
                    }
                } else {
                    *reinterpret_cast<f32x4*>(Xs + loff[i]) = v;
                }
            }
        };

        f32x4 pa[ITEMS], pb[ITEMS];
        unsigned oka = 0, okb = 0;
        pload(0, pa, oka);
        if (nunits > 1) pload(1, pb, okb);
        stamp(2);                                      // first loads issued
        setup_shared();
        pcommit(0, pa, oka, Xs0);
        stamp(3);                                      // unit 0 committed
        __syncthreads();                               // unit 0 staged
        stamp(4);
        for (int u = 0; u < nunits; u += 2) {
            // consumers multiply unit u (buffer 0): stage unit u+1 into buffer 1, fetch unit u+2
            if (u + 1 < nunits) {
                if (u + 2 < nunits) pload(u + 2, pa, oka);
                pcommit(u + 1, pb, okb, Xs0 + bufsz);
            }
            stamp(5);                                  // producer: staged the next unit
            __syncthreads();                           // end of unit u
            stamp(6);
            if (u + 1 >= nunits) break;
            // consumers multiply unit u+1 (buffer 1): stage unit u+2 into buffer 0, fetch unit u+3
            if (u + 2 < nunits) {
                if (u + 3 < nunits) pload(u + 3, pb, okb);
                pcommit(u + 2, pa, oka, Xs0);
            }
            stamp(5);
            __syncthreads();                           // end of unit u+1
            stamp(6);
        }
    } else {
        // ================================ CONSUMER WAVES ================================
        constexpr bool POLY = (MODE == MODE_POLY);
        f32x4 acc[(POLY || WINO || DEC2) ? 1 : NW][MW];
        f32x4 acc3[3][POLY ? NW : 1][MW];              // polyphase: a / z / c accumulator sets
        f32x4 acc4[4][WINO ? NW : 1][MW];              // Winograd: m0..m3
        f32x4 acc2[2][DEC2 ? NW : 1][MW];              // fused decimating convs: k=3 / 1x1
        float s1[MW], s2[MW];
        UnitWeightStream<MW, NSTEPS> wst;
        wst.init(p.w + (long)sig * p.w_sig + (long)(active ? mg : 0) * p.Q * 64 * MW,
                 (FASTSVC_DBG_ON(p, DBG_NO_WEIGHTS)) ? NSTEPS : p.Q, lane);
        const int colbase = (lane >> 4) * XS + (lane & 15) + wave_n * (NW * 16) + (halo_al - halo);
        // Winograd: pair index of this lane in M-tile 0 -> (plane position, residue) -> plane r / r+D
        const int wpair = wave_n * (NW * 16) + (lane & 15);
        const int wq = (lane >> 4) * XS + wpair / S + 4 / S;
        const int colr = wq + (wpair % S) * p.ps;
        const int colrd = wq + (wpair % S + S) * p.ps;
        constexpr bool EST = ws_estage<MW, NW, MODE, EPI>();
        // this wave's epilogue-operand slots, behind the window buffers (launch_conv_pipe sizes them)
        float* Ew = Xs0 + ((p.nchunks > 1 || p.tpw > 1) ? 2 : 1) * bufsz
                        + cw * ((EPI == EPI_RES ? 1 : p.res ? 3 : 2) * MW * NW * EST_ITEM_FLOATS);
        EpiRsrc R;
        {
            const long ct = (long)p.COUT * p.ldy;       // rows of y / y2 / res / scale / shift at the output pitch
            const float* nul = p.bias;                  // any valid address for unused descriptors
            R.y = act_rsrc(p.y ? p.y : nul, p.y ? (long)sig * p.y_sig + (long)b * p.y_b : 0, p.y ? ct : 0);
            const bool has_y2 = DEC2 || (flags & F_AFF_OUT);
            R.y2 = act_rsrc(has_y2 ? p.y2 : nul, has_y2 ? (long)sig * p.y2_sig + (long)b * p.y2_b : 0, has_y2 ? ct : 0);
            R.res = act_rsrc(p.res ? p.res : nul, p.res ? (long)sig * p.res_sig + (long)b * p.res_b : 0, p.res ? ct : 0);
            const bool has_ss = (flags & (F_STATS | F_AFF_OUT)) != 0;
            R.ss = act_rsrc(has_ss ? p.ss_out : nul, has_ss ? (long)b * p.ss_out_b : 0, has_ss ? 2 * ct : 0);
            R.r1x = make_rsrc(p.r1x ? p.r1x + (long)sig * p.r1x_sig + (long)b * p.r1x_b : nul, p.r1x ? p.ldy : 0);
        }
        // (variants already at their register budget fetch them per tile instead)
#ifdef FASTSVC_ACT_BF16
        constexpr bool HOIST_OK = false;        // the bf16 pack / unpack temporaries take those registers
#else
        constexpr bool HOIST_OK = true;
#endif
        constexpr bool HOIST = HOIST_OK && !(WINO && ((MW == 3 && NW == 2) || EPI == EPI_RANK1)) &&
                               !(MODE == MODE_STRETCH && NW == 4) && !(MW == 3 && EPI == EPI_RANK1) &&
                               !(POLY && S == 5 && EPI == EPI_AFF) &&
                               !(MW == 2 && NW == 4 && (EPI == EPI_RANK1 || EPI == EPI_AFF)) && !(DEC2 && MW == 2 && NW == 2);
        float k_bias[MW], k_bias2[MW], k_r1w[MW], k_r1b[MW];
        auto load_consts = [&]() {
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int cot = (mg * MW + m) * 16 + (lane & 15);
            const int co = cot < p.COUT ? cot : 0;
            const bool cok = active && cot < p.COUT;
            k_bias[m] = cok ? p.bias[(long)sig * p.bias_sig + co] : 0.f;
            k_bias2[m] = 0.f; k_r1w[m] = 0.f; k_r1b[m] = 0.f;
            if constexpr (DEC2) k_bias2[m] = cok ? p.bias2[(long)sig * p.bias2_sig + co] : 0.f;
            if constexpr (EPI == EPI_RANK1) {
                k_r1w[m] = cok ? p.r1w[(long)sig * p.r1_sig + co] : 0.f;
                k_r1b[m] = cok ? p.r1b[(long)sig * p.r1_sig + co] : 0.f;
            }
        }
        };
        if constexpr (HOIST) load_consts();
        const EpiConst<MW, HOIST> K{k_bias, k_bias2, k_r1w, k_r1b, nullptr, 0};      // (no operand scales in this family)
        stamp(2);                                      // weight stream issued
        setup_shared();
        __syncthreads();                               // unit 0 staged
        stamp(4);
        int u = 0;
        for (int tl = 0; tl < ntiles; ++tl) {
            if constexpr (DEC2) {
                #pragma unroll
                for (int k = 0; k < 2; ++k)
                    #pragma unroll
                    for (int n = 0; n < NW; ++n)
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) acc2[k][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if constexpr (WINO) {
                #pragma unroll
                for (int k = 0; k < 4; ++k)
                    #pragma unroll
                    for (int n = 0; n < NW; ++n)
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) acc4[k][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if constexpr (POLY) {
                #pragma unroll
                for (int k = 0; k < 3; ++k)
                    #pragma unroll
                    for (int n = 0; n < NW; ++n)
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) acc3[k][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                #pragma unroll
                for (int n = 0; n < NW; ++n)
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int ch = 0; ch < p.nchunks; ++ch, ++u) {
                if constexpr (EST) {
                    // one unit earlier when the tile has several K chunks: more time to land
                    if (active && ch == max(p.nchunks - 2, 0) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE)))
                        ws_epilogue_stage<MW, NW, EPI>(p, R, Ew, mg, (tile0 + tl) * NT + wave_n * (NW * 16), lane);
                }
                if (active && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA))) {
                    constexpr bool RL = !WSTATIC;      // WSTATIC: the ring holds the whole layer, nothing to re-request
                    if constexpr (DEC2) mfma_unit_dec2<MW, NW, RL>(acc2, Xs0 + (u & 1) * bufsz + colbase, XS, wst);
                    else if constexpr (WINO) mfma_unit_wino<MW, NW, S, NSTEPS, RL>(acc4, Xs0 + (u & 1) * bufsz + colr, Xs0 + (u & 1) * bufsz + colrd, XS, wst);
                    else if constexpr (POLY) mfma_unit_poly<MW, NW, RL>(acc3, Xs0 + (u & 1) * bufsz + colbase, XS, wst);
                    else mfma_unit<MW, NW, NSTEPS, RL>(acc, Xs0 + (u & 1) * bufsz + colbase, XS, wst, p.dil);
                }
                stamp(7);                              // consumer: MFMAs of the unit issued
                if (ch + 1 == p.nchunks) {
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                    if constexpr (DEC2)
                        ws_epilogue_dec2<MW, NW>(p, R, acc2, sig, mg,
                                                 (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
                    else if constexpr (WINO)
                        ws_epilogue_wino<MW, NW, EPI, S>(p, R, acc4, sig, mg,
                                                         (tile0 + tl) * NT + wave_n * (NW * 32), active, lane, K);
                    else if constexpr (POLY)
                        ws_epilogue_poly<MW, NW, EPI, S>(p, R, acc3, s1, s2, mg,
                                                         (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
                    else if constexpr (EPI == EPI_GENERIC)
                        ws_epilogue_tile<MW, NW>(p, R, acc, s1, s2, sig, mg,
                                                 (tile0 + tl) * NT + wave_n * (NW * 16), active, lane);
                    else {
                        if constexpr (EST) { if (active) ws_epilogue_stage_wait<NSTEPS>(!WSTATIC && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA))); }
                        ws_epilogue_kind<MW, NW, EPI, EST>(p, R, acc, s1, s2, sig, mg,
                                                           (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K, Ew);
                    }
                    if constexpr (TRACKS) { if (p.amax_out) amax_tile_flush(R); }
                    if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
                        // fp32 partials stay short (this tile only); the running sums are f64 in LDS
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) {
                            float a1 = s1[m], a2 = s2[m];
                            a1 = row_xsum(a1); a2 = row_xsum(a2);
                            if (active && lane < 16) {
                                const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                                atomicAdd(&sstat[slot + 0], (double)a1);
                                atomicAdd(&sstat[slot + 1], (double)a2);
                            }
                        }
                    }
                }
                stamp(8);                              // consumer: epilogue (if any) issued
                __syncthreads();                       // end of unit u
                stamp(6);
            }
        }
        if constexpr (TRACKS) amax_flush(p, R, &s_amax, &s_cnt, 4, sig, b, lane, blockIdx.x);   // (float32 storage: the next conv's split-binary16 scale)
    }
    // one f64 global atomic per channel per workgroup
    if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += 512) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}

#ifndef FASTSVC_ACT_BF16
template <int MW, int NW>
static hipError_t launch_conv_generic(const ConvParams& p, int nsig, hipStream_t stream) {
    constexpr int WM = 1, WN = 4;
    constexpr int NT = 16 * NW * WN;
    dim3 grid((p.T + NT - 1) / NT, (p.ngroups + WM - 1) / WM, nsig * p.B);
    dim3 block(64 * WM * WN);
    const int CINp = p.nchunks * p.KC;
    const size_t smem = sizeof(double) * 2 * 16 * MW * WM + sizeof(float) * (3 * (size_t)CINp + (size_t)p.KC * p.xs);
    hipLaunchKernelGGL((conv_mfma_kernel<MW, NW, WM, WN>), grid, block, smem, stream, p);
    return hipGetLastError();
}

#endif   // generic kernel: float32 storage only

// tile shapes the polyphase variant is compiled for: three accumulator sets, so NW * MW <= 4
template <int MW, int NW, int WM, int WN>
constexpr bool poly_shape() { return WM != 4 && ((MW == 3 && NW == 1) || (MW == 2 && NW <= 2)); }

#ifndef FASTSVC_ACT_BF16      // storage-independent host queries: defined once
int conv_ws_resident(int MW, int NW, int mode, int epi_kind) {
    // workgroups per CU the register budget of the compiled variant allows (see ws_min_waves)
    if (mode == MODE_POLY) return (MW <= 2 && NW == 1) ? 2 : 1;
    if (mode == MODE_WINO) return (MW == 2 && NW == 1) ? 2 : 1;
    if (mode == MODE_DEC2) return MW <= 2 ? 2 : 1;
    if (MW <= 2 || NW == 1) return 2;
    if (NW == 2 && mode == MODE_DIRECT && (epi_kind == EPI_PLAIN || epi_kind == EPI_RES)) return 2;
    return 1;
}

bool conv_ws_tail_ok(int MW, int NW, int mode, int epi_kind, int S) {
    // variants compiled with the row-end (T % 4 != 0) handling, see ws_tail_ok
    if (mode == MODE_POLY) return !(MW == 2 && S >= 4 && epi_kind == EPI_AFF);
    return !(MW == 2 && NW == 4);
}

bool conv_poly_shape(int MW, int NW, int WM, int WN) {
    return WM != 4 && ((MW == 3 && NW == 1) || (MW == 2 && NW <= 2)) && WM * WN == 4;
}

#endif

template <auto KERNEL>
static hipError_t launch_instance(dim3 grid, dim3 block, size_t smem, hipStream_t stream, const ConvParams& p) {
    if (smem > 64 * 1024) {                            // above the default dynamic-LDS limit: once per instance
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);   // (+ a few static bytes)
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(KERNEL, grid, block, smem, stream, p);
    return hipGetLastError();
}

// One launch of the pipelined kernel: picks the resident-weights instance (MW == 2 and a single unit of
// weights per channel group) and adds the epilogue-operand slots of the ws_estage variants to the LDS size.
template <int MW, int NW, int WM, int WN, int MODE, int NTAPS, int EPI = EPI_GENERIC, int S = 1>
static hipError_t launch_ws(dim3 grid, dim3 block, size_t smem, hipStream_t stream, const ConvParams& p) {
    if constexpr (ws_estage<MW, NW, MODE, EPI>()) {
        const bool aff_epi = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
        smem += sizeof(float) * 4 * (size_t)(aff_epi ? (p.res ? 3 : 2) : 1) * MW * NW * EST_ITEM_FLOATS;
    }
    constexpr int NSTEPS = (MODE == MODE_WINO || MODE == MODE_DEC2) ? 24 : 6 * NTAPS;
    if constexpr (MW == 2) {
        if (p.Q == NSTEPS && !(FASTSVC_DBG_ON(p, DBG_NO_WEIGHTS)))
            return launch_instance<&conv_mfma_ws_kernel<MW, NW, WM, WN, MODE, NTAPS, EPI, S, true>>(grid, block, smem, stream, p);
    }
    return launch_instance<&conv_mfma_ws_kernel<MW, NW, WM, WN, MODE, NTAPS, EPI, S, false>>(grid, block, smem, stream, p);
}

template <int MW, int NW, int WM, int WN>
static hipError_t launch_conv_pipe(const ConvParams& p, int nsig, hipStream_t stream) {
    const int NT = (p.mode == MODE_WINO ? 32 : 16) * NW * WN;
    const int ntx = (p.T + NT - 1) / NT;
    const int tpw = p.tpw > 0 ? p.tpw : 1;
    dim3 grid((ntx + tpw - 1) / tpw, (p.ngroups + WM - 1) / WM, nsig * p.B);
    dim3 block(64 * WM * WN);
    const int CINp = p.nchunks * p.KC;
    const int nbuf = (p.nchunks > 1 || tpw > 1) ? 2 : 1;
    const size_t smem = sizeof(double) * 2 * 16 * MW * WM
                      + sizeof(float) * (2 * (size_t)CINp + (size_t)nbuf * p.KC * p.xs);
    block = dim3(512);                                  // 4 consumer + 4 producer waves
    if (p.mode == MODE_WINO) {
        if constexpr ((MW == 3 && NW <= 2) || (MW == 2 && NW == 1)) {
            const bool res = p.res != nullptr;
            if constexpr (MW == 2) {                        // rank-1 residual: the stage-0 chain (C_in = 1 residual path)
                if (p.r1x) {
                    if (p.dil == 4) return launch_ws<MW, NW, WM, WN, MODE_WINO, 3, EPI_RANK1, 4>(grid, block, smem, stream, p);
                    else if (p.dil == 2) return launch_ws<MW, NW, WM, WN, MODE_WINO, 3, EPI_RANK1, 2>(grid, block, smem, stream, p);
                    else if (p.dil == 1) return launch_ws<MW, NW, WM, WN, MODE_WINO, 3, EPI_RANK1, 1>(grid, block, smem, stream, p);
                    else return hipErrorInvalidValue;
                    return hipGetLastError();
                }
            } else if (p.r1x) {
                return hipErrorInvalidValue;
            }
#define FASTSVC_WINO(dv) \
            if (p.dil == dv) { \
                if (res) return launch_ws<MW, NW, WM, WN, MODE_WINO, 3, EPI_RES, dv>(grid, block, smem, stream, p); \
                else return launch_ws<MW, NW, WM, WN, MODE_WINO, 3, EPI_PLAIN, dv>(grid, block, smem, stream, p); \
                return hipGetLastError(); \
            }
            FASTSVC_WINO(1) FASTSVC_WINO(2) FASTSVC_WINO(4)
#undef FASTSVC_WINO
        }
        return hipErrorInvalidValue;
    }
    if constexpr (MW == 2 && WM != 1) {
        return hipErrorInvalidValue;                        // (2,1,2,2) / (2,1,4,1) exist for Winograd only
    } else
    if (p.mode == MODE_DEC2) {
        if constexpr (NW <= 2) {
            return launch_ws<MW, NW, WM, WN, MODE_DEC2, 3>(grid, block, smem, stream, p);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    } else
    if (p.mode == MODE_POLY) {
        if constexpr (poly_shape<MW, NW, WM, WN>()) {
            const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
#define FASTSVC_POLY(sv) \
            if (p.s == sv) { \
                if (aff) return launch_ws<MW, NW, WM, WN, MODE_POLY, 3, EPI_AFF, sv>(grid, block, smem, stream, p); \
                else return launch_ws<MW, NW, WM, WN, MODE_POLY, 3, EPI_PLAIN, sv>(grid, block, smem, stream, p); \
                return hipGetLastError(); \
            }
            FASTSVC_POLY(2) FASTSVC_POLY(4) FASTSVC_POLY(5)
#undef FASTSVC_POLY
        }
        return hipErrorInvalidValue;
    } else if (p.ntaps == 1) {
        return launch_ws<MW, NW, WM, WN, MODE_DECIMATE, 1>(grid, block, smem, stream, p);
    } else if (p.ntaps == 3 && (p.mode == MODE_STRETCH || p.mode == MODE_DIRECT)) {
        // compile-time specialised epilogue
        const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
        const int kind = aff ? EPI_AFF : p.r1x ? EPI_RANK1 : p.res ? EPI_RES : EPI_PLAIN;
        {
#define FASTSVC_EPI(mode, k) return launch_ws<MW, NW, WM, WN, mode, 3, k>(grid, block, smem, stream, p)
            if (p.mode == MODE_STRETCH) {
                if (kind == EPI_AFF) FASTSVC_EPI(MODE_STRETCH, EPI_AFF); else FASTSVC_EPI(MODE_STRETCH, EPI_PLAIN);
            } else {
                if (kind == EPI_AFF) FASTSVC_EPI(MODE_DIRECT, EPI_AFF);
                else if (kind == EPI_RANK1) FASTSVC_EPI(MODE_DIRECT, EPI_RANK1);
                else if (kind == EPI_RES) FASTSVC_EPI(MODE_DIRECT, EPI_RES);
                else FASTSVC_EPI(MODE_DIRECT, EPI_PLAIN);
            }
#undef FASTSVC_EPI
        }
    } else if (p.mode == MODE_STRETCH) {
        return launch_ws<MW, NW, WM, WN, MODE_STRETCH, 3>(grid, block, smem, stream, p);
    } else if (p.mode == MODE_DECIMATE) {
        return launch_ws<MW, NW, WM, WN, MODE_DECIMATE, 3>(grid, block, smem, stream, p);
    } else {
        return launch_ws<MW, NW, WM, WN, MODE_DIRECT, 3>(grid, block, smem, stream, p);
    }
    return hipGetLastError();
}

#ifndef FASTSVC_ACT_BF16
bool conv_pipe_supported(const ConvParams& p) {
    if (p.KC != 24) return false;                              // 6 k-steps per tap per chunk, compiled in
    if (p.flags & F_PRE_AFFINE) return false;                  // only the generic kernel fuses the affine
    if (p.mode == MODE_DEC2) return p.ntaps == 3 && p.dil == 1 && p.bias2 && p.y2 && !p.res && !p.r1x && p.flags == 0;
    if (p.ntaps == 1) return p.mode == MODE_DECIMATE;          // the 1x1 residual convs of the down nets
    if (p.ntaps != 3) return false;
    if (p.mode == MODE_DIRECT) return true;                    // any row length (element masks at the row end)
    if (p.mode == MODE_WINO)                                   // F(2,3) along time: plain / residual epilogue only
        return p.T == p.x_T && (p.dil == 1 || p.dil == 2 || p.dil == 4) &&
               !(p.flags & (F_STATS | F_AFF_OUT)) && p.ps > 0;
    if (p.mode == MODE_POLY)                                   // input-rate tiles, float4 window loads
        return p.T == p.x_T && p.dil == 1 && (p.s == 2 || p.s == 4 || p.s == 5) &&
               !p.res && !p.r1x && !(p.flags & F_PRE_NORM);
    return p.mode == MODE_STRETCH || p.mode == MODE_DECIMATE;
}

#endif

hipError_t launch_conv(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
    if (cfg.pipe) {
#define FASTSVC_PIPE(mw, nw, wm, wn) \
        if (cfg.MW == mw && cfg.NW == nw && cfg.WM == wm && cfg.WN == wn) return launch_conv_pipe<mw, nw, wm, wn>(p, cfg.nsig, stream);
        FASTSVC_PIPE(2, 4, 1, 4) FASTSVC_PIPE(2, 2, 1, 4) FASTSVC_PIPE(2, 1, 1, 4)
        FASTSVC_PIPE(2, 1, 2, 2) FASTSVC_PIPE(2, 1, 4, 1)
        FASTSVC_PIPE(3, 4, 1, 4) FASTSVC_PIPE(3, 2, 1, 4) FASTSVC_PIPE(3, 1, 1, 4)
        FASTSVC_PIPE(3, 4, 2, 2) FASTSVC_PIPE(3, 2, 2, 2) FASTSVC_PIPE(3, 1, 2, 2)
        FASTSVC_PIPE(3, 4, 4, 1) FASTSVC_PIPE(3, 2, 4, 1)
#undef FASTSVC_PIPE
        return hipErrorInvalidValue;
    }
#ifndef FASTSVC_ACT_BF16
#define FASTSVC_CASE(mw, nw) if (cfg.MW == mw && cfg.NW == nw) return launch_conv_generic<mw, nw>(p, cfg.nsig, stream);
    FASTSVC_CASE(1, 1) FASTSVC_CASE(1, 2) FASTSVC_CASE(1, 4)
    FASTSVC_CASE(2, 1) FASTSVC_CASE(2, 2) FASTSVC_CASE(2, 4)
    FASTSVC_CASE(3, 1) FASTSVC_CASE(3, 2) FASTSVC_CASE(3, 4)
#undef FASTSVC_CASE
#endif
    return hipErrorInvalidValue;      // (the generic kernel exists for float32 storage only)
}

// ---------------------------------------------------------------------------------------------
// Down-sampling stage 0, first conv: C_in = 1 (fastsvc.py:173, downsample_block.2 of net 0).
// K = 3 only: a VALU kernel; every thread produces 4 consecutive samples for all C channels.
// HBM-write bound (C floats written per float read).
// ---------------------------------------------------------------------------------------------
#ifdef FASTSVC_ACT_BF16
constexpr int IN1_SPT = 8;          // samples per thread: 16-byte bf16 stores
#else
constexpr int IN1_SPT = 4;          // 16-byte float stores
#endif
__global__ __launch_bounds__(256)
void in1_conv_kernel(const float* __restrict__ x, long x_sig, const float* __restrict__ w,
                     const float* __restrict__ bias, long w_sig, long b_sig, float* __restrict__ y,
                     int B, int C, int ld, const int* __restrict__ lens, int len_mul, float* __restrict__ amax_out) {
    constexpr int SPT = IN1_SPT;
    const int z = blockIdx.z;
    const int sig = z / B;
    const int T = lens ? lens[z - sig * B] * len_mul : ld;          // valid length of this utterance (pitch ld)
    const int t = (blockIdx.x * 256 + threadIdx.x) * SPT;
    if (blockIdx.x * 256 * SPT >= T) return;                        // (whole workgroup; lanes past T store nothing)
    float amx = 0.f;
    const float* xr = x + (long)sig * x_sig + (long)(z - sig * B) * ld;      // the two signals are separate tensors
    float xv[SPT + 2];
    #pragma unroll
    for (int i = 0; i < SPT + 2; ++i) {
        const int tt = t - 1 + i;
        xv[i] = (tt >= 0 && tt < T) ? lrelu(xr[tt]) : 0.f;
    }
    const float* ws = w + sig * w_sig;
    const float* bs = bias + sig * b_sig;
    act_t* yb = reinterpret_cast<act_t*>(y) + (long)z * C * ld;
    const bool full = (t + SPT - 1 < T) && ((ld & (SPT - 1)) == 0);
    for (int co = 0; co < C; ++co) {
        const float w0 = ws[co * 3 + 0], w1 = ws[co * 3 + 1], w2 = ws[co * 3 + 2], bb = bs[co];
        float o[SPT];
        #pragma unroll
        for (int i = 0; i < SPT; ++i) { o[i] = bb + (w0 * xv[i] + w1 * xv[i + 1]) + w2 * xv[i + 2]; amx = fmaxf(amx, fabsf(o[i])); }
        act_t* yr = yb + (long)co * ld + t;
#ifdef FASTSVC_ACT_BF16
        if (full) {
            u32x4 q;
            q.x = bf16_pack2(o[0], o[1]);
            q.y = bf16_pack2(o[2], o[3]);
            q.z = bf16_pack2(o[4], o[5]);
            q.w = bf16_pack2(o[6], o[7]);
            *reinterpret_cast<u32x4*>(yr) = q;
        } else {
            for (int i = 0; i < SPT && t + i < T; ++i) yr[i] = (act_t)f32_to_bf16_bits(o[i]);
        }
#else
        if (full) {
            *reinterpret_cast<f32x4*>(yr) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
            for (int i = 0; i < SPT && t + i < T; ++i) yr[i] = o[i];
        }
#endif
    }
    if (amax_out) {                                                 // largest |c1| (lanes past T: bias-sized values)
        const unsigned a = wave_max_u32_lane63(__builtin_bit_cast(unsigned, amx));
        if ((threadIdx.x & 63) == 63) atomicMax(reinterpret_cast<unsigned*>(amax_out) + z * AMAX_ENTRY + ((blockIdx.x + (threadIdx.x >> 6)) & (AMAX_W - 1)) * AMAX_STRIDE, a);
    }
}

hipError_t launch_in1_conv(const float* x, long x_sig, const float* w, const float* bias, long w_sig, long b_sig,
                           float* y, int nsig, int B, int C, int T, const int* lens, int len_mul, hipStream_t stream,
                           float* amax_out) {
    dim3 grid((T + 256 * IN1_SPT - 1) / (256 * IN1_SPT), 1, nsig * B);
    hipLaunchKernelGGL(in1_conv_kernel, grid, dim3(256), 0, stream, x, x_sig, w, bias, w_sig, b_sig, y, B, C, T, lens, len_mul, amax_out);
    return hipGetLastError();
}

#ifndef FASTSVC_ACT_BF16
// ---------------------------------------------------------------------------------------------
// Largest magnitude of every input row (fastsvc_kernels.h, launch_amax_inputs): what the split-binary16 kernels
// scale their staged activations by.  gridDim = (AMAX_W, 3 B input rows + zeroing rows): block x of an input row scans
// the x-th part of it and STORES its max into slot x of the row's entry (no atomics, nothing to zero first); the rows
// behind them zero the intermediate tensors' entries, which later kernels of the forward accumulate into.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float amax_span(const float* __restrict__ x, long i0, long i1, int tid) {
    float a = 0.f;
    if ((reinterpret_cast<size_t>(x + i0) & 15) == 0) {
        long i = i0 + 4 * tid;
        for (; i + 3 * 1024 + 3 < i1; i += 4096) {          // four independent 16-byte loads in flight per thread
            f32x4 v[4];
            #pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(x + i + 1024 * k);
            #pragma unroll
            for (int k = 0; k < 4; ++k) a = fmaxf(fmaxf(a, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
        }
        for (; i + 3 < i1; i += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
            a = fmaxf(fmaxf(a, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (long j = i0 + ((i1 - i0) & ~3L) + tid; j < i1; j += 256) a = fmaxf(a, fabsf(x[j]));
    } else {
        for (long j = i0 + tid; j < i1; j += 256) a = fmaxf(a, fabsf(x[j]));
    }
    return a;
}

__global__ __launch_bounds__(256)
void amax_inputs_kernel(const float* __restrict__ sig, long sig_stride, const float* __restrict__ ppg, int B, int C, int F,
                        int hop, const int* __restrict__ lens, float* __restrict__ amax_in, float* __restrict__ zero, int nzero) {
    const int row = blockIdx.y;
    const int part = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ float red[4];
    if (row >= 3 * B) {
        for (int i = ((row - 3 * B) * AMAX_W + part) * 1024 + 4 * tid; i < nzero; i += (gridDim.y - 3 * B) * AMAX_W * 1024)
            *reinterpret_cast<f32x4*>(zero + i) = f32x4{0.f, 0.f, 0.f, 0.f};      // (nzero is a multiple of 4)
        return;
    }
    float a = 0.f;
    if (row < 2 * B) {
        const int s = row / B, b = row - s * B;
        const long n = (long)(lens ? lens[b] : F) * hop;
        const long per = ((n + AMAX_W - 1) / AMAX_W + 3) & ~3L;
        a = amax_span(sig + (long)s * sig_stride + (long)b * F * hop, min(n, part * per), min(n, (part + 1) * per), tid);
    } else {
        const int b = row - 2 * B;
        const int nf = lens ? lens[b] : F;
        const float* x = ppg + (long)b * C * F;
        if (nf == F) {
            const long n = (long)C * F;
            const long per = ((n + AMAX_W - 1) / AMAX_W + 3) & ~3L;
            a = amax_span(x, min(n, part * per), min(n, (part + 1) * per), tid);
        } else {
            for (int c = part; c < C; c += AMAX_W)                   // ragged: the valid frames of each channel row
                for (int i = tid; i < nf; i += 256) a = fmaxf(a, fabsf(x[(long)c * F + i]));
        }
    }
    const unsigned m = wave_max_u32_lane63(__builtin_bit_cast(unsigned, a));
    if ((tid & 63) == 63) red[tid >> 6] = __builtin_bit_cast(float, m);
    __syncthreads();
    if (tid == 0) amax_in[row * AMAX_ENTRY + part * AMAX_STRIDE] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// InstanceNorm sums of a FiLM-affined tensor, EXACT: sum u and sum u^2 over each row's own length with every element
// and every product in float64 (a float32 squared is exact in float64).  The conv epilogues accumulate these sums per
// lane in float32 - fine for real rows, not for a row that is nearly constant (var << 1e-5 mean^2: the cancellation in
// E[u^2] - mean^2 eats the partial sums' last bits), which is what a 1-frame utterance's two samples at the first block
// are.  Launched behind the producing conv for batches of at most 4 frames (fastsvc_plan.cpp, g_exact_f32) and
// OVERWRITES the (b, c) entries that conv accumulated.  u: (B, C, ld) float32; grid (C, B), one wave per row.
__global__ __launch_bounds__(64)
void stats_exact_kernel(const float* __restrict__ u, double* __restrict__ st, int C, int ld, const int* __restrict__ lens, int len_mul,
                        int max_frames) {
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (lens && lens[b] > max_frames) return;            // (a ragged batch: only its SHORT utterances need this)
    const int T = lens ? lens[b] * len_mul : ld;
    const float* row = u + ((long)b * C + c) * ld;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < T; t += 64) { const double v = (double)row[t]; s1 += v; s2 += v * v; }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
    if (lane == 0) { st[((long)b * C + c) * 2 + 0] = s1; st[((long)b * C + c) * 2 + 1] = s2; }
}
hipError_t launch_stats_exact(const float* u, double* st, int B, int C, int ld, const int* lens, int len_mul, int max_frames,
                              hipStream_t stream) {
    hipLaunchKernelGGL(stats_exact_kernel, dim3((unsigned)C, (unsigned)B), dim3(64), 0, stream, u, st, C, ld, lens, len_mul, max_frames);
    return hipGetLastError();
}

// an empty launch: what fastsvc_plan.cpp times a fork / join between two streams with (ExecCtx calibration)
__global__ void noop_kernel() {}
hipError_t launch_noop(hipStream_t stream) {
    hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

hipError_t launch_amax_inputs(const float* sig, long sig_stride, const float* ppg, int B, int C, int F, int hop,
                              const int* lens, float* amax_in, float* zero, int nzero, hipStream_t stream) {
    const int zrows = (nzero + AMAX_W * 1024 * 4 - 1) / (AMAX_W * 1024 * 4);
    hipLaunchKernelGGL(amax_inputs_kernel, dim3(AMAX_W, 3 * B + (zrows < 1 ? 1 : zrows)), dim3(256), 0, stream,
                       sig, sig_stride, ppg, B, C, F, hop, lens, amax_in, zero, nzero);
    return hipGetLastError();
}
#endif

// ---------------------------------------------------------------------------------------------
// conv_last: 1x1 conv C -> O (fastsvc.py:301,330), HBM-read bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void pointwise_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                          const float* __restrict__ bias, float* __restrict__ y,
                          int C, int O, int ld, const int* __restrict__ lens, int len_mul) {
    const int b = blockIdx.z;
    const int T = lens ? lens[b] * len_mul : ld;          // valid length of this utterance (pitch ld)
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= ld) return;
    if (t >= T) {                                         // ragged batch: the padding of the output is zero
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < 4 && t + i < ld; ++i) y[((long)b * O + o) * ld + t + i] = 0.f;
        return;
    }
    const act_t* xb = reinterpret_cast<const act_t*>(x) + (long)b * C * ld;
    const bool full = (t + 3 < T) && ((ld & 3) == 0);
    auto ld1 = [&](long idx) -> float {
#ifdef FASTSVC_ACT_BF16
        return __builtin_bit_cast(float, (unsigned)xb[idx] << 16);
#else
        return xb[idx];
#endif
    };
    auto ld4 = [&](long idx) -> f32x4 {
#ifdef FASTSVC_ACT_BF16
        const u32x2v v = *reinterpret_cast<const u32x2v*>(xb + idx);
        return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
#else
        return *reinterpret_cast<const f32x4*>(xb + idx);
#endif
    };
    for (int o = 0; o < O; ++o) {
        f32x4 acc = f32x4{bias[o], bias[o], bias[o], bias[o]};
        if (full) {
            for (int c = 0; c < C; ++c)
                acc += ld4((long)c * ld + t) * w[o * C + c];
            *reinterpret_cast<f32x4*>(y + ((long)b * O + o) * ld + t) = acc;
        } else {
            for (int i = 0; i < 4 && t + i < ld; ++i) {
                float a = 0.f;
                if (t + i < T) {
                    a = bias[o];
                    for (int c = 0; c < C; ++c) a += ld1((long)c * ld + t + i) * w[o * C + c];
                }
                y[((long)b * O + o) * ld + t + i] = a;
            }
        }
    }
}

hipError_t launch_pointwise_out(const float* x, const float* w, const float* bias, float* y,
                                int B, int C, int O, int T, const int* lens, int len_mul, hipStream_t stream) {
    dim3 grid((T + 1023) / 1024, 1, B);
    hipLaunchKernelGGL(pointwise_out_kernel, grid, dim3(256), 0, stream, x, w, bias, y, C, O, T, lens, len_mul);
    return hipGetLastError();
}

#ifdef FASTSVC_ACT_BF16
// float32 -> bfloat16 copy of an external input (the PPG) into the workspace
__global__ __launch_bounds__(256)
void act_convert_kernel(const float* __restrict__ src, act_t* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (act_t)f32_to_bf16_bits(src[i]);
}

hipError_t launch_act_convert(const float* src, float* dst, long n, hipStream_t stream) {
    hipLaunchKernelGGL(act_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       src, reinterpret_cast<act_t*>(dst), n);
    return hipGetLastError();
}
#else
// ---------------------------------------------------------------------------------------------
// Speaker bias p = Linear(F.normalize(spk_emb)) for every up block (fastsvc.py:135-137).
// grid = (B, nblocks); one wave per output channel, lanes stride over the embedding.
// ---------------------------------------------------------------------------------------------
struct SpkArgs {
    SpkBlock blk[8];
    int cstart[9];      // prefix sums of the blocks' channel counts
    int nblocks;
};

__global__ __launch_bounds__(256)
void spk_proj_kernel(const float* __restrict__ emb, const SpkArgs args, int E) {
    extern __shared__ __attribute__((aligned(16))) float e_s[];   // [E] normalised embedding
    __shared__ float red[4];
    const int b = blockIdx.x;
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
    const int cglob = blockIdx.y * 4 + wave;            // one wave per output channel
    if (cglob >= args.cstart[args.nblocks]) return;
    int k = 0;
    while (cglob >= args.cstart[k + 1]) ++k;
    const SpkBlock blk = args.blk[k];
    const int c = cglob - args.cstart[k];
    const float* wr = blk.w + (long)c * E;
    float a = 0.f;
    for (int i = lane; i < E; i += 64) a += wr[i] * e_s[i];
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) {
        const float pc = a + blk.bias[c];
        blk.out[(long)b * blk.C + c] = pc;
        // largest |p| of the utterance: with sqrt(T) it bounds a normalised row (scale of the split-binary16 staging)
        if (blk.amax) atomicMax(reinterpret_cast<unsigned*>(blk.amax) + b * AMAX_ENTRY + (c & (AMAX_W - 1)) * AMAX_STRIDE, __builtin_bit_cast(unsigned, fabsf(pc)));
    }
}

hipError_t launch_spk_proj(const float* emb, const SpkBlock* blocks, int nblocks, int B, int E,
                           hipStream_t stream) {
    if (nblocks > 8 || nblocks < 1) return hipErrorInvalidValue;
    SpkArgs args;
    args.nblocks = nblocks;
    args.cstart[0] = 0;
    for (int i = 0; i < 8; ++i) {
        args.blk[i] = blocks[i < nblocks ? i : 0];
        args.cstart[i + 1] = args.cstart[i] + (i < nblocks ? blocks[i].C : 0);
    }
    const int ctot = args.cstart[nblocks];
    hipLaunchKernelGGL(spk_proj_kernel, dim3(B, (ctot + 3) / 4), dim3(256), sizeof(float) * E, stream,
                       emb, args, E);
    return hipGetLastError();
}

#endif   // speaker projection: storage independent, defined once

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
