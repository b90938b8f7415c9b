// fastsvc_cond.hip - gfx950 (CDNA4 / MI355X): a WHOLE conditioning stage of the FastSVC generator as one launch.
//
// Reference dataflow (harana/models/fastsvc.py): FastSVCDownsampleNet.forward (:164-193) for both conditioning
// signals (loudness, sine excitation), then FastSVCFiLMNet.forward (:220-232) of the same stage for both signals,
// summed as FastSVCUpsampleNet._feature_affine sums them (:129-130):
//
//     per signal:  c1 = conv3_d1(lrelu(x)) + b1            x: the stage's input (stage 0: the raw 1-channel signal)
//                  c2 = conv3_d2(lrelu(c1)) + b2
//                  h  = conv3_d4(lrelu(c2)) + b3 + conv1x1(x)
//                  u  = lrelu(conv3_d1(h) + b4)            film.conv
//     both:        [scale ; shift] = conv3_d1([u_lft ; u_sine]) + b5     film.conv_scale / conv_shift, K-concatenated
//
// None of these layers has a reduction over time (the InstanceNorms live in the up blocks), so a time tile of the
// stage's OUTPUT depends on a +-9 column window of its input only: one workgroup walks time tiles of one utterance,
// keeps every intermediate tile (both signals) in LDS in the MFMA operand format and writes only
//     ss   (B, 2C, T)       scale / shift of the stage - what the up block reads, and
//     hd   (2B, C, T / s')  h[..., ::s'] COMPACT - the only part of h the next stage reads (Squeeze2d, upsample.py:53-74)
// i.e. per output column 2C + 2C/s' elements are written and 2 read, where the separate launches move ~6C (c123 writes
// h, the FiLM chain reads it and writes ss, the next stage's decimating pair gathers every line of h again).
//
// Work layout (stage 0, C = 24): 4 waves, two workgroups per CU.  Layers c2 / c3 / film.conv: wave = (signal, every
// second 16-column tile), SWAPPED MFMA operands (A = weights, B = activations), so a lane ends with 4 consecutive
// CHANNELS of one time step = one 8-byte store into the next layer's time-major LDS tile.  Heads: wave = every fourth
// time tile, all three 16-channel output tiles, plain operand order, result staged channel-major in LDS and copied
// out in whole rows (16 B per lane).  The first conv (C_in = 1) runs on the VALU.  All weight fragments of a wave -
// its signal's three layers and the heads: 36 KB-sized fragments = 144 registers - stay resident for the whole
// launch (re-fetched per tile they would be 144 KB of L2 traffic per 23 KB of output).
#include "fastsvc_kernels.h"

namespace fastsvc {
#ifdef FASTSVC_ACT_BF16
namespace bf16 {
#endif

#include "fastsvc_device.inc"

#ifdef FASTSVC_ACT_BF16
typedef __bf16 cs_t;
constexpr int CS_NP = 1;
#else
typedef _Float16 cs_t;
constexpr int CS_NP = 2;
#endif
typedef cs_t cs8 __attribute__((ext_vector_type(8)));
typedef cs_t cs4 __attribute__((ext_vector_type(4)));
typedef unsigned cs_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 cs_mfma(cs8 a, cs8 b, f32x4 c) {
#ifdef FASTSVC_ACT_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

constexpr int CS_ROW = 64;               // bytes of one LDS tile row: 32 channels
constexpr int CS_FRAG = 1024;            // bytes of one packed weight fragment (64 lanes x 8 halves)
constexpr int CS_C = 24;                 // channels of the stage this file is compiled for (fastsvc.yaml mid_channels[-1])
constexpr int CS_NQ = 2;                 // waves per (signal, 16-channel output tile): every CS_NQ-th time tile each
constexpr int CS_NM = 2;                 // 16-channel output tiles of a C -> C layer (C = 24 pads to 32)
constexpr int CS_NWAVES = 2 * CS_NM * CS_NQ;
constexpr int CS_NTHREADS = CS_NWAVES * 64;

// LDS byte offset of 16-byte slot `oct` of tile row `row`: the slot index is XORed with row bits 1..2.  Searched
// exhaustively over the per-row slot permutations (tools: the 4^8 tables indexed by row & 7): ds_read_b128 of a
// fragment (lanes = 16 consecutive rows x 4 slots) is conflict-free at ANY row offset, the 16-byte row stores of the
// staging phases (8 consecutive rows, one slot) are conflict-free, and the 8-byte epilogue stores (16 consecutive rows,
// one half slot) are 2-way - the floor for a 64-byte row pitch; with the swizzle of fastsvc_hx.hip (row pairs swapped,
// slot ^= (row >> 1) & 2) they were 4-way and bank conflicts 35-45 % of this file's LDS cycles (SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int cs_off(int row, int oct) {
    return row * CS_ROW + ((oct ^ ((row >> 1) & 3)) << 4);
}

struct CsFrag { cs8 p[CS_NP]; };
__device__ __forceinline__ CsFrag cs_read(const unsigned char* plane, int off, int lo_off) {
    CsFrag f;
    f.p[0] = *reinterpret_cast<const cs8*>(plane + off);
    if constexpr (CS_NP == 2) f.p[1] = *reinterpret_cast<const cs8*>(plane + lo_off + off);
    return f;
}
struct CsW { u32x4 p[CS_NP]; };
// (through a buffer descriptor: a wave-uniform base in SGPRs + ONE per-lane offset register for every fragment of the
// launch - with per-fragment 64-bit addresses hipcc spilled the address pairs and reloaded each in front of its load,
// `scratch_load; s_waitcnt vmcnt(0); global_load`: the weight stream of a layer ran one memory latency per fragment)
__device__ __forceinline__ CsW cs_wload(const unsigned char* frag, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(frag), 0, CS_NP * CS_FRAG, 0x00020000);
    CsW w;
    #pragma unroll
    for (int q = 0; q < CS_NP; ++q) w.p[q] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, q * CS_FRAG, 0);
    return w;
}
// acc += w (.) a: bf16 one product; split binary16: hi*hi + hi*lo + lo*hi.  SWAP: weights are the A operand.
template <bool SWAP>
__device__ __forceinline__ f32x4 cs_prod(const CsW& w, const CsFrag& a, f32x4 acc) {
    #pragma unroll
    for (int pr = 0; pr < (CS_NP == 2 ? 3 : 1); ++pr) {
        const int ia = (CS_NP == 2 && pr == 2) ? 1 : 0, iw = (CS_NP == 2 && pr == 1) ? 1 : 0;
        if constexpr (SWAP) acc = cs_mfma(__builtin_bit_cast(cs8, w.p[iw]), a.p[ia], acc);
        else acc = cs_mfma(a.p[ia], __builtin_bit_cast(cs8, w.p[iw]), acc);
    }
    return acc;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// LeakyReLU(0.2) = max(v, 0.2 v), the multiply as packed float32 (two values per instruction)
__device__ __forceinline__ f32x4 cs_lrelu4(f32x4 v) {
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const f32x2 ta = a * LRELU_SLOPE, tb = b * LRELU_SLOPE;
    return f32x4{fmaxf(a.x, ta.x), fmaxf(a.y, ta.y), fmaxf(b.x, tb.x), fmaxf(b.y, tb.y)};
}

// Low pieces of a pair of values: lo = f16(e - (float)hi) as ONE mixed-precision FMA each (binary16 source from either
// half of the packed hi register, float32 addend, binary16 result into the low / high half) - bit-identical to convert
// back, subtract, convert (e - hi is exact in float32), 1 instruction per value instead of 2.5 (same as hx_lo_pair of
// fastsvc_hx.hip; the split is 40 % of a float32-storage epilogue's VALU work).
__device__ __forceinline__ unsigned cs_lo_pair(unsigned hi_pair, float e0, float e1) {
#ifdef FASTSVC_ACT_BF16
    (void)hi_pair; (void)e0; (void)e1;
    return 0u;
#else
    unsigned d;
    const float minus1 = -1.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi_pair), "s"(minus1), "v"(e0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi_pair), "s"(minus1), "v"(e1));
    return d;
#endif
}

__device__ __forceinline__ void cs_store4_masked(unsigned char* dst, int lo_off, f32x4 v, unsigned keep) {
    const cs4 h = __builtin_convertvector(v, cs4);
    cs_u2 hp = __builtin_bit_cast(cs_u2, h);
    if constexpr (CS_NP == 2) {
        cs_u2 lp = {cs_lo_pair(hp.x, v[0], v[1]), cs_lo_pair(hp.y, v[2], v[3])};
        lp.x &= keep; lp.y &= keep;
        *reinterpret_cast<cs_u2*>(dst + lo_off) = lp;
    }
    hp.x &= keep; hp.y &= keep;
    *reinterpret_cast<cs_u2*>(dst) = hp;
}

// 8 consecutive channels of one time step -> the 16-byte slot of a time-major tile row (one store per piece)
__device__ __forceinline__ void cs_store8_masked(unsigned char* dst, int lo_off, f32x4 a, f32x4 b, unsigned keep) {
    const cs4 ha = __builtin_convertvector(a, cs4), hb = __builtin_convertvector(b, cs4);
    const cs_u2 pa = __builtin_bit_cast(cs_u2, ha), pb = __builtin_bit_cast(cs_u2, hb);
    *reinterpret_cast<u32x4*>(dst) = u32x4{pa.x & keep, pa.y & keep, pb.x & keep, pb.y & keep};
    if constexpr (CS_NP == 2) {
        *reinterpret_cast<u32x4*>(dst + lo_off) = u32x4{cs_lo_pair(pa.x, a[0], a[1]) & keep, cs_lo_pair(pa.y, a[2], a[3]) & keep,
                                                        cs_lo_pair(pb.x, b[0], b[1]) & keep, cs_lo_pair(pb.y, b[2], b[3]) & keep};
    }
}

// One k=3 layer on this wave's share: signal `sig`, 16-channel output tile `m`, time tiles j = q, q + NQ, ...;
// swapped operands.
//   in:  plane of the layer's input (time-major rows), rows = t - t0 + 16
//   out: tile rows out_row0 + 16 j + (lane & 15), channels 16 m + 4 g .. + 3
// KIND 0: lrelu(acc + b) -> own signal's plane;  1: acc + b + rank-1 residual, raw;
//      2: lrelu(acc + b) -> the 2C-channel plane pair [lft ; sine] the heads read
// The bias rides in the accumulator's initial value.  A wave holds 3 weight fragments per layer: four waves per SIMD
// fit the register file, and their interleaving is what hides the LDS and matrix-pipe latencies here (two waves per
// SIMD with all of a signal's fragments resident ran 51 % of their cycles parked in waits, measured).
template <int NTL, int KIND>
__device__ __forceinline__ void cs_layer(const unsigned char* in_plane, unsigned char* out_planes, int lo_off, int plane_bytes,
                                         const CsW (&W)[3], int dil, int out_row0, const float* kb /* LDS [32] */,
                                         const float* kiv /* LDS [32], float32 storage */,
                                         const float* xs_sig /* LDS, KIND 1 */, const float* r1w_lds /* LDS [32], KIND 1 */,
                                         int sig, int m, int q, int t0, int Tv, int lane) {
    constexpr int NI = NTL / CS_NQ;                    // tiles per wave
    constexpr int G = 4;                               // tiles per group (independent accumulators between dependent products)
    constexpr int NG = NI / G;
    static_assert(NTL % (CS_NQ * G) == 0, "tiles per wave must be a whole number of groups");
    const int l15 = lane & 15, g = lane >> 4;
    int aoff[3];
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap) aoff[tap] = cs_off(out_row0 + (tap - 1) * dil + l15, g) + q * 16 * CS_ROW;
    const int co0 = m * 16 + 4 * g;
    const f32x4 kbv = *reinterpret_cast<const f32x4*>(kb + co0);
    f32x4 r1w = kbv, kivv = kbv;
    if constexpr (KIND == 1) r1w = *reinterpret_cast<const f32x4*>(r1w_lds + co0);
    // float32 storage: the accumulator holds conv * (input tile's scale * weight scale of the channel); one multiply
    // moves it to the output tile's scale (exact: powers of two); bias and residual enter pre-multiplied (kb, r1w)
    if constexpr (CS_NP == 2) kivv = *reinterpret_cast<const f32x4*>(kiv + co0);
    int wbase;
    if constexpr (KIND == 2) {
        // (this signal's channel padding co >= C would land on the other signal's channels: its zeros go to the
        // pair's own padding 2C + 8 .. instead - no divergent store)
        const int cc0 = co0 < CS_C ? sig * CS_C + co0 : 2 * CS_C + 8 + (co0 - CS_C);
        wbase = (cc0 >> 5) * plane_bytes + cs_off(out_row0 + l15, (cc0 & 31) >> 3) + (cc0 & 7) * 2 + q * 16 * CS_ROW;
    } else {
        wbase = sig * plane_bytes + cs_off(out_row0 + l15, co0 >> 3) + (co0 & 7) * 2 + q * 16 * CS_ROW;
    }
    const int tbase = t0 - 16 + out_row0 + l15;        // this lane's time in tile 0
    #pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        f32x4 acc[G];
        #pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            // every accumulator starts from the bias vector - a register written long before the products; the rank-1
            // residual joins AFTER them (see p0_layer_chunk: no VALU result is ever an MFMA's C operand)
            acc[ii] = kbv;
        }
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            #pragma unroll
            for (int ii = 0; ii < G; ++ii) {
                const CsFrag a = cs_read(in_plane, aoff[tap] + (CS_NQ * (grp * G + ii)) * 16 * CS_ROW, lo_off);
                acc[ii] = cs_prod<true>(W[tap], a, acc[ii]);
            }
        #pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int jr = CS_NQ * (grp * G + ii);
            // outside the utterance: the next conv's zero padding - applied to the PACKED values (one AND per register)
            const unsigned keep = (unsigned)(tbase + 16 * (q + jr)) < (unsigned)Tv ? 0xffffffffu : 0u;
            f32x4 v = acc[ii];
            if constexpr (KIND == 1) v = r1w * xs_sig[out_row0 + 16 * (q + jr) + l15] + v;
            if constexpr (CS_NP == 2) v = v * kivv;
            if constexpr (KIND != 1) v = cs_lrelu4(v);
            cs_store4_masked(out_planes + wbase + jr * 16 * CS_ROW, lo_off, v, keep);
        }
    }
}

template <int NTL>
struct CsGeom {
    static constexpr int NTO = NTL - 1;                // output tiles per workgroup tile
    static constexpr int NT = 16 * NTO;                // output columns per workgroup tile
    static constexpr int ROWS = NT + 40;               // LDS rows: row r <-> t = t0 - 16 + r (the deepest read ends at NT + 34)
    static constexpr int PLANE = CS_NP * ROWS * CS_ROW;
    static constexpr int XS_BYTES = 2 * ROWS * 4;
    static constexpr int CONST_FLOATS = 2 * 32 * 4 /* in1 w0 w1 w2 b */ + 2 * 32 /* r1 w */ + 2 * 3 * 2 * 32 /* biases, inverse scales */ + 2 * 64 /* heads */;
    static constexpr int SP = NT * (int)sizeof(act_t) + 16;  // staging pitch of a scale / shift row (bytes; 16-byte aligned rows)
    static constexpr size_t LDS = XS_BYTES + CONST_FLOATS * 4 + 4 * (size_t)PLANE;
    static_assert(2 * CS_C * SP <= 2 * PLANE, "staging rows must fit the planes they alias");
};

template <int NTL>
__global__ __launch_bounds__(CS_NTHREADS, CS_NP == 1 ? 4 : 2)
void cond_stage0_kernel(const CondStage0Params p) {
    using GEO = CsGeom<NTL>;
    constexpr int NT = GEO::NT, ROWS = GEO::ROWS, PLANE = GEO::PLANE, NTO = GEO::NTO;
    constexpr int lo_off = ROWS * CS_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = reinterpret_cast<float*>(smem);                                  // [2][ROWS] raw signal tiles (float32)
    float* kin1 = xs + 2 * ROWS;                                                  // [2][w0 w1 w2 b][32] of the 1 -> C conv
    float* kr1 = kin1 + 2 * 32 * 4;                                               // [2][32]  rank-1 residual weights (1x1 conv, C_in = 1; its bias joins c3's)
    float* kbias = kr1 + 2 * 32;                                                  // [3][2][32]  c2 / c3 / film.conv
    float* kinv = kbias + 3 * 2 * 32;                                             // [3][2][32]  float32 storage: inverse operand scales of those layers
    float* kb5 = kinv + 3 * 2 * 32;                                               // [64] heads' bias
    float* k5inv = kb5 + 64;                                                      // [64] heads' inverse operand scales
    unsigned char* bufA = reinterpret_cast<unsigned char*>(k5inv + 64);           // 2 planes: c1 -> h -> output staging
    unsigned char* bufB = bufA + 2 * PLANE;                                       // 2 planes: c2 -> u
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int Tv = p.lens ? p.lens[b] * p.len_mul : p.T;
    const int ntx = (Tv + NT - 1) / NT;
    const int tpw = p.tpw & 0xffff, dbg = p.tpw >> 16;      // (dbg: developer ablation switches, tools/cond_check.py)
    const int tile_begin = blockIdx.x * tpw;
    const int tile_end = min(tile_begin + tpw, ntx);
    if (tile_begin >= tile_end) return;

    // ---- per-wave weight fragments, resident for the whole launch ----
    // layers: wave = (signal, output channel tile m, time-tile phase q); heads: wave = (output channel tile m5 of 3,
    // half of the output tiles) for waves 0 .. 5 (waves 6, 7 copy the next tile's signal rows meanwhile)
    const int sig = wave & 1, mt = (wave >> 1) & 1, q = wave >> 2;
    CsW W2[3], W3[3], W4[3];
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const long fo = (long)(tap * 2 + mt) * CS_NP * CS_FRAG;
        W2[tap] = cs_wload(reinterpret_cast<const unsigned char*>(p.w[0][sig]) + fo, lane);
        W3[tap] = cs_wload(reinterpret_cast<const unsigned char*>(p.w[1][sig]) + fo, lane);
        W4[tap] = cs_wload(reinterpret_cast<const unsigned char*>(p.w[2][sig]) + fo, lane);
    }
    const int m5 = wave % 3, half5 = wave / 3;
    CsW W5[2][3];                                                                 // heads: [K chunk][tap] of channel tile m5
    #pragma unroll
    for (int ch = 0; ch < 2; ++ch)
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            W5[ch][tap] = cs_wload(reinterpret_cast<const unsigned char*>(p.w5) + (long)((ch * 3 + tap) * 3 + m5) * CS_NP * CS_FRAG, lane);

    // ---- constants (no tile is zeroed: every row / channel a needed output depends on is written before it is read -
    // c1 incl. its channel padding by P1, c2 / h with padding by their layers - and what feeds only the unneeded halo
    // columns of a layer never reaches a stored value: columns are independent in a convolution) ----
    // float32 storage (split-binary16 products): every LDS-resident tensor of utterance b is held times a power of two
    // that puts its BOUND into [2^14, 2^15) - the bound of the raw signal is its measured maximum (amax_in), the bounds of
    // c1, c2, h, u follow through the layers' (l1, bmax): |conv(x)| <= l1 max|x| + bmax (ConvParams, "dynamic range").
    // sc[k][s]: scale of c1, c2, h (per signal) and of u (k = 3: one scale for the channel pair the heads contract over).
    float sc[4][2] = {{1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}};
    if constexpr (CS_NP == 2) {
        float bu = 0.f;
        #pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float ax = amax_read(p.amax_in, s * p.B + b);
            // per-channel bounds alpha[c] * ax + beta[c] (packer), the tile's scale from the largest of them: lane = channel,
            // a wave-wide max (every wave computes the same: no exchange through LDS, no barrier)
            float bt[4];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = lane < CS_C ? p.cbnd[s][(t * 2 + 0) * CS_C + lane] * ax + p.cbnd[s][(t * 2 + 1) * CS_C + lane] : 0.f;
                bt[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, fmaxf(v, 0.f))), 63));
            }
            bu = fmaxf(bu, bt[3]);
            sc[0][s] = hx_scale_for(bt[0]); sc[1][s] = hx_scale_for(bt[1]); sc[2][s] = hx_scale_for(bt[2]);
        }
        sc[3][0] = sc[3][1] = hx_scale_for(bu);
    }
    for (int i = tid; i < 2 * 32; i += CS_NTHREADS) {
        const int s = i >> 5, c = i & 31;
        const bool ok = c < CS_C;
        const float* wp = p.in1_w[s] + (ok ? c : 0) * 3;
        // (the 1 -> C conv runs on the VALU in float32: its taps carry c1's scale)
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) kin1[(s * 4 + tap) * 32 + c] = ok ? wp[tap] * sc[0][s] : 0.f;
        kin1[(s * 4 + 3) * 32 + c] = ok ? p.in1_b[s][c] * sc[0][s] : 0.f;
        #pragma unroll
        for (int l = 0; l < 3; ++l) {
            // accumulator scale of layer l: input tile's scale / inverse weight scale of the channel
            float as = 1.f, os = 1.f;
            if constexpr (CS_NP == 2) {
                const float wi = ok ? p.winv[l][s][c] : 1.f;
                as = sc[l][s] / wi;
                os = sc[l + 1][s];
                kinv[(l * 2 + s) * 32 + c] = ok ? os / as : 0.f;
            }
            // (c3's bias carries the 1x1 residual conv's: both join h)
            kbias[(l * 2 + s) * 32 + c] = ok ? (p.bias[l][s][c] + (l == 1 ? p.r1b[s][c] : 0.f)) * as : 0.f;
            if (l == 1) kr1[s * 32 + c] = ok ? p.r1w[s][c] * as : 0.f;
        }
        {
            const bool ok5 = i < 2 * CS_C;
            float as = 1.f;
            if constexpr (CS_NP == 2) {
                as = sc[3][0] / (ok5 ? p.winv5[i] : 1.f);
                k5inv[i] = ok5 ? 1.f / as : 0.f;
            }
            kb5[i] = ok5 ? p.b5[i] * as : 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t xr0 = make_rsrc(p.x + (long)b * p.x_b, Tv);
    const __amdgpu_buffer_rsrc_t xr1 = make_rsrc(p.x + p.x_sig + (long)b * p.x_b, Tv);
    const __amdgpu_buffer_rsrc_t ssr = act_rsrc(reinterpret_cast<const float*>(p.ss), (long)b * p.ss_b, (long)2 * CS_C * p.ld);
    const int hdTv = Tv / p.hd_s;
    const __amdgpu_buffer_rsrc_t hdr0 = act_rsrc(reinterpret_cast<const float*>(p.hd), (long)b * p.hd_b, (long)CS_C * p.hd_ld);
    const __amdgpu_buffer_rsrc_t hdr1 = act_rsrc(reinterpret_cast<const float*>(p.hd), p.hd_sig + (long)b * p.hd_b, (long)CS_C * p.hd_ld);

    // raw signal rows t0 - 16 .. of both signals for one tile, 0 outside the utterance: fetched one tile ahead
    constexpr int XN = (ROWS + CS_NTHREADS - 1) / CS_NTHREADS;
    float xpre[2][XN];
    auto xfetch = [&](int t0n) {
        #pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int r = tid + k * CS_NTHREADS;
            const int t = t0n - 16 + r;
            const int o = (r < ROWS && (unsigned)t < (unsigned)Tv) ? t * 4 : OOB_OFF;
            xpre[0][k] = buf_load1(xr0, o, 0);
            xpre[1][k] = buf_load1(xr1, o, 0);
        }
    };
    xfetch(tile_begin * NT);
#if defined(FASTSVC_COND_TRACE) && defined(FASTSVC_ACT_BF16)
    unsigned long long* trace = (p.amax_hd && blockIdx.x == 1 && blockIdx.z == 0 && lane == 0)
        ? reinterpret_cast<unsigned long long*>(p.amax_hd) + wave * 64 : nullptr;
    int tri = 0;
#define CS_STAMP() do { if (trace && tri < 64) trace[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define CS_STAMP() do {} while (0)
#endif
    // the signal rows of a tile reach LDS one tile ahead, from registers requested two tiles ahead (xs_commit sits in
    // front of the output stores of the tile before: vmcnt counts loads and stores in one in-order queue, and behind
    // those stores the wait for the rows also waited for the stores' acknowledgements - 12 % of the launch, measured)
    auto xs_commit = [&]() {
        #pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int r = tid + k * CS_NTHREADS;
            if (r < ROWS) { xs[r] = xpre[0][k]; xs[ROWS + r] = xpre[1][k]; }
        }
    };
    xs_commit();
    xfetch((tile_begin + 1) * NT);
    float hmax[2] = {0.f, 0.f};                            // float32 storage: largest |value| this lane wrote to hd, per signal
    __syncthreads();
    CS_STAMP();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int t0 = tile * NT;
        CS_STAMP();
        // ---- P1: c1 = lrelu(conv3(lrelu(x)) + b1) on the VALU, rows 8 .. NT + 24: thread = row, one (signal, octet of 8
        // channels) per step, whose taps are wave-uniform LDS reads ----
        if (!(dbg & 2)) {
            static_assert(2 * 16 * NTL <= CS_NTHREADS && (16 * NTL) % 64 == 0, "one (signal, c1 row) per thread, a wave = one signal");
            const int r = 8 + (tid & (16 * NTL - 1));
            const int t = t0 - 16 + r;
            const unsigned keep = (unsigned)t < (unsigned)Tv ? 0xffffffffu : 0u;    // outside the utterance: c2's zero padding
            if (tid < 2 * 16 * NTL) {
                const int s1 = wave / (16 * NTL / 64);
                const float* xrow = xs + s1 * ROWS;
                float xa = xrow[r - 1], xb = xrow[r], xc = xrow[r + 1];
                xa = fmaxf(xa, xa * LRELU_SLOPE); xb = fmaxf(xb, xb * LRELU_SLOPE); xc = fmaxf(xc, xc * LRELU_SLOPE);
                const f32x2 xa2 = {xa, xa}, xb2 = {xb, xb}, xc2 = {xc, xc};
                #pragma unroll
                for (int oct = 0; oct < 3; ++oct) {
                    f32x4 o4[2];
                    #pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float* kw = kin1 + s1 * 4 * 32 + oct * 8 + hh * 4;
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(kw), w1 = *reinterpret_cast<const f32x4*>(kw + 32),
                                    w2 = *reinterpret_cast<const f32x4*>(kw + 64), wb = *reinterpret_cast<const f32x4*>(kw + 96);
                        #pragma unroll
                        for (int pq = 0; pq < 2; ++pq) {          // packed float32: two channels per instruction
                            const f32x2 a0 = {w0[2 * pq], w0[2 * pq + 1]}, a1 = {w1[2 * pq], w1[2 * pq + 1]},
                                        a2 = {w2[2 * pq], w2[2 * pq + 1]}, ab = {wb[2 * pq], wb[2 * pq + 1]};
                            const f32x2 u = __builtin_elementwise_fma(a2, xc2, __builtin_elementwise_fma(a1, xb2, __builtin_elementwise_fma(a0, xa2, ab)));
                            o4[hh][2 * pq] = u.x; o4[hh][2 * pq + 1] = u.y;
                        }
                        o4[hh] = cs_lrelu4(o4[hh]);
                    }
                    unsigned char* dst = bufA + s1 * PLANE + cs_off(r, oct);
                    cs_store8_masked(dst, lo_off, o4[0], o4[1], keep);
                }
                // channel padding 24 .. 31 (the staging rows of the previous tile lay here)
                unsigned char* pad = bufA + s1 * PLANE + cs_off(r, 3);
                *reinterpret_cast<u32x4*>(pad) = u32x4{0u, 0u, 0u, 0u};
                if constexpr (CS_NP == 2) *reinterpret_cast<u32x4*>(pad + lo_off) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        CS_STAMP();
        __syncthreads();
        CS_STAMP();
        // ---- P2: c2 = lrelu(conv3_d2(c1) + b2): output rows 10 .. (t = -6 ..) ----
        if (!(dbg & 4)) cs_layer<NTL, 0>(bufA + sig * PLANE, bufB, lo_off, PLANE, W2, 2, 10, kbias + (0 * 2 + sig) * 32, kinv + (0 * 2 + sig) * 32, nullptr, nullptr, sig, mt, q, t0, Tv, lane);
        CS_STAMP();
        __syncthreads();
        CS_STAMP();
        // ---- P3: h = conv3_d4(c2) + b3 + (r1w x + r1b): output rows 14 .. (t = -2 ..) ----
        if (!(dbg & 4)) cs_layer<NTL, 1>(bufB + sig * PLANE, bufA, lo_off, PLANE, W3, 4, 14, kbias + (1 * 2 + sig) * 32, kinv + (1 * 2 + sig) * 32, xs + sig * ROWS, kr1 + sig * 32, sig, mt, q, t0, Tv, lane);
        CS_STAMP();
        __syncthreads();
        CS_STAMP();
        // ---- P4: u = lrelu(conv3_d1(h) + b4) -> [lft ; sine] channel planes; h[::s'] -> hd ----
        if (!(dbg & 4)) cs_layer<NTL, 2>(bufA + sig * PLANE, bufB, lo_off, PLANE, W4, 1, 15, kbias + (2 * 2 + sig) * 32, kinv + (2 * 2 + sig) * 32, nullptr, nullptr, sig, mt, q, t0, Tv, lane);
        if (p.hd && !(dbg & 8)) {
            // lane = decimated column, wave = the (signal, channel) rows sc = wave, wave + 8, ...: every LDS read of a
            // lane issued before the first store; addresses = four per-lane slot bases + immediates
            const int j_lo = (t0 + p.hd_s - 1) / p.hd_s;
            const int j_hi = min((min(t0 + NT, Tv) + p.hd_s - 1) / p.hd_s, hdTv);
            constexpr int NR = 2 * CS_C / CS_NWAVES;
            for (int j = j_lo + lane; j < j_hi; j += 64) {
                const int row = j * p.hd_s - t0 + 16;
                const unsigned char* bo[4];                  // the row's four slots (swizzled): immediates do the rest
                #pragma unroll
                for (int o = 0; o < 4; ++o) bo[o] = bufA + cs_off(row, o);
#ifdef FASTSVC_ACT_BF16
                unsigned short v[NR];
#else
                float v[NR];
                const float ih0 = 1.0f / sc[2][0], ih1 = 1.0f / sc[2][1];      // (exact: powers of two)
#endif
                #pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int sc = wave + CS_NWAVES * k;               // (wave-uniform)
                    const int s = sc >= CS_C ? 1 : 0, c = sc - s * CS_C;
                    const unsigned char* src = bo[c >> 3] + s * PLANE + (c & 7) * 2;
#ifdef FASTSVC_ACT_BF16
                    v[k] = *reinterpret_cast<const unsigned short*>(src);
#else
                    v[k] = ((float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + lo_off)) * (s ? ih1 : ih0);
                    hmax[s] = fmaxf(hmax[s], fabsf(v[k]));
#endif
                }
                #pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int sc = wave + CS_NWAVES * k;
                    const int s = sc >= CS_C ? 1 : 0, c = sc - s * CS_C;
#ifdef FASTSVC_ACT_BF16
                    __builtin_amdgcn_raw_buffer_store_b16(v[k], s ? hdr1 : hdr0, j * 2, c * p.hd_ld * 2, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), s ? hdr1 : hdr0, j * 4, c * p.hd_ld * 4, 0);
#endif
                }
            }
        }
        CS_STAMP();
        __syncthreads();
        CS_STAMP();
        // ---- P5: [scale ; shift] = conv3_d1([u_lft ; u_sine]) + b5: wave = (channel tile m5, half of the output tiles) ----
        if (!(dbg & 16) && wave < 6) {
            constexpr int NH = (NTO + 1) / 2;              // tiles per half (the second half may be one short)
            const int jb = half5 * NH;
            const int l15 = lane & 15, g = lane >> 4;
            int aoff[3];
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap) aoff[tap] = cs_off(16 + (tap - 1) + l15, g) + jb * 16 * CS_ROW;
            const float bias = kb5[m5 * 16 + l15];
            float oinv = 1.f;
            if constexpr (CS_NP == 2) oinv = k5inv[m5 * 16 + l15];
            unsigned char* srow = bufA + (m5 * 16 + l15) * GEO::SP + (jb * 16 + 4 * g) * (int)sizeof(act_t);
            constexpr int G5 = 4;
            #pragma unroll
            for (int i0 = 0; i0 < NH; i0 += G5) {
                f32x4 acc[G5];
                #pragma unroll
                for (int ii = 0; ii < G5; ++ii) acc[ii] = f32x4{bias, bias, bias, bias};
                #pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    #pragma unroll
                    for (int tap = 0; tap < 3; ++tap)
                        #pragma unroll
                        for (int ii = 0; ii < G5; ++ii) {
                            if (i0 + ii >= NH) continue;
                            const CsFrag a = cs_read(bufB + ch * PLANE, aoff[tap] + (i0 + ii) * 16 * CS_ROW, lo_off);
                            acc[ii] = cs_prod<false>(W5[ch][tap], a, acc[ii]);
                        }
                #pragma unroll
                for (int ii = 0; ii < G5; ++ii) {
                    if (i0 + ii >= NH) continue;
                    if (jb + i0 + ii >= NTO) continue;     // (wave-uniform: the odd tile of the second half)
                    f32x4 v = acc[ii];
                    if constexpr (CS_NP == 2) v = v * oinv;
                    unsigned char* dst = srow + (i0 + ii) * 16 * (int)sizeof(act_t);
#ifdef FASTSVC_ACT_BF16
                    *reinterpret_cast<cs4*>(dst) = __builtin_convertvector(v, cs4);
#else
                    // (not __builtin_bit_cast(unsigned, v.y): on a vector-element lvalue this hipcc reads element 0)
                    *reinterpret_cast<f32x4*>(dst) = v;
#endif
                }
            }
        }
        CS_STAMP();
        __syncthreads();
        CS_STAMP();
        // ---- P0 of the next tile: its signal rows -> LDS (every reader of xs is past the third barrier), the rows
        // after it requested (past the last tile: offsets out of range, nothing is fetched) ----
        if (!(dbg & 1)) { xs_commit(); xfetch(t0 + 2 * NT); }
        // ---- P6: staged rows -> ss, 16 bytes per lane, whole rows ----
        if (!(dbg & 32)) {
            // thread = (16-byte chunk k of a row, row cc = tid / 32 + 8 i): one LDS read and one store per chunk, the row
            // stepping in immediates / scalar offsets
            constexpr int CH = NT * (int)sizeof(act_t) / 16;       // 16-byte chunks per row
            constexpr int EPC = 16 / (int)sizeof(act_t);           // elements per chunk
            static_assert(CH <= 32 || (CH <= 64 && sizeof(act_t) == 4), "chunks of a row fit the lanes that copy it");
            constexpr int LPR = CH <= 32 ? 32 : 64;                // lanes per row
            constexpr int RPP = CS_NTHREADS / LPR;                 // rows per pass
            const int k = tid % LPR, r0 = tid / LPR;
            const int t = t0 + k * EPC;
            const bool ok = k < CH && t < Tv;
            const unsigned char* src = bufA + r0 * GEO::SP + k * 16;
            const int o = ok ? (r0 * p.ld + t) * (int)sizeof(act_t) : OOB_OFF;
            #pragma unroll
            for (int i = 0; i < 2 * CS_C / RPP; ++i) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(src + i * RPP * GEO::SP);
                __builtin_amdgcn_raw_buffer_store_b128(w, ssr, o, i * RPP * p.ld * (int)sizeof(act_t), 0);
            }
        }
        CS_STAMP();
        __syncthreads();                                   // staging reads and xs writes before the next tile's P1
        CS_STAMP();
    }
    if constexpr (CS_NP == 2) {
        // the maxima of what went to hd (the scale of stage 1's split-binary16 staging, ConvParams::amax_in): one atomic per
        // wave and signal, spread over the entry's 8 slots (fastsvc_device.inc)
        if (p.amax_hd && p.hd) {
            #pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, hmax[s])), 63);
                if (lane == 0 && a != 0u)
                    atomicMax(reinterpret_cast<unsigned*>(p.amax_hd) + (s * p.B + b) * AMAX_ENTRY + ((blockIdx.x + wave) & (AMAX_W - 1)) * AMAX_STRIDE, a);
            }
        }
    }
}

template <int NTL>
static hipError_t cond_stage0_instance(const CondStage0Params& p, hipStream_t stream) {
    using GEO = CsGeom<NTL>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_stage0_kernel<NTL>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS);
    if (attr != hipSuccess) return attr;
    const int ntx = (p.T + GEO::NT - 1) / GEO::NT;
    const int tpw = p.tpw & 0xffff;
    dim3 grid((ntx + tpw - 1) / tpw, 1, p.B);
    hipLaunchKernelGGL(cond_stage0_kernel<NTL>, grid, dim3(CS_NTHREADS), GEO::LDS, stream, p);
    return hipGetLastError();
}


// =====================================================================================================================
// Stage 0 as a LAYER PIPELINE (round 5): the same arithmetic, another schedule.  In the kernel above every wave walks
// the layers of a tile one after the other, six barriers per tile, every MFMA behind its own 1 KB activation fragment
// read: LDS array 0.74 ms busy + VALU 0.41 + matrix pipe 0.36 = the launch's 1.45 ms at cfg3 bfloat16 - the three
// resources take turns instead of overlapping (profiles/r4_cond_stage_sq_counters*.txt).  Here a WAVE OWNS A LAYER:
//
//     waves 0, 1    c1 of loudness / sine (VALU; reads the raw signal from memory)
//     waves 2, 3    c2        waves 4, 5   c3 (+ rank-1 residual)        waves 6, 7   film.conv
//     waves 8..10   one 16-channel tile of the heads each                wave 11       ss and hd to memory
//
// and one workgroup streams a long run of consecutive 16 N-column chunks of one utterance through them: at step s layer
// l works on chunk s - l (shifted one more 16-column tile per layer: its right halo is then already there), ONE barrier
// per step.  A layer wave keeps BOTH 16-channel output tiles of its layer: one fragment read feeds two MFMAs (half the
// reads of the kernel above), its 6 weight fragments stay resident, and at any moment the waves of a SIMD are in
// different kinds of work (VALU conv, matrix layers, copy-out).  Nothing is recomputed at chunk borders (the kernel
// above computes N + 1 tiles per layer for N stored); the price is the fill: 5 + 8 / N steps before the first column
// leaves, so launch_cond_stage0 takes this kernel for long runs only (>= 24 chunks per workgroup).
//
// LDS: every inter-layer tensor is a RING of three chunks per plane (producer writes chunk position (s - l) % 3 while the
// consumer reads position (s - l - 1) % 3 and the last two tiles of the one before), the step loop is unrolled three
// times so that every ring position - hence every LDS address - is a compile-time immediate on a per-lane base.  Eight
// guard rows below and above a ring mirror its last / first rows (written by whoever writes those), so that a tap
// offset never wraps inside an instruction.  Tile P of a layer's output ring holds the columns of tile P - 1 (...) of
// the ring before: a layer reads its input ring at tiles P - 2 .. P, row offset -16 + (tap - 1) dilation.
constexpr int P0_NWAVES = 12, P0_NTHREADS = P0_NWAVES * 64;

template <int N>
struct P0Geom {
    static constexpr int NT = 16 * N;                  // columns per chunk (step)
    static constexpr int RING = 3 * NT;
    static constexpr int GUARD = 8;
    static constexpr int PROWS = RING + 2 * GUARD;     // physical rows of a plane: ring row r at GUARD + r
    static constexpr int PLANE_P = PROWS * CS_ROW;     // one piece (hi; float32 storage: lo behind it)
    static constexpr int PLANE = CS_NP * PLANE_P;
    static constexpr int SP = NT * (int)sizeof(act_t) + 16;
    static constexpr int STG = 2 * CS_C * SP;          // one staging buffer of [scale ; shift] rows
    static constexpr int CONST_FLOATS = 2 * 32 * 4 + 2 * 32 + 2 * 3 * 2 * 32 + 2 * 64;
    static constexpr int LAG = 8 / N;                  // chunks of the heads' stream in front of the workgroup's first column
    static constexpr int DUMMY = 64 * 32;              // 32 bytes per lane (shared by the waves: nobody reads them): where the guard-row copies of unmirrored lanes go
    // the raw signals in LDS: a ring of four blocks of XB chunks per signal, a block requested XB steps before it is
    // committed (one step ahead the c1 / c3 waves sat through a memory latency per step: ~1.7 us, the whole pipeline's pace)
    static constexpr int XB = 8;                       // chunks per block
    static constexpr int BLK = XB * NT;                // floats per block
    static constexpr int XR = 4 * BLK;                 // floats per signal ring (a power of two)
    static constexpr int XPL = BLK / 64;               // floats per lane and block
    static constexpr size_t LDS = CONST_FLOATS * 4 + 8 * (size_t)PLANE + 2 * (size_t)STG + DUMMY + 2 * (size_t)XR * 4;
    static_assert((XR & (XR - 1)) == 0, "ring index by masking");
    static_assert(8 % N == 0 && N % 2 == 0, "the layers' total lag (8 tiles) is a whole number of chunks; the heads take tiles in pairs");
};

// workgroup barrier of the pipeline: LDS traffic of the step done (lgkmcnt), memory traffic left in flight - the loads
// a role issued for its NEXT step and the copy wave's stores must not be waited for here
__device__ __forceinline__ void p0_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// packed tile values, computed once and stored twice where a tile also feeds guard rows: the copy goes through a SELECT of
// the destination (lanes outside the mirrored rows store into a dummy slot nobody reads) instead of a divergent branch
struct CsPk4 { cs_u2 h, l; };
__device__ __forceinline__ CsPk4 cs_pack4(f32x4 v, unsigned keep) {
    const cs4 h = __builtin_convertvector(v, cs4);
    CsPk4 r;
    r.h = __builtin_bit_cast(cs_u2, h);
    r.l = r.h;
    if constexpr (CS_NP == 2) { r.l = cs_u2{cs_lo_pair(r.h.x, v[0], v[1]) & keep, cs_lo_pair(r.h.y, v[2], v[3]) & keep}; }
    r.h.x &= keep; r.h.y &= keep;
    return r;
}
__device__ __forceinline__ void cs_put4(unsigned char* hi_dst, unsigned char* lo_dst, const CsPk4& r) {
    if constexpr (CS_NP == 2) *reinterpret_cast<cs_u2*>(lo_dst) = r.l;
    *reinterpret_cast<cs_u2*>(hi_dst) = r.h;
}
struct CsPk8 { u32x4 h, l; };
__device__ __forceinline__ CsPk8 cs_pack8(f32x4 a, f32x4 b, unsigned keep) {
    const cs4 ha = __builtin_convertvector(a, cs4), hb = __builtin_convertvector(b, cs4);
    const cs_u2 pa = __builtin_bit_cast(cs_u2, ha), pb = __builtin_bit_cast(cs_u2, hb);
    CsPk8 r;
    r.h = u32x4{pa.x & keep, pa.y & keep, pb.x & keep, pb.y & keep};
    r.l = r.h;
    if constexpr (CS_NP == 2)
        r.l = u32x4{cs_lo_pair(pa.x, a[0], a[1]) & keep, cs_lo_pair(pa.y, a[2], a[3]) & keep,
                    cs_lo_pair(pb.x, b[0], b[1]) & keep, cs_lo_pair(pb.y, b[2], b[3]) & keep};
    return r;
}
__device__ __forceinline__ void cs_put8(unsigned char* hi_dst, unsigned char* lo_dst, const CsPk8& r) {
    *reinterpret_cast<u32x4*>(hi_dst) = r.h;
    if constexpr (CS_NP == 2) *reinterpret_cast<u32x4*>(lo_dst) = r.l;
}

// the step loop, unrolled three times: J = step % 3 is a compile-time constant inside the body, and with it every ring position
template <int V> struct P0Const { static constexpr int value = V; };
template <class F>
__device__ __forceinline__ void p0_steps(int nsteps, F&& body) {
    for (int s = 0;;) {
        if (s >= nsteps) break;
        body(P0Const<0>{}, s); p0_barrier(); ++s;
        if (s >= nsteps) break;
        body(P0Const<1>{}, s); p0_barrier(); ++s;
        if (s >= nsteps) break;
        body(P0Const<2>{}, s); p0_barrier(); ++s;
    }
}

// one chunk of a k=3 layer: both 16-channel output tiles per activation fragment (swapped operands, see cs_layer)
//   KIND 0: lrelu -> own signal's plane;  1: + rank-1 residual, raw;  2: lrelu -> the 2C-channel plane pair of the heads
template <int N, int POS, int KIND>
__device__ __forceinline__ void p0_layer_chunk(const unsigned char* in_plane, unsigned char* out_base, const CsW (&W)[3][2],
                                               const int (&rd)[3], const int (&wr)[2], const f32x4 (&kbv)[2], const f32x4 (&kivv)[2],
                                               const f32x4 (&r1wv)[2], const float (&xv)[N], int tstart, int Tv, int lane,
                                               unsigned char* dummy /* this lane's 32 bytes nobody reads */) {
    using GEO = P0Geom<N>;
    constexpr int lo_off = GEO::PLANE_P;
    constexpr int NTL3 = 3 * N;
    const int l15 = lane & 15;
    // every fragment of the chunk is requested before the first product and every store comes after the last read: with
    // reads and stores of consecutive tiles interleaved hipcc kept them in program order (it cannot see that the two rings
    // never overlap) and a wave paid one LDS latency + one MFMA drain per tile
    CsFrag a[N][3];
    #pragma unroll
    for (int i = 0; i < N; ++i) {
        const int PIN = (POS * N + i - 1 + NTL3) % NTL3;
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) a[i][tap] = cs_read(in_plane, rd[tap] + PIN * 16 * CS_ROW, lo_off);
    }
    f32x4 acc[N][2];
    #pragma unroll
    for (int i = 0; i < N; ++i)
        #pragma unroll
        for (int m = 0; m < 2; ++m) {
            // Every accumulator starts from the bias vector, a register written long before the products; the rank-1 residual
            // (KIND 1) joins AFTER them.  Round 5 computed r1w * x + bias here, (packed) float32 FMAs right in front of the
            // v_mfma that read the result as its C operand, and the float32-storage pipeline returned run-to-run DIFFERENT values:
            // single registers of lanes 48-63 of an initial value stale, in a handful of tiles per 10^6 at 64 x 1500
            // (tools/cond_pipe_determinism.py) - hipcc's two wait states between such a write and the MFMA were not enough under
            // this kernel's load; eight (`s_nop 7`) hid it, empirically.  Now no VALU result is ever an MFMA's C operand in these
            // kernels: nothing to fence (tests/test_parity_gpu.py runs the determinism check at the shape that exposed it).
            acc[i][m] = kbv[m];
        }
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap)
        #pragma unroll
        for (int i = 0; i < N; ++i)
            #pragma unroll
            for (int m = 0; m < 2; ++m) acc[i][m] = cs_prod<true>(W[tap][m], a[i][tap], acc[i][m]);
    #pragma unroll
    for (int i = 0; i < N; ++i) {
        const int P = POS * N + i;                     // (compile-time after unrolling)
        const int tt = tstart + 16 * i;
        const bool edge = tt < 0 || tt + 16 > Tv;      // (wave-uniform) a tile that crosses an end of the utterance
        unsigned keep = 0xffffffffu;
        if (edge) keep = (unsigned)(tt + l15) < (unsigned)Tv ? 0xffffffffu : 0u;
        #pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 v = acc[i][m];
            if constexpr (KIND == 1) v = r1wv[m] * xv[i] + v;
            if constexpr (CS_NP == 2) v = v * kivv[m];
            if constexpr (KIND != 1) v = cs_lrelu4(v);
            unsigned char* dst = out_base + wr[m] + P * 16 * CS_ROW;
            const CsPk4 pk = cs_pack4(v, keep);
            cs_put4(dst, dst + lo_off, pk);
            // guard rows: the ring's first 8 rows again above it, its last 8 again below it (lanes outside them: a dummy slot)
            if (P == 0) { unsigned char* g2 = l15 < 8 ? dst + GEO::RING * CS_ROW : dummy; cs_put4(g2, g2 + (l15 < 8 ? lo_off : 16), pk); }
            if (P == NTL3 - 1) { unsigned char* g2 = l15 >= 8 ? dst - GEO::RING * CS_ROW : dummy; cs_put4(g2, g2 + (l15 >= 8 ? lo_off : 16), pk); }
        }
    }
}

template <int N>
__global__ __launch_bounds__(P0_NTHREADS, 1)
void cond_stage0_pipe_kernel(const CondStage0Params p) {
    using GEO = P0Geom<N>;
    constexpr int NT = GEO::NT, RING = GEO::RING, PLANE = GEO::PLANE, GUARD = GEO::GUARD, LAG = GEO::LAG;
    constexpr int lo_off = GEO::PLANE_P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* kin1 = reinterpret_cast<float*>(smem);                                 // tables: see cond_stage0_kernel
    float* kr1 = kin1 + 2 * 32 * 4;
    float* kbias = kr1 + 2 * 32;
    float* kinv = kbias + 3 * 2 * 32;
    float* kb5 = kinv + 3 * 2 * 32;
    float* k5inv = kb5 + 64;
    unsigned char* planes = reinterpret_cast<unsigned char*>(k5inv + 64);         // [c1, c2, h][signal] and the u pair: 8 planes
    unsigned char* stg = planes + 8 * PLANE;                                      // 2 staging buffers
    unsigned char* dummy = stg + 2 * GEO::STG + (threadIdx.x & 63) * 32;
    float* xring = reinterpret_cast<float*>(stg + 2 * GEO::STG + GEO::DUMMY);     // [signal][XR]: x(t) at (t - tx0) & (XR - 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int Tv = p.lens ? p.lens[b] * p.len_mul : p.T;
    const int Kc = p.tpw & 0xffff;                         // output chunks per workgroup
    const int dbg = p.tpw >> 16;                           // developer ablation switches (tools/cond_pipe_ablate.sh): results invalid
    const int T0 = blockIdx.x * Kc * NT;
    if (T0 >= Tv) return;
    const int nch = min(Kc, (Tv - T0 + NT - 1) / NT);
    const int T1 = min(T0 + nch * NT, Tv);
    const int Ktot = nch + LAG;                            // chunks every layer runs
    const int nsteps = Ktot + 5;                           // layer l is busy at steps l .. l + Ktot - 1; l = 5: the copy wave
    const int tx0 = T0 - 64 - GEO::BLK;                    // time of ring index 0 (block 0 = the block in front of c1's first chunk)

    // ---- operand scales (float32 storage) and tables: exactly cond_stage0_kernel's ----
    float sc[4][2] = {{1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}};
    if constexpr (CS_NP == 2) {
        float bu = 0.f;
        #pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float ax = amax_read(p.amax_in, s * p.B + b);
            float bt[4];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = lane < CS_C ? p.cbnd[s][(t * 2 + 0) * CS_C + lane] * ax + p.cbnd[s][(t * 2 + 1) * CS_C + lane] : 0.f;
                bt[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, fmaxf(v, 0.f))), 63));
            }
            bu = fmaxf(bu, bt[3]);
            sc[0][s] = hx_scale_for(bt[0]); sc[1][s] = hx_scale_for(bt[1]); sc[2][s] = hx_scale_for(bt[2]);
        }
        sc[3][0] = sc[3][1] = hx_scale_for(bu);
    }
    for (int i = tid; i < 2 * 32; i += P0_NTHREADS) {
        const int s = i >> 5, c = i & 31;
        const bool ok = c < CS_C;
        const float* wp = p.in1_w[s] + (ok ? c : 0) * 3;
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) kin1[(s * 4 + tap) * 32 + c] = ok ? wp[tap] * sc[0][s] : 0.f;
        kin1[(s * 4 + 3) * 32 + c] = ok ? p.in1_b[s][c] * sc[0][s] : 0.f;
        #pragma unroll
        for (int l = 0; l < 3; ++l) {
            float as = 1.f, os = 1.f;
            if constexpr (CS_NP == 2) {
                const float wi = ok ? p.winv[l][s][c] : 1.f;
                as = sc[l][s] / wi;
                os = sc[l + 1][s];
                kinv[(l * 2 + s) * 32 + c] = ok ? os / as : 0.f;
            }
            kbias[(l * 2 + s) * 32 + c] = ok ? (p.bias[l][s][c] + (l == 1 ? p.r1b[s][c] : 0.f)) * as : 0.f;
            if (l == 1) kr1[s * 32 + c] = ok ? p.r1w[s][c] * as : 0.f;
        }
        {
            const bool ok5 = i < 2 * CS_C;
            float as = 1.f;
            if constexpr (CS_NP == 2) {
                as = sc[3][0] / (ok5 ? p.winv5[i] : 1.f);
                k5inv[i] = ok5 ? 1.f / as : 0.f;
            }
            kb5[i] = ok5 ? p.b5[i] * as : 0.f;
        }
    }
    // every plane starts as zeros: the channel padding of the rings is never written again (c1's slot 3, the pair's
    // channels 2C + 8 ..), and what a layer computes from not yet written rows during the fill must be finite
    for (int o = tid * 16; o < 8 * PLANE + 2 * GEO::STG + GEO::DUMMY + 2 * GEO::XR * 4; o += P0_NTHREADS * 16)
        *reinterpret_cast<u32x4*>(planes + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const int l15 = lane & 15, g = lane >> 4;
    if (wave < 2) {
        // ================= c1 = lrelu(conv3(lrelu(x)) + b1) on the VALU: lane = column (x 64 / NT channel groups) =================
        const int sig = wave;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + sig * p.x_sig + (long)b * p.x_b, Tv);
        constexpr int LPC = 64 / NT;                       // lanes per column: 1 (NT = 64: three octets each) or 2 (octets {0, 1} | {2})
        const int col = lane & (NT - 1), part = lane / NT;
        unsigned char* plane = planes + (0 * 2 + sig) * PLANE;
        // the taps and bias of this lane's channels stay in registers (as wave-uniform LDS reads they were 24 ds_read_b128
        // per step, each a latency the in-order wave sat through)
        constexpr int NO = LPC == 1 ? 3 : 2;               // octets per lane
        f32x2 wk[NO][4][4];                                // [octet][channel pair][w0 w1 w2 b]
        #pragma unroll
        for (int oo = 0; oo < NO; ++oo) {
            const int oct = LPC == 1 ? oo : (part == 0 ? oo : 2);
            #pragma unroll
            for (int pq = 0; pq < 4; ++pq)
                #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* kw = kin1 + (sig * 4 + c) * 32 + oct * 8 + 2 * pq;
                    wk[oo][pq][c] = f32x2{kw[0], kw[1]};
                }
        }
        // this wave keeps its signal's ring filled: block j (ring indices j BLK ..) is requested when block j - 1 is committed,
        // XB steps before its own commit; blocks 0 .. 2 before the first step
        constexpr int XB = GEO::XB, BLK = GEO::BLK, XR = GEO::XR, XPL = GEO::XPL;
        float* xs = xring + sig * XR;
        float xblk[XPL];
        auto xrequest = [&](int j) {
            #pragma unroll
            for (int r = 0; r < XPL; ++r) {
                const int t = tx0 + j * BLK + r * 64 + lane;
                xblk[r] = buf_load1(xr, ((unsigned)t < (unsigned)Tv && !(dbg & 1)) ? t * 4 : OOB_OFF, 0);
            }
        };
        auto xcommit = [&](int j) {
            #pragma unroll
            for (int r = 0; r < XPL; ++r) xs[(j * BLK + r * 64 + lane) & (XR - 1)] = xblk[r];
        };
        for (int j = 0; j < 3; ++j) { xrequest(j); xcommit(j); }
        xrequest(3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");       // (the wave's own reads below follow its writes in order)
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            if (s < Ktot && !(dbg & 8)) {
                constexpr int POS = J;
                const int tc0 = T0 - 64 + s * NT;
                const int u = BLK + s * NT + col;              // ring index of this lane's column
                float xa = xs[(u - 1) & (XR - 1)], xb = xs[u & (XR - 1)], xc = xs[(u + 1) & (XR - 1)];
                xa = fmaxf(xa, xa * LRELU_SLOPE); xb = fmaxf(xb, xb * LRELU_SLOPE); xc = fmaxf(xc, xc * LRELU_SLOPE);
                const f32x2 xa2 = {xa, xa}, xb2 = {xb, xb}, xc2 = {xc, xc};
                const bool edge = tc0 < 0 || tc0 + NT > Tv;
                unsigned keep = 0xffffffffu;
                if (edge) keep = (unsigned)(tc0 + col) < (unsigned)Tv ? 0xffffffffu : 0u;
                const int row = GUARD + POS * NT + col;
                #pragma unroll
                for (int oo = 0; oo < NO; ++oo) {
                    const int oct = LPC == 1 ? oo : (part == 0 ? oo : 2);
                    if (LPC == 2 && part == 1 && oo == 1) break;
                    f32x4 o4[2];
                    #pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        #pragma unroll
                        for (int pq = 0; pq < 2; ++pq) {          // packed float32: two channels per instruction
                            const f32x2 (&w)[4] = wk[oo][hh * 2 + pq];
                            const f32x2 u = __builtin_elementwise_fma(w[2], xc2, __builtin_elementwise_fma(w[1], xb2, __builtin_elementwise_fma(w[0], xa2, w[3])));
                            o4[hh][2 * pq] = u.x; o4[hh][2 * pq + 1] = u.y;
                        }
                        o4[hh] = cs_lrelu4(o4[hh]);
                    }
                    unsigned char* dst = plane + cs_off(row, oct);
                    const CsPk8 pk = cs_pack8(o4[0], o4[1], keep);
                    cs_put8(dst, dst + lo_off, pk);
                    if (POS == 0) { unsigned char* g2 = col < 8 ? dst + RING * CS_ROW : dummy; cs_put8(g2, g2 + (col < 8 ? lo_off : 16), pk); }
                    if (POS == 2) { unsigned char* g2 = col >= NT - 8 ? dst - RING * CS_ROW : dummy; cs_put8(g2, g2 + (col >= NT - 8 ? lo_off : 16), pk); }
                }
            }
            // block turn-over every XB steps: commit block m + 2 (requested XB steps ago; first needed by the last chunk of
            // this block of steps), request block m + 3 - into the slot of block m - 1, whose last reader (c3, two steps
            // and two tiles behind) left it a block of steps ago
            if (s > 0 && (s & (XB - 1)) == 0) {
                const int m = s / XB;
                xcommit(m + 2);
                xrequest(m + 3);
            }
        });
    } else if (wave < 8) {
        // ================= c2 / c3 / film.conv of one signal: both 16-channel output tiles per fragment =================
        const int L = 1 + ((wave - 2) >> 1), sig = wave & 1;         // layer 1 c2, 2 c3, 3 film.conv
        CsW W[3][2];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            #pragma unroll
            for (int m = 0; m < 2; ++m)
                W[tap][m] = cs_wload(reinterpret_cast<const unsigned char*>(p.w[L - 1][sig]) + (long)(tap * 2 + m) * CS_NP * CS_FRAG, lane);
        const int dil = L == 1 ? 2 : L == 2 ? 4 : 1;
        int rd[3], wr[2];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) rd[tap] = cs_off(GUARD + (tap - 1) * dil + l15, g);
        f32x4 kbv[2], kivv[2], r1wv[2];
        #pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int co0 = m * 16 + 4 * g;
            kbv[m] = *reinterpret_cast<const f32x4*>(kbias + ((L - 1) * 2 + sig) * 32 + co0);
            kivv[m] = CS_NP == 2 ? *reinterpret_cast<const f32x4*>(kinv + ((L - 1) * 2 + sig) * 32 + co0) : kbv[m];
            r1wv[m] = L == 2 ? *reinterpret_cast<const f32x4*>(kr1 + sig * 32 + co0) : kbv[m];
            if (L == 3) {
                // (this signal's channel padding would land on the other signal's channels: its zeros go to the pair's own padding)
                const int cc0 = co0 < CS_C ? sig * CS_C + co0 : 2 * CS_C + 8 + (co0 - CS_C);
                wr[m] = (cc0 >> 5) * PLANE + cs_off(GUARD + l15, (cc0 & 31) >> 3) + (cc0 & 7) * 2;
            } else {
                wr[m] = cs_off(GUARD + l15, co0 >> 3) + (co0 & 7) * 2;
            }
        }
        const unsigned char* in_plane = planes + ((L - 1) * 2 + sig) * PLANE;
        unsigned char* out_base = planes + (L == 3 ? 6 * PLANE : (L * 2 + sig) * PLANE);
        const float* xs = xring + sig * GEO::XR;
        float xv[N];
        #pragma unroll
        for (int i = 0; i < N; ++i) xv[i] = 0.f;
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            const int k = s - L;
            if (k >= 0 && k < Ktot && !(dbg & 16)) {
                constexpr int P1 = (J + 2) % 3, P2 = (J + 1) % 3, P3 = J;       // (J - L) mod 3
                const int tstart = T0 - 64 + 16 * (k * N - L);
                if (L == 1) p0_layer_chunk<N, P1, 0>(in_plane, out_base, W, rd, wr, kbv, kivv, r1wv, xv, tstart, Tv, lane, dummy);
                else if (L == 2) {
                    #pragma unroll
                    for (int i = 0; i < N; ++i) xv[i] = xs[(GEO::BLK + 16 * (k * N - 2) + 16 * i + l15) & (GEO::XR - 1)];   // the raw signal at c3's columns
                    p0_layer_chunk<N, P2, 1>(in_plane, out_base, W, rd, wr, kbv, kivv, r1wv, xv, tstart, Tv, lane, dummy);
                } else p0_layer_chunk<N, P3, 2>(in_plane, out_base, W, rd, wr, kbv, kivv, r1wv, xv, tstart, Tv, lane, dummy);
            }
        });
    } else if (wave < 11) {
        // ================= heads: [scale ; shift] = conv3([u_lft ; u_sine]) + b5, one 16-channel output tile per wave =================
        const int m5 = wave - 8;
        CsW W5[2][3];
        #pragma unroll
        for (int ch = 0; ch < 2; ++ch)
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap)
                W5[ch][tap] = cs_wload(reinterpret_cast<const unsigned char*>(p.w5) + (long)((ch * 3 + tap) * 3 + m5) * CS_NP * CS_FRAG, lane);
        int rd[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) rd[tap] = cs_off(GUARD + (tap - 1) + l15, g);
        const float bias = kb5[m5 * 16 + l15];
        float oinv = 1.f;
        if constexpr (CS_NP == 2) oinv = k5inv[m5 * 16 + l15];
        const unsigned char* upl = planes + 6 * PLANE;
        const int srow = (m5 * 16 + l15) * GEO::SP + 4 * g * (int)sizeof(act_t);
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            const int k = s - 4;
            if (k >= LAG && k < Ktot && !(dbg & 32)) {
                constexpr int POS = (J + 2) % 3;               // (J - 4) mod 3
                unsigned char* sb = stg + (s & 1) * GEO::STG + srow;
                // two tiles at a time (two independent accumulator chains), a K chunk's fragments requested together
                #pragma unroll
                for (int i0 = 0; i0 < N; i0 += 2) {
                    f32x4 acc[2] = {{bias, bias, bias, bias}, {bias, bias, bias, bias}};
                    #pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        CsFrag a[2][3];
                        #pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int PIN = (POS * N + i0 + ii - 1 + 3 * N) % (3 * N);
                            #pragma unroll
                            for (int tap = 0; tap < 3; ++tap) a[ii][tap] = cs_read(upl + ch * PLANE, rd[tap] + PIN * 16 * CS_ROW, lo_off);
                        }
                        #pragma unroll
                        for (int tap = 0; tap < 3; ++tap)
                            #pragma unroll
                            for (int ii = 0; ii < 2; ++ii) acc[ii] = cs_prod<false>(W5[ch][tap], a[ii][tap], acc[ii]);
                    }
#ifdef FASTSVC_ACT_BF16
                    cs_u2 outv[2];
#else
                    f32x4 outv[2];
#endif
                    #pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        f32x4 v = acc[ii];
                        if constexpr (CS_NP == 2) v = v * oinv;
                        unsigned char* dst = sb + (i0 + ii) * 16 * (int)sizeof(act_t);
#ifdef FASTSVC_ACT_BF16
                        outv[ii] = __builtin_bit_cast(cs_u2, __builtin_convertvector(v, cs4));
                        *reinterpret_cast<cs_u2*>(dst) = outv[ii];
#else
                        outv[ii] = v;
                        *reinterpret_cast<f32x4*>(dst) = v;
#endif
                    }
                }
            }
        });
    } else {
        // ================= copy-out: staged [scale ; shift] rows -> ss (16 bytes per lane), h[::s'] -> hd =================
        const __amdgpu_buffer_rsrc_t ssr = act_rsrc(reinterpret_cast<const float*>(p.ss), (long)b * p.ss_b, (long)2 * CS_C * p.ld);
        const int hdTv = (int)udiv_small((unsigned)Tv, p.hd_s);
        // ONE descriptor over both signals' rows of utterance b (a lane-dependent descriptor or scalar offset makes hipcc loop
        // over the distinct values - twelve waterfall loops per step were 60 % of the launch); the launcher checks the 2 GB range
        const __amdgpu_buffer_rsrc_t hdr = act_rsrc(reinterpret_cast<const float*>(p.hd ? p.hd : p.ss), p.hd ? (long)b * p.hd_b : 0,
                                                    p.hd ? p.hd_sig + (long)CS_C * p.hd_ld : 0);
        constexpr int CH = NT * (int)sizeof(act_t) / 16;       // 16-byte pieces per staged row (8)
        constexpr int EPC = 16 / (int)sizeof(act_t);
        static_assert(64 % CH == 0 && (2 * CS_C) % (64 / CH) == 0, "whole passes over the staged rows");
        constexpr int RPP = 64 / CH;                           // rows per pass
        const int ck = lane % CH, cr = lane / CH;
        float hmax[2] = {0.f, 0.f};
        const int jj = lane & 15, q = lane >> 4;               // hd: lane = (decimated column, 12 channels of one signal)
        const int hs = q >> 1, hc0 = (q & 1) * 12;
#ifndef FASTSVC_ACT_BF16
        const float ih = 1.0f / sc[2][hs];                     // (exact: powers of two)
#endif
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            // ---- ss: the chunk the heads staged in the step before ----
            {
                const int kk = s - 5 - LAG;
                if (kk >= 0 && kk < nch && !(dbg & 2)) {
                    const unsigned char* sb = stg + ((s - 1) & 1) * GEO::STG + cr * GEO::SP + ck * 16;
                    const int t = T0 + kk * NT + ck * EPC;
                    const int o = t < Tv ? (cr * p.ld + t) * (int)sizeof(act_t) : OOB_OFF;
                    #pragma unroll
                    for (int i = 0; i < 2 * CS_C / RPP; ++i) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(sb + i * RPP * GEO::SP);
                        __builtin_amdgcn_raw_buffer_store_b128(w, ssr, o, i * RPP * p.ld * (int)sizeof(act_t), 0);
                    }
                }
            }
            // ---- hd: every hd_s-th column of the h chunk c3 finished in the step before ----
            {
                const int k2 = s - 3;
                if (p.hd && k2 >= 0 && k2 < Ktot && !(dbg & 4)) {
                    constexpr int POS = J;                     // (J - 3) mod 3
                    const int th0 = T0 - 64 + 16 * (k2 * N - 2);
                    const int ta = max(th0, T0), tb = min(th0 + NT, T1);
                    const int j_lo = (int)udiv_small((unsigned)(ta + p.hd_s - 1), p.hd_s);
                    const int j_hi = min((int)udiv_small((unsigned)(max(tb, ta) + p.hd_s - 1), p.hd_s), hdTv);
                    const int hrow0 = (int)(hs * p.hd_sig) + hc0 * p.hd_ld;          // element offset of this lane's first row
                    for (int j = j_lo + jj; j < j_hi; j += 16) {
                        const int row = GUARD + POS * NT + (j * p.hd_s - th0);
                        const unsigned char* src = planes + (2 * 2 + hs) * PLANE;
                        cs_u2 hv[3];
                        #pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const int c = hc0 + 4 * e;
                            hv[e] = *reinterpret_cast<const cs_u2*>(src + cs_off(row, c >> 3) + (c & 7) * 2);
                        }
#ifdef FASTSVC_ACT_BF16
                        #pragma unroll
                        for (int e = 0; e < 3; ++e)
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const unsigned w = u < 2 ? hv[e].x : hv[e].y;
                                const unsigned short v = (unsigned short)((u & 1) ? (w >> 16) : (w & 0xffffu));
                                __builtin_amdgcn_raw_buffer_store_b16(v, hdr, (hrow0 + j) * 2, (4 * e + u) * p.hd_ld * 2, 0);
                            }
#else
                        cs_u2 lv[3];
                        #pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const int c = hc0 + 4 * e;
                            lv[e] = *reinterpret_cast<const cs_u2*>(src + lo_off + cs_off(row, c >> 3) + (c & 7) * 2);
                        }
                        #pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const cs4 h4 = __builtin_bit_cast(cs4, hv[e]), l4 = __builtin_bit_cast(cs4, lv[e]);
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float v = ((float)h4[u] + (float)l4[u]) * ih;
                                hmax[hs] = fmaxf(hmax[hs], fabsf(v));
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), hdr, (hrow0 + j) * 4, (4 * e + u) * p.hd_ld * 4, 0);
                            }
                        }
#endif
                    }
                }
            }
        });
        if constexpr (CS_NP == 2) {
            if (p.amax_hd && p.hd) {
                #pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, hmax[s])), 63);
                    if (lane == 0 && a != 0u)
                        atomicMax(reinterpret_cast<unsigned*>(p.amax_hd) + (s * p.B + b) * AMAX_ENTRY + (blockIdx.x & (AMAX_W - 1)) * AMAX_STRIDE, a);
                }
            }
        }
    }
}

template <int N>
static hipError_t cond_stage0_pipe_instance(const CondStage0Params& p, hipStream_t stream) {
    using GEO = P0Geom<N>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_stage0_pipe_kernel<N>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS);
    if (attr != hipSuccess) return attr;
    const int nchunks = (p.T + GEO::NT - 1) / GEO::NT;
    const int kc = p.tpw & 0xffff;
    dim3 grid((nchunks + kc - 1) / kc, 1, p.B);
    hipLaunchKernelGGL(cond_stage0_pipe_kernel<N>, grid, dim3(P0_NTHREADS), GEO::LDS, stream, p);
    return hipGetLastError();
}

// p.small: 1 = the 112-column tile variant (more, shorter workgroups: batches that do not fill the chip with 240-column
// tiles); 2 = the layer pipeline (long runs of chunks per workgroup; tpw = chunks per workgroup)
#ifdef FASTSVC_ACT_BF16
constexpr int P0_N = 4;
#else
constexpr int P0_N = 2;
#endif
hipError_t launch_cond_stage0(const CondStage0Params& p, hipStream_t stream) {
    if (p.C != CS_C || (p.T % 8) != 0 || (p.ld % 8) != 0 || (p.tpw & 0xffff) < 1) return hipErrorInvalidValue;
    if (p.small == 2) {
        if (p.hd && (p.hd_sig + (long)CS_C * p.hd_ld) * (long)sizeof(act_t) >= (1L << 31)) return hipErrorInvalidValue;   // (one descriptor over both signals)
        return cond_stage0_pipe_instance<P0_N>(p, stream);
    }
    return p.small ? cond_stage0_instance<8>(p, stream) : cond_stage0_instance<16>(p, stream);
}

#ifndef FASTSVC_ACT_BF16
// columns per tile (small = 0 / 1) or per chunk of the pipeline (small = 2: float32 storage; 3: bfloat16 storage)
int cond_stage0_tile_columns(int small) {
    return small == 2 ? 16 * 2 : small == 3 ? 16 * 4 : small ? CsGeom<8>::NT : CsGeom<16>::NT;
}
#endif


// =====================================================================================================================
// Stage k >= 1 with C = 48 (C_in = 24; the yaml's stage 1): the same whole-stage launch.  Differences from stage 0:
//   * the stage's input is a C_in-channel tensor - the COMPACT decimated output of the stage before - staged time-major in
//     LDS twice (raw for the 1x1 residual conv, LeakyReLU'd for the first k=3 conv: the MODE_DEC2 pair of fastsvc_hx.hip),
//     fetched one tile ahead; both convs of the pair run on the matrix pipe, the residual conv's tile stays in registers
//     until it becomes the initial accumulator of the chain's last conv;
//   * C = 48 = two 32-channel K chunks per signal (8 LDS planes for the two ping-pong buffers), three 16-channel output
//     tiles: wave = (signal, output tile), all time tiles of the layer; heads 96 -> 96: wave = one of the six output tiles;
//   * 31 weight fragments per wave do not stay resident: each layer's are requested one layer ahead (two register sets
//     that trade places every tile), 186 KB of L2 traffic per 112-column tile against ~11 k cycles of matrix work.
constexpr int C1_C = 48, C1_CIN = 24, C1_NM = 3, C1_NC = 2;
constexpr int C1_NWAVES = 2 * C1_NM, C1_NTHREADS = C1_NWAVES * 64;

template <int NTL>
struct C1Geom {
    static constexpr int NTO = NTL - 1;
    static constexpr int NT = 16 * NTO;
    static constexpr int ROWS = NT + 36;               // rows r <-> t = t0 - 16 + r; the deepest read (c3's taps) ends at NT + 34
    static constexpr int PLANE = CS_NP * ROWS * CS_ROW;
    static constexpr int TAB = 2 * 64;                 // one per-(signal, channel) table
    static constexpr int CONST_FLOATS = (CS_NP == 2 ? 2 : 1) * (TAB /* c1 */ + 3 * TAB /* c2 c3 film */ + 128 /* heads */) + TAB /* r -> c3 */;
    static constexpr int SP = NT * (int)sizeof(act_t) + 16;
    static constexpr size_t LDS = CONST_FLOATS * 4 + 8 * (size_t)PLANE;
    static_assert(2 * C1_C * SP <= 4 * PLANE, "staging rows must fit the buffer they alias");
};

// one k=3 layer of (signal, output tile m): NC K chunks, all NTL time tiles, swapped operands (see cs_layer)
//   KIND 0: lrelu -> own signal's planes;  1: raw, accumulator starts from racc * kr;  2: lrelu -> the 2C-channel planes
template <int NTL, int NC, int KIND>
__device__ __forceinline__ void c1_layer(const unsigned char* in_planes, unsigned char* out_planes, int lo_off, int plane_bytes,
                                         const CsW* W /* [NC][3] */, int dil, int out_row0, const float* kb, const float* kiv,
                                         const f32x4* racc, const float* kr, int sig, int m, int t0, int Tv, int lane) {
    constexpr int G = 2;
    const int l15 = lane & 15, g = lane >> 4;
    int aoff[3];
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap) aoff[tap] = cs_off(out_row0 + (tap - 1) * dil + l15, g);
    const int co0 = m * 16 + 4 * g;
    const f32x4 kbv = *reinterpret_cast<const f32x4*>(kb + co0);
    f32x4 kivv = kbv, krv = kbv;
    if constexpr (CS_NP == 2) kivv = *reinterpret_cast<const f32x4*>(kiv + co0);
    if constexpr (CS_NP == 2 && KIND == 1) krv = *reinterpret_cast<const f32x4*>(kr + co0);
    int wbase;
    if constexpr (KIND == 2) {
        const int cc0 = sig * C1_C + co0;
        wbase = (cc0 >> 5) * plane_bytes + cs_off(out_row0 + l15, (cc0 & 31) >> 3) + (cc0 & 7) * 2;
    } else {
        wbase = (sig * C1_NC + (co0 >> 5)) * plane_bytes + cs_off(out_row0 + l15, (co0 & 31) >> 3) + (co0 & 7) * 2;
    }
    const int tbase = t0 - 16 + out_row0 + l15;
    #pragma unroll
    for (int j0 = 0; j0 < NTL; j0 += G) {
        f32x4 acc[G];
        #pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            // KIND 1, bfloat16 storage: the residual conv's finished tile (an MFMA result) is the initial value; float32
            // storage: it needs a multiply first, and a VALU result must not be an MFMA's C operand (see p0_layer_chunk):
            // the products start from zero (an inline constant) and the scaled tile joins after them
            if constexpr (KIND == 1) {
                if constexpr (CS_NP == 2) acc[ii] = f32x4{0.f, 0.f, 0.f, 0.f}; else acc[ii] = racc[j0 + ii];
            } else acc[ii] = kbv;
        }
        #pragma unroll
        for (int ch = 0; ch < NC; ++ch)
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap)
                #pragma unroll
                for (int ii = 0; ii < G; ++ii) {
                    const CsFrag a = cs_read(in_planes + ch * plane_bytes, aoff[tap] + (j0 + ii) * 16 * CS_ROW, lo_off);
                    acc[ii] = cs_prod<true>(W[ch * 3 + tap], a, acc[ii]);
                }
        #pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const unsigned keep = (unsigned)(tbase + 16 * (j0 + ii)) < (unsigned)Tv ? 0xffffffffu : 0u;
            f32x4 v = acc[ii];
            if constexpr (CS_NP == 2 && KIND == 1) v = racc[j0 + ii] * krv + v;
            if constexpr (CS_NP == 2) v = v * kivv;
            if constexpr (KIND != 1) v = cs_lrelu4(v);
            cs_store4_masked(out_planes + wbase + (j0 + ii) * 16 * CS_ROW, lo_off, v, keep);
        }
    }
}

template <int NTL>
__global__ __launch_bounds__(C1_NTHREADS, CS_NP == 1 ? 3 : 1)
void cond_stage1_kernel(const CondStage1Params p) {
    using GEO = C1Geom<NTL>;
    constexpr int NT = GEO::NT, ROWS = GEO::ROWS, PLANE = GEO::PLANE, NTO = GEO::NTO, TAB = GEO::TAB;
    constexpr int lo_off = ROWS * CS_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // tables [signal][64 channels] (heads: [128]): accumulator initial values (bias x accumulator scale) and, float32
    // storage, the factors that move an accumulator to its output tile's scale
    float* k1b = reinterpret_cast<float*>(smem);                                  // c1
    float* klb = k1b + TAB;                                                       // [3] c2, c3 (+ residual conv's bias), film.conv
    float* k5b = klb + 3 * TAB;                                                   // heads [128]
    float* krr = k5b + 128;                                                       // residual conv's tile -> c3's accumulator scale
    float* k1i = krr + TAB;                                                       // float32 storage only from here
    float* kli = k1i + TAB;
    float* k5i = kli + 3 * TAB;
    unsigned char* bufA = smem + GEO::CONST_FLOATS * 4;                           // 4 planes: c1 -> h -> output staging
    unsigned char* bufB = bufA + 4 * PLANE;                                       // 4 planes: x (raw, lrelu) x 2 signals -> c2 -> u (3 planes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int Tv = p.lens ? p.lens[b] * p.len_mul : p.T;
    const int ntx = (Tv + NT - 1) / NT;
    const int tpw = p.tpw & 0xffff;
    const int tile_begin = blockIdx.x * tpw;
    const int tile_end = min(tile_begin + tpw, ntx);
    if (tile_begin >= tile_end) return;
    const int sig = wave / C1_NM, m = wave - sig * C1_NM;

    // ---- operand scales (float32 storage; see cond_stage0_kernel) and tables ----
    float sx[2] = {1.f, 1.f}, sc[4][2] = {{1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}};
    if constexpr (CS_NP == 2) {
        float bu = 0.f;
        #pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float ax = amax_read(p.amax_in, s * p.B + b);
            // per-channel bounds alpha[c] * ax + beta[c] (packer), the tile's scale from the largest of them: lane = channel,
            // a wave-wide max (every wave computes the same: no exchange through LDS, no barrier)
            float bt[4];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = lane < C1_C ? p.cbnd[s][(t * 2 + 0) * C1_C + lane] * ax + p.cbnd[s][(t * 2 + 1) * C1_C + lane] : 0.f;
                bt[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, fmaxf(v, 0.f))), 63));
            }
            bu = fmaxf(bu, bt[3]);
            sx[s] = hx_scale_for(ax);
            sc[0][s] = hx_scale_for(bt[0]); sc[1][s] = hx_scale_for(bt[1]); sc[2][s] = hx_scale_for(bt[2]);
        }
        sc[3][0] = sc[3][1] = hx_scale_for(bu);
    }
    for (int i = tid; i < TAB; i += C1_NTHREADS) {
        const int s = i >> 6, c = i & 63;
        const bool ok = c < C1_C;
        float a1 = 1.f, ar = 1.f;                          // accumulator scales of the pair: input scale / inverse weight scale
        if constexpr (CS_NP == 2) {
            a1 = sx[s] / (ok ? p.winv1[s][c] : 1.f);
            ar = sx[s] / (ok ? p.winv1[s][C1_C + c] : 1.f);
            k1i[i] = ok ? sc[0][s] / a1 : 0.f;
        }
        k1b[i] = ok ? p.b1[s][c] * a1 : 0.f;
        float a3 = 1.f;
        #pragma unroll
        for (int l = 0; l < 3; ++l) {
            float as = 1.f;
            if constexpr (CS_NP == 2) {
                as = sc[l][s] / (ok ? p.winv[l][s][c] : 1.f);
                kli[l * TAB + i] = ok ? sc[l + 1][s] / as : 0.f;
            }
            if (l == 1) a3 = as;
            // (c3's own bias joins the residual conv's tile, which starts from both: see krr)
            klb[l * TAB + i] = ok && l != 1 ? p.bias[l][s][c] * as : 0.f;
        }
        // the residual conv's accumulator starts from (its bias + c3's bias) at ITS scale; krr moves the finished tile to
        // c3's accumulator scale (1 in bfloat16 storage)
        klb[1 * TAB + i] = ok ? (p.br[s][c] + p.bias[1][s][c]) * ar : 0.f;
        krr[i] = ok ? a3 / ar : 0.f;
    }
    for (int i = tid; i < 128; i += C1_NTHREADS) {
        const bool ok5 = i < 2 * C1_C;
        float as = 1.f;
        if constexpr (CS_NP == 2) {
            as = sc[3][0] / (ok5 ? p.winv5[i] : 1.f);
            k5i[i] = ok5 ? 1.f / as : 0.f;
        }
        k5b[i] = ok5 ? p.b5[i] * as : 0.f;
    }

    const __amdgpu_buffer_rsrc_t xr0 = act_rsrc(reinterpret_cast<const float*>(p.x), (long)b * p.x_b, (long)C1_CIN * p.ldx);
    const __amdgpu_buffer_rsrc_t xr1 = act_rsrc(reinterpret_cast<const float*>(p.x), p.x_sig + (long)b * p.x_b, (long)C1_CIN * p.ldx);
    const __amdgpu_buffer_rsrc_t ssr = act_rsrc(reinterpret_cast<const float*>(p.ss), (long)b * p.ss_b, (long)2 * C1_C * p.ld);
    const int hdTv = p.hd ? Tv / p.hd_s : 0;
    const __amdgpu_buffer_rsrc_t hdr0 = act_rsrc(reinterpret_cast<const float*>(p.hd ? p.hd : p.ss), p.hd ? (long)b * p.hd_b : 0, p.hd ? (long)C1_C * p.hd_ld : 0);
    const __amdgpu_buffer_rsrc_t hdr1 = act_rsrc(reinterpret_cast<const float*>(p.hd ? p.hd : p.ss), p.hd ? p.hd_sig + (long)b * p.hd_b : 0, p.hd ? (long)C1_C * p.hd_ld : 0);

    // ---- the stage's input tile: item = (signal, octet of 8 channels, quad of 4 rows), rows 4 .. ROWS ----
    constexpr int NQD = (ROWS - 4) / 4;
    const bool has_item = tid < 2 * 3 * NQD;
    const int it_s = tid / (3 * NQD), it_r = tid - it_s * 3 * NQD;
    const int it_q = it_r / 3, it_o = it_r - it_q * 3;
    f32x4 px[8];
    auto xfetch = [&](int t0n) {
        const int t = t0n - 16 + 4 + 4 * it_q;
        const bool tok = has_item && (unsigned)t < (unsigned)Tv;
        #pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int o = tok ? ((it_o * 8 + c) * p.ldx + t) * 4 : OOB_OFF;
            px[c] = it_s ? act_load4(xr1, o, 0) : act_load4(xr0, o, 0);
        }
    };
    auto xcommit = [&]() {
        if (has_item) {
            const float s_in = sx[it_s];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 lo4 = {px[0][j], px[1][j], px[2][j], px[3][j]}, hi4 = {px[4][j], px[5][j], px[6][j], px[7][j]};
                if constexpr (CS_NP == 2) { lo4 = lo4 * s_in; hi4 = hi4 * s_in; }
                unsigned char* dst = bufB + (it_s * 2) * PLANE + cs_off(4 + 4 * it_q + j, it_o);
                cs_store8_masked(dst, lo_off, lo4, hi4, 0xffffffffu);                             // raw
                cs_store8_masked(dst + PLANE, lo_off, cs_lrelu4(lo4), cs_lrelu4(hi4), 0xffffffffu);  // LeakyReLU'd
            }
        }
        // channel padding 24 .. 31 of the four input planes (c2 / u lay there), and - float32 storage - the padding
        // 48 .. 63 of c1's second planes (the staging rows lay there: float32 bits are not finite binary16 values)
        for (int i = tid; i < 4 * (ROWS - 4); i += C1_NTHREADS) {
            const int pl = i / (ROWS - 4), r = 4 + i - pl * (ROWS - 4);
            unsigned char* pad = bufB + pl * PLANE + cs_off(r, 3);
            *reinterpret_cast<u32x4*>(pad) = u32x4{0u, 0u, 0u, 0u};
            if constexpr (CS_NP == 2) *reinterpret_cast<u32x4*>(pad + lo_off) = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // channel padding 48 .. 63 of c1's second planes: the staged output rows of the tile before lay there (float32 bits
    // are not finite binary16 values; on the first tile the planes are whatever LDS held) - zeroed in P1, i.e. behind the
    // barrier that ends the copy-out and in front of the one c2 waits for
    auto zero_c1_padding = [&]() {
        for (int i = tid; i < 2 * 2 * ROWS; i += C1_NTHREADS) {
            const int s = i / (2 * ROWS), rr = i - s * 2 * ROWS, r = rr >> 1, o = 2 + (rr & 1);
            unsigned char* pad = bufA + (s * 2 + 1) * PLANE + cs_off(r, o);
            *reinterpret_cast<u32x4*>(pad) = u32x4{0u, 0u, 0u, 0u};
            if constexpr (CS_NP == 2) *reinterpret_cast<u32x4*>(pad + lo_off) = u32x4{0u, 0u, 0u, 0u};
        }
    };

    // ---- weight fragments: two register sets that trade places every tile ----
    const unsigned char* wl1 = reinterpret_cast<const unsigned char*>(p.w1[sig]);
    const unsigned char* wl[3] = {reinterpret_cast<const unsigned char*>(p.w[0][sig]), reinterpret_cast<const unsigned char*>(p.w[1][sig]),
                                  reinterpret_cast<const unsigned char*>(p.w[2][sig])};
    const unsigned char* wl5 = reinterpret_cast<const unsigned char*>(p.w5) + (long)(wave / 3) * 27 * CS_NP * CS_FRAG;
    const int m5t = wave % 3;
    auto load_pair = [&](CsW (&W)[9]) {                    // slots w0 w1 w2 | w1x1
        #pragma unroll
        for (int sl = 0; sl < 4; ++sl) W[sl] = cs_wload(wl1 + (long)(sl * 3 + m) * CS_NP * CS_FRAG, lane);
    };
    auto load_layer = [&](CsW (&W)[9], int l) {
        #pragma unroll
        for (int q = 0; q < 6; ++q) W[q] = cs_wload(wl[l] + (long)(q * 3 + m) * CS_NP * CS_FRAG, lane);
    };
    auto load_heads = [&](CsW (&W)[9]) {
        #pragma unroll
        for (int q = 0; q < 9; ++q) W[q] = cs_wload(wl5 + (long)(q * 3 + m5t) * CS_NP * CS_FRAG, lane);
    };
    float hmax[2] = {0.f, 0.f};
    const int l15 = lane & 15, g4 = lane >> 4;

    auto do_tile = [&](CsW (&WA)[9], CsW (&WB)[9], int tile) {
        const int t0 = tile * NT;
        // ---- P1: the pair on the stage's input: c1 = lrelu(conv3(lrelu(x)) + b1) -> A; r = conv1x1(x) (+ biases) -> registers ----
        load_layer(WB, 0);
        zero_c1_padding();
        f32x4 racc[NTL];
        {
            const unsigned char* xraw = bufB + (sig * 2) * PLANE;
            const unsigned char* xact = xraw + PLANE;
            const int co0 = m * 16 + 4 * g4;
            const f32x4 kbv = *reinterpret_cast<const f32x4*>(k1b + sig * 64 + co0);
            const f32x4 rbv = *reinterpret_cast<const f32x4*>(klb + 1 * TAB + sig * 64 + co0);
            f32x4 kivv = kbv;
            if constexpr (CS_NP == 2) kivv = *reinterpret_cast<const f32x4*>(k1i + sig * 64 + co0);
            int aoff[3];
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap) aoff[tap] = cs_off(8 + (tap - 1) + l15, g4);
            const int roff = cs_off(14 + l15, g4);
            const int wbase = (sig * C1_NC + (co0 >> 5)) * PLANE + cs_off(8 + l15, (co0 & 31) >> 3) + (co0 & 7) * 2;
            #pragma unroll
            for (int j0 = 0; j0 < NTL; j0 += 2) {
                f32x4 acc[2] = {kbv, kbv};
                #pragma unroll
                for (int tap = 0; tap < 3; ++tap)
                    #pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const CsFrag a = cs_read(xact, aoff[tap] + (j0 + ii) * 16 * CS_ROW, lo_off);
                        acc[ii] = cs_prod<true>(WA[tap], a, acc[ii]);
                    }
                #pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const CsFrag a = cs_read(xraw, roff + (j0 + ii) * 16 * CS_ROW, lo_off);
                    racc[j0 + ii] = cs_prod<true>(WA[3], a, rbv);
                }
                #pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const unsigned keep = (unsigned)(t0 - 16 + 8 + l15 + 16 * (j0 + ii)) < (unsigned)Tv ? 0xffffffffu : 0u;
                    f32x4 v = acc[ii];
                    if constexpr (CS_NP == 2) v = v * kivv;
                    cs_store4_masked(bufA + wbase + (j0 + ii) * 16 * CS_ROW, lo_off, cs_lrelu4(v), keep);
                }
            }
        }
        __syncthreads();
        // ---- P2: c2 = lrelu(conv3_d2(c1) + b2) -> B ----
        load_layer(WA, 1);
        c1_layer<NTL, C1_NC, 0>(bufA + sig * C1_NC * PLANE, bufB, lo_off, PLANE, WB, 2, 10, klb + 0 * TAB + sig * 64, kli + 0 * TAB + sig * 64,
                                nullptr, nullptr, sig, m, t0, Tv, lane);
        __syncthreads();
        // ---- P3: h = conv3_d4(c2) + b3 + r -> A ----
        load_layer(WB, 2);
        c1_layer<NTL, C1_NC, 1>(bufB + sig * C1_NC * PLANE, bufA, lo_off, PLANE, WA, 4, 14, klb + 1 * TAB + sig * 64, kli + 1 * TAB + sig * 64,
                                racc, krr + sig * 64, sig, m, t0, Tv, lane);
        __syncthreads();
        // ---- P4: u = lrelu(conv3_d1(h) + b4) -> the 96-channel planes of B; h[::s'] -> hd; next input tile requested ----
        load_heads(WA);
        c1_layer<NTL, C1_NC, 2>(bufA + sig * C1_NC * PLANE, bufB, lo_off, PLANE, WB, 1, 15, klb + 2 * TAB + sig * 64, kli + 2 * TAB + sig * 64,
                                nullptr, nullptr, sig, m, t0, Tv, lane);
        if (p.hd) {
            const int j_lo = (t0 + p.hd_s - 1) / p.hd_s;
            const int j_hi = min((min(t0 + NT, Tv) + p.hd_s - 1) / p.hd_s, hdTv);
            constexpr int NR = 2 * C1_C / C1_NWAVES;
            for (int j = j_lo + lane; j < j_hi; j += 64) {
                const int row = j * p.hd_s - t0 + 16;
                const unsigned char* bo[4];                  // the row's four slots (swizzled): immediates do the rest
                #pragma unroll
                for (int o = 0; o < 4; ++o) bo[o] = bufA + cs_off(row, o);
#ifdef FASTSVC_ACT_BF16
                unsigned short v[NR];
#else
                float v[NR];
                const float ih0 = 1.0f / sc[2][0], ih1 = 1.0f / sc[2][1];
#endif
                #pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int scn = wave + C1_NWAVES * k;
                    const int s = scn >= C1_C ? 1 : 0, c = scn - s * C1_C;
                    const unsigned char* src = bo[(c & 31) >> 3] + (s * C1_NC + (c >> 5)) * PLANE + (c & 7) * 2;
#ifdef FASTSVC_ACT_BF16
                    v[k] = *reinterpret_cast<const unsigned short*>(src);
#else
                    v[k] = ((float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + lo_off)) * (s ? ih1 : ih0);
                    hmax[s] = fmaxf(hmax[s], fabsf(v[k]));
#endif
                }
                #pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int scn = wave + C1_NWAVES * k;
                    const int s = scn >= C1_C ? 1 : 0, c = scn - s * C1_C;
#ifdef FASTSVC_ACT_BF16
                    __builtin_amdgcn_raw_buffer_store_b16(v[k], s ? hdr1 : hdr0, j * 2, c * p.hd_ld * 2, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), s ? hdr1 : hdr0, j * 4, c * p.hd_ld * 4, 0);
#endif
                }
            }
        }
        xfetch(t0 + NT);                                   // (past the last tile: out of range, nothing is fetched)
        __syncthreads();
        // ---- P5: [scale ; shift] = conv3_d1([u_lft ; u_sine]) + b5: wave = one of the six 16-channel output tiles ----
        load_pair(WB);
        {
            int aoff[3];
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap) aoff[tap] = cs_off(16 + (tap - 1) + l15, g4);
            const float bias = k5b[wave * 16 + l15];
            float oinv = 1.f;
            if constexpr (CS_NP == 2) oinv = k5i[wave * 16 + l15];
            unsigned char* srow = bufA + (wave * 16 + l15) * GEO::SP + (4 * g4) * (int)sizeof(act_t);
            constexpr int G5 = 4;
            #pragma unroll
            for (int i0 = 0; i0 < NTO; i0 += G5) {
                f32x4 acc[G5];
                #pragma unroll
                for (int ii = 0; ii < G5; ++ii) acc[ii] = f32x4{bias, bias, bias, bias};
                #pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    #pragma unroll
                    for (int tap = 0; tap < 3; ++tap)
                        #pragma unroll
                        for (int ii = 0; ii < G5; ++ii) {
                            if (i0 + ii >= NTO) continue;
                            const CsFrag a = cs_read(bufB + ch * PLANE, aoff[tap] + (i0 + ii) * 16 * CS_ROW, lo_off);
                            acc[ii] = cs_prod<false>(WA[ch * 3 + tap], a, acc[ii]);
                        }
                #pragma unroll
                for (int ii = 0; ii < G5; ++ii) {
                    if (i0 + ii >= NTO) continue;
                    f32x4 v = acc[ii];
                    if constexpr (CS_NP == 2) v = v * oinv;
                    unsigned char* dst = srow + (i0 + ii) * 16 * (int)sizeof(act_t);
#ifdef FASTSVC_ACT_BF16
                    *reinterpret_cast<cs4*>(dst) = __builtin_convertvector(v, cs4);
#else
                    *reinterpret_cast<f32x4*>(dst) = v;
#endif
                }
            }
        }
        __syncthreads();
        // ---- P0 of the next tile (its input -> B: every reader of u is past the barrier), then P6: staged rows -> ss ----
        xcommit();
        {
            constexpr int CH = NT * (int)sizeof(act_t) / 16;
            constexpr int EPC = 16 / (int)sizeof(act_t);
            constexpr int LPR = CH <= 16 ? 16 : CH <= 32 ? 32 : 64;
            static_assert(CH <= 64, "chunks of a row fit the lanes that copy it");
            constexpr int RPP = C1_NTHREADS / LPR;
            static_assert((2 * C1_C) % RPP == 0, "whole passes over the staged rows");
            const int k = tid % LPR, r0 = tid / LPR;
            const int t = t0 + k * EPC;
            const bool ok = k < CH && t < Tv;
            const unsigned char* src = bufA + r0 * GEO::SP + k * 16;
            const int o = ok ? (r0 * p.ld + t) * (int)sizeof(act_t) : OOB_OFF;
            #pragma unroll
            for (int i = 0; i < 2 * C1_C / RPP; ++i) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(src + i * RPP * GEO::SP);
                __builtin_amdgcn_raw_buffer_store_b128(w, ssr, o, i * RPP * p.ld * (int)sizeof(act_t), 0);
            }
        }
        __syncthreads();
    };

    CsW Wx[9], Wy[9];
    xfetch(tile_begin * NT);
    load_pair(Wx);
    __syncthreads();                                       // tables
    xcommit();
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; tile += 2) {
        do_tile(Wx, Wy, tile);
        if (tile + 1 < tile_end) do_tile(Wy, Wx, tile + 1);
    }
    if constexpr (CS_NP == 2) {
        if (p.amax_hd && p.hd) {
            #pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)wave_max_u32_lane63(__builtin_bit_cast(unsigned, hmax[s])), 63);
                if (lane == 0 && a != 0u)
                    atomicMax(reinterpret_cast<unsigned*>(p.amax_hd) + (s * p.B + b) * AMAX_ENTRY + ((blockIdx.x + wave) & (AMAX_W - 1)) * AMAX_STRIDE, a);
            }
        }
    }
}

template <int NTL>
static hipError_t cond_stage1_instance(const CondStage1Params& p, hipStream_t stream) {
    using GEO = C1Geom<NTL>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_stage1_kernel<NTL>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS);
    if (attr != hipSuccess) return attr;
    const int ntx = (p.T + GEO::NT - 1) / GEO::NT;
    const int tpw = p.tpw & 0xffff;
    dim3 grid((ntx + tpw - 1) / tpw, 1, p.B);
    hipLaunchKernelGGL(cond_stage1_kernel<NTL>, grid, dim3(C1_NTHREADS), GEO::LDS, stream, p);
    return hipGetLastError();
}

#ifdef FASTSVC_ACT_BF16
// =====================================================================================================================
// Stage 1 (C = 48, C_in = 24) as a LAYER PIPELINE - bfloat16 storage only (the float32 instance's hi + lo planes do not fit
// three-chunk rings).  Same scheme as cond_stage0_pipe_kernel; what differs:
//   * 12 waves: 0-1 stage-in of a signal's compact input (raw -> a 6-chunk ring for the 1x1 residual conv, LeakyReLU'd -> the
//     3-chunk ring c1 reads; loads requested two steps ahead) AND its c1; 2-3 c2; 4-5 c3 (starts from the 1x1 residual conv of
//     the raw tile: the phase kernel's order of summation); 6-7 film.conv; 8-10 two 16-channel tiles of the heads each; 11 copy-out.
//   * a layer wave holds ALL THREE 16-channel output tiles of its layer (18-21 weight fragments resident - the phase kernel
//     streams 31 per wave and tile from L2) and feeds them from ONE activation fragment: a third of the fragment reads.
//   * the 48-channel tensors c1 / c2 / h live in planes of 96-BYTE rows (64-byte rows pad 48 channels to 64: 19 planes would
//     not fit): bank = (24 row + 4 slot) mod 64 is conflict-free for ds_read_b128 at any row offset without a swizzle; the
//     second K chunk of a row reads 32 bytes into the next row - finite values under zero weight rows.
//   * chunks of 32 columns (N = 2); the heads' stream lags 5 tiles = 5 chunks incl. alignment behind the stage-in.
constexpr int Q1_NWAVES = 12, Q1_NTHREADS = Q1_NWAVES * 64;
struct Q1Geom {
    static constexpr int N = 2, NT = 32, RING = 96, GUARD = 8, PROWS = RING + 2 * GUARD;
    static constexpr int P64 = PROWS * 64, P96 = PROWS * 96;
    static constexpr int XRT = 12;                         // tiles of the raw-input ring (the 1x1 conv reads 3 steps + 3 tiles behind)
    static constexpr int XRAW = XRT * 16 * 64;
    static constexpr int SP = NT * 2 + 16;                 // staging pitch
    static constexpr int STG = 2 * C1_C * SP;
    static constexpr int LAG = 5;                          // chunks of the heads' stream in front of the first output column
    static constexpr int TAB = 128;
    static constexpr int CONST_FLOATS = TAB + 3 * TAB + 128;
    // planes in this order: xact[2] (P64), u[3] (P64), xraw[2], c1[2] c2[2] h[2] (P96), 64 bytes of zeros, staging x 2, dummy
    static constexpr int O_XACT = 0, O_U = O_XACT + 2 * P64, O_XRAW = O_U + 3 * P64, O_C1 = O_XRAW + 2 * XRAW, O_C2 = O_C1 + 2 * P96,
                         O_H = O_C2 + 2 * P96, O_STG = O_H + 2 * P96 + 64, O_DUMMY = O_STG + 2 * STG, O_END = O_DUMMY + 64 * 32;
    static constexpr size_t LDS = CONST_FLOATS * 4 + (size_t)O_END;
};

// one chunk (2 tiles) of a stage-1 layer: all three 16-channel output tiles per activation fragment, swapped operands
//   KIND 0: lrelu -> a 96-byte-row plane (c1, c2);  1: c3 = raw, accumulator starts from the 1x1 residual conv of the raw input tile;
//        2: lrelu -> the 96-channel planes [lft ; sine] of the heads (64-byte rows, swizzled)
//   NC: K chunks of the input; IN96: input plane has 96-byte rows (else one swizzled 64-byte-row plane)
template <int POS, int NC, bool IN96, int KIND>
__device__ __forceinline__ void q1_layer_chunk(const unsigned char* in_plane, unsigned char* out_base, const CsW* W /* [NC][3][3] */,
                                               const CsW* W1 /* [3], KIND 1 */, const unsigned char* xraw_tile0, const unsigned char* xraw_tile1,
                                               const int (&rd)[3], const int (&wr)[3], const f32x4 (&kbv)[3], int tstart, int Tv, int lane,
                                               unsigned char* dummy) {
    using GEO = Q1Geom;
    constexpr int N = GEO::N, NTL3 = 3 * N;
    constexpr int TS_IN = 16 * (IN96 ? 96 : 64), TS_OUT = 16 * (KIND == 2 ? 64 : 96), GOFF = GEO::RING * (KIND == 2 ? 64 : 96);
    const int l15 = lane & 15, g = lane >> 4;
    f32x4 acc[N][3];
    #pragma unroll
    for (int i = 0; i < N; ++i)
        #pragma unroll
        for (int m = 0; m < 3; ++m) acc[i][m] = kbv[m];
    if constexpr (KIND == 1) {
        #pragma unroll
        for (int i = 0; i < N; ++i) {
            CsFrag xr;
            xr.p[0] = *reinterpret_cast<const cs8*>((i ? xraw_tile1 : xraw_tile0) + cs_off(l15, g));
            #pragma unroll
            for (int m = 0; m < 3; ++m) acc[i][m] = cs_prod<true>(W1[m], xr, acc[i][m]);
        }
    }
    #pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        CsFrag a[N][3];
        #pragma unroll
        for (int i = 0; i < N; ++i) {
            const int PIN = (POS * N + i - 1 + NTL3) % NTL3;
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap) a[i][tap].p[0] = *reinterpret_cast<const cs8*>(in_plane + rd[tap] + ch * 64 + PIN * TS_IN);
        }
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            #pragma unroll
            for (int i = 0; i < N; ++i)
                #pragma unroll
                for (int m = 0; m < 3; ++m) acc[i][m] = cs_prod<true>(W[(ch * 3 + tap) * 3 + m], a[i][tap], acc[i][m]);
    }
    #pragma unroll
    for (int i = 0; i < N; ++i) {
        const int P = POS * N + i;
        const int tt = tstart + 16 * i;
        const bool edge = tt < 0 || tt + 16 > Tv;
        unsigned keep = 0xffffffffu;
        if (edge) keep = (unsigned)(tt + l15) < (unsigned)Tv ? 0xffffffffu : 0u;
        #pragma unroll
        for (int m = 0; m < 3; ++m) {
            f32x4 v = acc[i][m];
            if constexpr (KIND != 1) v = cs_lrelu4(v);
            unsigned char* dst = out_base + wr[m] + P * TS_OUT;
            const CsPk4 pk = cs_pack4(v, keep);
            cs_put4(dst, dst, pk);
            if (P == 0) { unsigned char* g2 = l15 < 8 ? dst + GOFF : dummy; cs_put4(g2, g2, pk); }
            if (P == NTL3 - 1) { unsigned char* g2 = l15 >= 8 ? dst - GOFF : dummy; cs_put4(g2, g2, pk); }
        }
    }
}

__global__ __launch_bounds__(Q1_NTHREADS, 1)
void cond_stage1_pipe_kernel(const CondStage1Params p) {
    using GEO = Q1Geom;
    constexpr int N = GEO::N, NT = GEO::NT, RING = GEO::RING, GUARD = GEO::GUARD, LAG = GEO::LAG, TAB = GEO::TAB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* k1b = reinterpret_cast<float*>(smem);           // [signal][64] c1 bias
    float* klb = k1b + TAB;                                // [3][signal][64]: c2, c3 (+ the 1x1 conv's bias), film.conv
    float* k5b = klb + 3 * TAB;                            // [128] heads
    unsigned char* L = smem + GEO::CONST_FLOATS * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int Tv = p.lens ? p.lens[b] * p.len_mul : p.T;
    const int Kc = p.tpw & 0xffff;
    const int T0 = blockIdx.x * Kc * NT;
    if (T0 >= Tv) return;
    const int nch = min(Kc, (Tv - T0 + NT - 1) / NT);
    const int T1 = min(T0 + nch * NT, Tv);
    const int Ktot = nch + LAG;
    const int nsteps = Ktot + 6;
    const int torg = T0 - 16 * LAG;                        // time of the stage-in's first column (tile Q)
    for (int i = tid; i < TAB; i += Q1_NTHREADS) {
        const int s = i >> 6, c = i & 63;
        const bool ok = c < C1_C;
        k1b[i] = ok ? p.b1[s][c] : 0.f;
        klb[0 * TAB + i] = ok ? p.bias[0][s][c] : 0.f;
        klb[1 * TAB + i] = ok ? p.br[s][c] + p.bias[1][s][c] : 0.f;
        klb[2 * TAB + i] = ok ? p.bias[2][s][c] : 0.f;
        k5b[i] = i < 2 * C1_C ? p.b5[i] : 0.f;
    }
    for (int o = tid * 16; o < GEO::O_END; o += Q1_NTHREADS * 16) *reinterpret_cast<u32x4*>(L + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int l15 = lane & 15, g = lane >> 4;
    unsigned char* dummy = L + GEO::O_DUMMY + lane * 32;

    if (wave < 2) {
        // ================= stage-in of this signal's input + c1 = lrelu(conv3(lrelu(x)) + b1) =================
        const int sig = wave;
        const __amdgpu_buffer_rsrc_t xr = act_rsrc(reinterpret_cast<const float*>(p.x), sig * p.x_sig + (long)b * p.x_b, (long)C1_CIN * p.ldx);
        const bool has_item = lane < 24;
        const int it_q = lane / 3, it_o = lane - it_q * 3;     // rows 4 q .. 4 q + 3 of the chunk, octet o
        unsigned char* xact = L + GEO::O_XACT + sig * GEO::P64;
        unsigned char* xraw = L + GEO::O_XRAW + sig * GEO::XRAW;
        cs_u2 px[3][8];                                        // three chunks in flight: requested two steps before their commit
        auto request = [&](cs_u2 (&set)[8], int k0) {
            const int t = torg + k0 * NT + 4 * it_q;
            const bool tok = has_item && (unsigned)t < (unsigned)Tv && k0 < Ktot;
            #pragma unroll
            for (int c = 0; c < 8; ++c)
                set[c] = __builtin_amdgcn_raw_buffer_load_b64(xr, tok ? ((it_o * 8 + c) * p.ldx + t) * 2 : OOB_OFF, 0, 0);
        };
        CsW W[9];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            #pragma unroll
            for (int m = 0; m < 3; ++m)
                W[tap * 3 + m] = cs_wload(reinterpret_cast<const unsigned char*>(p.w1[sig]) + (long)(tap * 3 + m) * CS_FRAG, lane);
        int rd[3], wr[3];
        f32x4 kbv[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) rd[tap] = cs_off(GUARD + (tap - 1) + l15, g);
        #pragma unroll
        for (int m = 0; m < 3; ++m) {
            wr[m] = (GUARD + l15) * 96 + (16 * m + 4 * g) * 2;
            kbv[m] = *reinterpret_cast<const f32x4*>(k1b + sig * 64 + 16 * m + 4 * g);
        }
        request(px[0], 0);
        request(px[1], 1);
        int xrt = 0;                                           // tile of the raw ring the next chunk goes to
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            if (s < Ktot) {
                // ---- stage-in of chunk s: raw -> the 12-tile ring, LeakyReLU'd -> position J of the 3-chunk ring ----
                request(px[(J + 2) % 3], s + 2);
                if (has_item) {
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned short e[8];
                        #pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned w = j < 2 ? px[J][c].x : px[J][c].y;
                            e[c] = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
                        }
                        const u32x4 raw = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                           (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
                        f32x4 lo4, hi4;
                        #pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            lo4[c] = __builtin_bit_cast(float, (unsigned)e[c] << 16);
                            hi4[c] = __builtin_bit_cast(float, (unsigned)e[4 + c] << 16);
                        }
                        const CsPk8 act = cs_pack8(cs_lrelu4(lo4), cs_lrelu4(hi4), 0xffffffffu);
                        const int rr = 4 * it_q + j;
                        *reinterpret_cast<u32x4*>(xraw + cs_off(xrt * 16 + rr, it_o)) = raw;
                        unsigned char* dst = xact + cs_off(GUARD + J * NT + rr, it_o);
                        cs_put8(dst, dst, act);
                        if (J == 0) { unsigned char* g2 = rr < 8 ? dst + RING * 64 : dummy; cs_put8(g2, g2, act); }
                        if (J == 2) { unsigned char* g2 = rr >= NT - 8 ? dst - RING * 64 : dummy; cs_put8(g2, g2, act); }
                    }
                }
                xrt = xrt + N >= GEO::XRT ? 0 : xrt + N;
            }
            const int k = s - 1;
            if (k >= 0 && k < Ktot)
                q1_layer_chunk<(J + 2) % 3, 1, false, 0>(xact, L + GEO::O_C1 + sig * GEO::P96, W, nullptr, nullptr, nullptr, rd, wr, kbv,
                                                         torg + 16 * (k * N - 1), Tv, lane, dummy);
        });
    } else if (wave < 8) {
        // ================= c2 / c3 / film.conv of one signal =================
        const int Lr = 2 + ((wave - 2) >> 1), sig = wave & 1;   // layer 2 c2, 3 c3, 4 film.conv
        const int li = Lr - 2;
        CsW W[18], W1[3];
        #pragma unroll
        for (int q = 0; q < 6; ++q)
            #pragma unroll
            for (int m = 0; m < 3; ++m)
                W[q * 3 + m] = cs_wload(reinterpret_cast<const unsigned char*>(p.w[li][sig]) + (long)(q * 3 + m) * CS_FRAG, lane);
        #pragma unroll
        for (int m = 0; m < 3; ++m)
            W1[m] = Lr == 3 ? cs_wload(reinterpret_cast<const unsigned char*>(p.w1[sig]) + (long)(3 * 3 + m) * CS_FRAG, lane) : W[m];
        const int dil = Lr == 2 ? 2 : Lr == 3 ? 4 : 1;
        int rd[3], wr[3];
        f32x4 kbv[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) rd[tap] = (GUARD + (tap - 1) * dil + l15) * 96 + g * 16;
        #pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int co0 = 16 * m + 4 * g;
            kbv[m] = *reinterpret_cast<const f32x4*>(klb + li * TAB + sig * 64 + co0);
            if (Lr == 4) {
                const int cc0 = sig * C1_C + co0;
                wr[m] = (cc0 >> 5) * GEO::P64 + cs_off(GUARD + l15, (cc0 & 31) >> 3) + (cc0 & 7) * 2;
            } else wr[m] = (GUARD + l15) * 96 + co0 * 2;
        }
        const unsigned char* in_plane = L + (Lr == 2 ? GEO::O_C1 : Lr == 3 ? GEO::O_C2 : GEO::O_H) + sig * GEO::P96;
        unsigned char* out_base = L + (Lr == 2 ? GEO::O_C2 + sig * GEO::P96 : Lr == 3 ? GEO::O_H + sig * GEO::P96 : GEO::O_U);
        const unsigned char* xraw = L + GEO::O_XRAW + sig * GEO::XRAW;
        int xrt = GEO::XRT - 3;                                // c3's chunk 0 starts 3 tiles in front of the raw ring's tile 0
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            const int k = s - Lr;
            if (k >= 0 && k < Ktot) {
                const int tstart = torg + 16 * (k * N - Lr);
                if (Lr == 2) q1_layer_chunk<(J + 1) % 3, 2, true, 0>(in_plane, out_base, W, W1, nullptr, nullptr, rd, wr, kbv, tstart, Tv, lane, dummy);
                else if (Lr == 3) {
                    const int t1 = xrt + 1 >= GEO::XRT ? 0 : xrt + 1;
                    q1_layer_chunk<J, 2, true, 1>(in_plane, out_base, W, W1, xraw + xrt * 1024, xraw + t1 * 1024, rd, wr, kbv, tstart, Tv, lane, dummy);
                    xrt = t1 + 1 >= GEO::XRT ? 0 : t1 + 1;
                } else q1_layer_chunk<(J + 2) % 3, 2, true, 2>(in_plane, out_base, W, W1, nullptr, nullptr, rd, wr, kbv, tstart, Tv, lane, dummy);
            }
        });
    } else if (wave < 11) {
        // ================= heads: two of the six 16-channel output tiles per wave, both from one fragment =================
        const int hw = wave - 8;
        CsW W5[2][9];
        #pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int tile = 2 * hw + mm, gq = tile / 3, m5t = tile - gq * 3;
            #pragma unroll
            for (int q = 0; q < 9; ++q)
                W5[mm][q] = cs_wload(reinterpret_cast<const unsigned char*>(p.w5) + (long)(gq * 27 + q * 3 + m5t) * CS_FRAG, lane);
        }
        int rd[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap) rd[tap] = cs_off(GUARD + (tap - 1) + l15, g);
        const float bias0 = k5b[(2 * hw) * 16 + l15], bias1 = k5b[(2 * hw + 1) * 16 + l15];
        const unsigned char* upl = L + GEO::O_U;
        const int srow = ((2 * hw) * 16 + l15) * GEO::SP + 4 * g * 2;
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            const int k = s - 5;
            if (k >= LAG && k < Ktot) {
                constexpr int POS = (J + 1) % 3;               // (J - 5) mod 3
                unsigned char* sb = L + GEO::O_STG + (s & 1) * GEO::STG + srow;
                f32x4 acc[N][2];
                #pragma unroll
                for (int i = 0; i < N; ++i) { acc[i][0] = f32x4{bias0, bias0, bias0, bias0}; acc[i][1] = f32x4{bias1, bias1, bias1, bias1}; }
                #pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    CsFrag a[N][3];
                    #pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const int PIN = (POS * N + i - 1 + 3 * N) % (3 * N);
                        #pragma unroll
                        for (int tap = 0; tap < 3; ++tap) a[i][tap].p[0] = *reinterpret_cast<const cs8*>(upl + ch * GEO::P64 + rd[tap] + PIN * 1024);
                    }
                    #pragma unroll
                    for (int tap = 0; tap < 3; ++tap)
                        #pragma unroll
                        for (int i = 0; i < N; ++i)
                            #pragma unroll
                            for (int mm = 0; mm < 2; ++mm) acc[i][mm] = cs_prod<false>(W5[mm][ch * 3 + tap], a[i][tap], acc[i][mm]);
                }
                #pragma unroll
                for (int i = 0; i < N; ++i)
                    #pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
                        *reinterpret_cast<cs_u2*>(sb + mm * 16 * GEO::SP + i * 32) = __builtin_bit_cast(cs_u2, __builtin_convertvector(acc[i][mm], cs4));
            }
        });
    } else {
        // ================= copy-out: staged [scale ; shift] rows -> ss, h[::s'] -> hd =================
        const __amdgpu_buffer_rsrc_t ssr = act_rsrc(reinterpret_cast<const float*>(p.ss), (long)b * p.ss_b, (long)2 * C1_C * p.ld);
        const int hds = p.hd ? p.hd_s : 1;
        const int hdTv = (int)udiv_small((unsigned)Tv, hds);
        const __amdgpu_buffer_rsrc_t hdr = act_rsrc(reinterpret_cast<const float*>(p.hd ? p.hd : p.ss), p.hd ? (long)b * p.hd_b : 0,
                                                    p.hd ? p.hd_sig + (long)C1_C * p.hd_ld : 0);
        const int ck = lane & 3, cr = lane >> 2;               // 4 pieces of 16 bytes per staged row, 16 rows per pass
        const int jj = lane & 7, q = lane >> 3;                // hd: lane = (decimated column, 12 channels of one signal)
        const int hs = q >> 2, hc0 = (q & 3) * 12;
        p0_steps(nsteps, [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            {
                const int kk = s - 6 - LAG;
                if (kk >= 0 && kk < nch) {
                    const unsigned char* sb = L + GEO::O_STG + ((s - 1) & 1) * GEO::STG + cr * GEO::SP + ck * 16;
                    const int t = T0 + kk * NT + ck * 8;
                    const int o = t < Tv ? (cr * p.ld + t) * 2 : OOB_OFF;
                    #pragma unroll
                    for (int i = 0; i < 2 * C1_C / 16; ++i) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(sb + i * 16 * GEO::SP);
                        __builtin_amdgcn_raw_buffer_store_b128(w, ssr, o, i * 16 * p.ld * 2, 0);
                    }
                }
            }
            {
                const int k3 = s - 4;                           // the h chunk c3 finished in the step before
                if (p.hd && k3 >= 0 && k3 < Ktot) {
                    constexpr int POS = (J + 2) % 3;           // (J - 4) mod 3
                    const int th0 = torg + 16 * (k3 * N - 3);
                    const int ta = max(th0, T0), tb = min(th0 + NT, T1);
                    const int j_lo = (int)udiv_small((unsigned)(ta + hds - 1), hds);
                    const int j_hi = min((int)udiv_small((unsigned)(max(tb, ta) + hds - 1), hds), hdTv);
                    const int hrow0 = (int)(hs * p.hd_sig) + hc0 * p.hd_ld;
                    for (int j = j_lo + jj; j < j_hi; j += 8) {
                        const int row = GUARD + POS * NT + (j * hds - th0);
                        const unsigned char* src = L + GEO::O_H + hs * GEO::P96 + row * 96 + hc0 * 2;
                        cs_u2 hv[3];
                        #pragma unroll
                        for (int e = 0; e < 3; ++e) hv[e] = *reinterpret_cast<const cs_u2*>(src + 8 * e);
                        #pragma unroll
                        for (int e = 0; e < 3; ++e)
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const unsigned w = u < 2 ? hv[e].x : hv[e].y;
                                const unsigned short v = (unsigned short)((u & 1) ? (w >> 16) : (w & 0xffffu));
                                __builtin_amdgcn_raw_buffer_store_b16(v, hdr, (hrow0 + j) * 2, (4 * e + u) * p.hd_ld * 2, 0);
                            }
                    }
                }
            }
        });
    }
}

static hipError_t cond_stage1_pipe_instance(const CondStage1Params& p, hipStream_t stream) {
    using GEO = Q1Geom;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_stage1_pipe_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS);
    if (attr != hipSuccess) return attr;
    if (p.hd && (p.hd_sig + (long)C1_C * p.hd_ld) * 2L >= (1L << 31)) return hipErrorInvalidValue;
    if (((long)C1_CIN * p.ldx) * 2L >= (1L << 31)) return hipErrorInvalidValue;
    const int nchunks = (p.T + GEO::NT - 1) / GEO::NT;
    const int kc = p.tpw & 0xffff;
    dim3 grid((nchunks + kc - 1) / kc, 1, p.B);
    hipLaunchKernelGGL(cond_stage1_pipe_kernel, grid, dim3(Q1_NTHREADS), GEO::LDS, stream, p);
    return hipGetLastError();
}
#endif

hipError_t launch_cond_stage1(const CondStage1Params& p, hipStream_t stream) {
    if (p.C != C1_C || p.Cin != C1_CIN || (p.T % 8) != 0 || (p.ld % 8) != 0 || (p.ldx % 4) != 0 || (p.tpw & 0xffff) < 1 ||
        (p.hd && p.hd_s < 2)) return hipErrorInvalidValue;
#ifdef FASTSVC_ACT_BF16
    if (p.small == 2) return cond_stage1_pipe_instance(p, stream);
#else
    if (p.small == 2) return hipErrorInvalidValue;         // (bfloat16 storage only: Q1Geom)
#endif
    return p.small ? cond_stage1_instance<4>(p, stream) : cond_stage1_instance<8>(p, stream);
}

#ifndef FASTSVC_ACT_BF16
int cond_stage1_tile_columns(int small) { return small == 2 ? 32 : small ? C1Geom<4>::NT : C1Geom<8>::NT; }
#endif

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
