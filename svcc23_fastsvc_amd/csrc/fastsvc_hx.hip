// fastsvc_hx.hip - gfx950 (CDNA4 / MI355X) half-precision-MFMA convolution kernels of the FastSVC
// generator forward: the same k=3 dilated "same" convolutions, fused prologues and epilogues as
// fastsvc_kernels.hip (reference: Conv1d1x3 / Conv2d1x3, harana/layers/upsample.py:76-83,99-106, used by
// harana/models/fastsvc.py:56-75,164-178,209-218), with the implicit GEMM on the 16x16x32 matrix
// instruction instead of the f32-input one (1/16 of its rate):
//
//   float32 storage (namespace fastsvc):  SPLIT-HALF products, fp32-class results.  Every operand is split
//       into two binary16 pieces, x = xh + xl with xh = f16(x), xl = f16(x - xh): 22 significand bits WHERE
//       |x| >= 2^-3 - below that the low piece is a binary16 subnormal (absolute step 2^-24) - and nothing at
//       all above 65504.  binary16's range is therefore managed explicitly (ConvParams, "dynamic range"):
//       weights are scaled per output channel and activations per (tensor, utterance) by EXACT powers of two
//       so that the largest magnitude sits in [2^14, 2^15): the split then keeps 22 bits for everything within
//       2^-17 of the tensor's maximum and an absolute error of 2^-40 of the maximum below; the factors are
//       divided out in the epilogue by the FMA that adds the bias.  The activation factor comes from the
//       running max |value| the PRODUCING kernel recorded (amax slots) - or, behind an InstanceNorm, from the
//       bound sqrt(T) + |p| of a normalised row.  Then
//           x * w  ~=  xh*wh + xh*wl + xl*wh          (the dropped xl*wl term is < 2^-22 |x w|)
//       costs three v_mfma_f32_16x16x32_f16 per 32 input channels = 3/16 of the f32-input MFMA time.
//       End to end the generator output moves by 4e-6 against the fp32 path (oracle simulation and
//       tests/test_parity_gpu.py), i.e. it stays at the reference's own fp32 noise floor (1e-5), and
//       tests/test_dynamic_range_gpu.py holds that over inputs / weights scaled by 2^-20 .. 2^8.
//   bfloat16 storage (namespace fastsvc::bf16, -DFASTSVC_ACT_BF16):  ONE v_mfma_f32_16x16x32_bf16 product
//       of the bf16-rounded operands - BASELINE config 3's "bf16 generator forward".
//
// Data movement (what bounds these kernels: with the matrix work this cheap every layer is HBM-side):
//   * workgroup = 4 consumer + 4 producer waves, one barrier per unit = (time tile, 32-channel K chunk),
//     double-buffered LDS tile, p.tpw consecutive tiles per workgroup - the pipeline of conv_mfma_ws_kernel;
//   * LDS tile = TIME-major rows of 32 channels (64 B of f16 / bf16): a lane's MFMA A-fragment - 8
//     consecutive input channels at one time step - is ONE ds_read_b128, and a conv tap is a ROW offset
//     (always 16-byte aligned whatever the dilation).  Rows and 16-byte slots are swizzled (hx_lds_off) so that
//     both the consumers' ds_read_b128 at any row offset and the producers' ds_write_b128 are conflict-free;
//   * producers load float4 (4 time steps) of 8 channels per thread - 256 B contiguous per channel row and
//     wave instruction -, apply InstanceNorm / LeakyReLU, split, transpose 4 x 8 in registers and write
//     4 (+4) ds_write_b128;
//   * weights: pre-split fragments in MFMA operand order ([chunk][tap][16-channel tile][hi,lo][lane][8]),
//     streamed from L2 into a unit-deep register ring with buffer_load_dwordx4 (resident when C_in <= 32);
//   * epilogues: shared with the f32 kernels (fastsvc_device.inc);
//   * fused launches (MODE_CHAIN / MODE_CHAIN1): two k=3 convs back to back with the tensor between them in LDS
//     (c2 -> c3 of a conditioning stage, a whole stage 0 from the raw signal, the FiLM net of a stage), and
//     conv_last computed by the epilogue of the last block's final conv (ConvParams::last_w) - see conv_hx_kernel.
// Rows must be a multiple of 4 long (float4 everywhere); run_conv falls back to conv_mfma_ws_kernel otherwise.
#include "fastsvc_kernels.h"

namespace fastsvc {
#ifdef FASTSVC_ACT_BF16
namespace bf16 {
#endif

#include "fastsvc_device.inc"

#include "fastsvc_hx_common.inc"

// DIRECT unit: acc[n][m] += sum_tap X[t + (tap-1) d] W[tap].  aoff[tap]: this lane's byte offset of (row of
// time tile 0 shifted by the tap, its octet).
// OPT_LAST (MODE_CHAIN's first conv): transposed result tiles (hx_prod SWAP), and time tile NW-1 is computed only
// when `do_last` (wave-uniform; the extra tile of the last wave along time) - a scalar branch around that
// tile's MFMAs only (two whole copies of the unit under an if/else made hipcc spill).
// NS / SOFF / ADV: the ring holds NS fragments per unit of which this conv's start at slot SOFF; the last conv of the
// unit advances the ring (a launch with a second operand runs two convs per unit, ConvParams::x2).
template <int MW, int NW, bool RELOAD, bool OPT_LAST = false, int NS = 3 * MW * HX_NP, int SOFF = 0, bool ADV = true>
__device__ __forceinline__ void hx_unit_direct(f32x4 (&acc)[NW][MW], const unsigned char* tile, const int (&aoff)[3],
                                               int lo_off, HxWeightStream<NS>& ws, bool do_last = true) {
    constexpr int NSTEP = 3 * NW;
    // The A fragments are read PF steps ahead.  bfloat16 storage: a step is MW products = 16 MW cycles of the matrix
    // pipe, less than an LDS round trip - one step ahead the consumer stalled at every step (film.3.heads: 2.0k cycles
    // per unit for 1.15k of products); the split-binary16 steps (3 MW products) cover it one step ahead.
    constexpr int PF = HX_NP == 1 ? (MW >= 3 ? 2 : 3) : 1;
    HxFrag a[PF + 1];
    #pragma unroll
    for (int q = 0; q < PF && q < NSTEP; ++q) a[q] = hx_read(tile, aoff[q / NW] + (q % NW) * 16 * HX_ROW, lo_off);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int tap = s / NW, n = s % NW;
        if (s + PF < NSTEP) {
            a[(s + PF) % (PF + 1)] = hx_read(tile, aoff[(s + PF) / NW] + ((s + PF) % NW) * 16 * HX_ROW, lo_off);
            __builtin_amdgcn_sched_barrier(0);                 // reads stay ahead of the MFMAs
        }
        if (!OPT_LAST || n + 1 < NW || do_last) hx_step<MW, OPT_LAST>(acc[n], a[s % (PF + 1)], &ws.wr[SOFF + tap * MW * HX_NP]);
        if (RELOAD && n + 1 == NW) {                            // last use of this tap's fragments
            #pragma unroll
            for (int q = 0; q < MW * HX_NP; ++q) ws.request(SOFF + tap * MW * HX_NP + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (RELOAD && ADV) ws.advance();
}

// DIRECT unit whose weight fragments are read from LDS (`wl`: this unit's [tap][m][piece] KB-sized fragments,
// lane-linear).  MODE_CHAIN's second conv where its whole weight slice fits next to the tiles: streamed from L2
// its units follow each other without a barrier in between, so a unit-deep register ring has one unit's MFMA time
// (1.3k cycles for 3 x 3 tiles) to cover the L2 latency and the phase ran at 40 % of its matrix rate.
template <int MW, int NW>
__device__ __forceinline__ void hx_unit_direct_wl(f32x4 (&acc)[NW][MW], const unsigned char* tile, const int (&aoff)[3],
                                                  int lo_off, const unsigned char* wl, int lane) {
    constexpr int NSTEP = 3 * NW;
    HxFrag a[2];
    u32x4 w[2][MW * HX_NP];
    a[0] = hx_read(tile, aoff[0], lo_off);
    #pragma unroll
    for (int q = 0; q < MW * HX_NP; ++q) w[0][q] = *reinterpret_cast<const u32x4*>(wl + q * HX_FRAG + lane * 16);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int tap = s / NW, n = s % NW;
        if (s + 1 < NSTEP) {
            a[(s + 1) & 1] = hx_read(tile, aoff[(s + 1) / NW] + ((s + 1) % NW) * 16 * HX_ROW, lo_off);
            if (n + 1 == NW) {                                  // next step starts the next tap: its fragments
                #pragma unroll
                for (int q = 0; q < MW * HX_NP; ++q)
                    w[(tap + 1) & 1][q] = *reinterpret_cast<const u32x4*>(wl + ((tap + 1) * MW * HX_NP + q) * HX_FRAG + lane * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        hx_step<MW>(acc[n], a[s & 1], &w[tap & 1][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// POLY unit (Stretch2d + k=3 conv at the INPUT rate, fastsvc_kernels.h MODE_POLY): three accumulator sets
//   a += W0 x[j-1] - W0 x[j],   z += (W0+W1+W2) x[j],   c += W2 x[j+1] - W2 x[j]
// (the differences of the f32 kernel become a second product with the negated fragment: exact for split pieces).
// Weight slots: W0 | W0+W1+W2 | W2.
template <int MW, int NW, bool RELOAD>
__device__ __forceinline__ void hx_unit_poly(f32x4 (&acc)[3][NW][MW], const unsigned char* tile, const int (&aoff)[3],
                                             int lo_off, HxWeightStream<3 * MW * HX_NP>& ws) {
    constexpr int NSTEP = 3 * NW;
    HxFrag side[2], mid[2];                                     // x[j -/+ 1] and x[j] of the step, one step ahead
    side[0] = hx_read(tile, aoff[0], lo_off);
    mid[0] = hx_read(tile, aoff[1], lo_off);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int slot = s / NW, n = s % NW;
        if (s + 1 < NSTEP) {
            const int slot1 = (s + 1) / NW, n1 = (s + 1) % NW;
            if (slot1 != 1) side[(s + 1) & 1] = hx_read(tile, aoff[slot1] + n1 * 16 * HX_ROW, lo_off);
            mid[(s + 1) & 1] = hx_read(tile, aoff[1] + n1 * 16 * HX_ROW, lo_off);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (slot == 1) {
            hx_step<MW>(acc[1][n], mid[s & 1], &ws.wr[slot * MW * HX_NP]);
        } else {
            hx_step<MW>(acc[slot][n], side[s & 1], &ws.wr[slot * MW * HX_NP]);
            hx_step<MW>(acc[slot][n], hx_neg(mid[s & 1]), &ws.wr[slot * MW * HX_NP]);
        }
        if (RELOAD && n + 1 == NW) {
            #pragma unroll
            for (int q = 0; q < MW * HX_NP; ++q) ws.request(slot * MW * HX_NP + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (RELOAD) ws.advance();
}

// DEC2 unit (first k=3 conv + 1x1 residual conv of a down stage on the decimated input, MODE_DEC2):
//   acc[0] += sum_tap lrelu(x)[t + tap - 1] W[tap]   (LeakyReLU'd tile),   acc[1] += x[t] W1x1   (raw tile)
// Weight slots: w0 w1 w2 | w1x1.  raw_off: byte offset of the raw tile behind the LeakyReLU'd one.
template <int MW, int NW, bool RELOAD>
__device__ __forceinline__ void hx_unit_dec2(f32x4 (&acc)[2][NW][MW], const unsigned char* tile, const int (&aoff)[3],
                                             int lo_off, int raw_off, HxWeightStream<4 * MW * HX_NP>& ws) {
    constexpr int NSTEP = 4 * NW;
    HxFrag a[2];
    a[0] = hx_read(tile, aoff[0], lo_off);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int slot = s / NW, n = s % NW;
        if (s + 1 < NSTEP) {
            const int slot1 = (s + 1) / NW, n1 = (s + 1) % NW;
            a[(s + 1) & 1] = slot1 < 3 ? hx_read(tile, aoff[slot1] + n1 * 16 * HX_ROW, lo_off)
                                       : hx_read(tile + raw_off, aoff[1] + n1 * 16 * HX_ROW, lo_off);
            __builtin_amdgcn_sched_barrier(0);
        }
        hx_step<MW>(acc[slot == 3][n], a[s & 1], &ws.wr[slot * MW * HX_NP]);
        if (RELOAD && n + 1 == NW) {
            #pragma unroll
            for (int q = 0; q < MW * HX_NP; ++q) ws.request(slot * MW * HX_NP + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (RELOAD) ws.advance();
}

// MODE_CHAIN: the first conv's result tiles -> the intermediate LDS tile the second conv reads, in the
// producers' format (time-major rows of 32 channels, hi + lo pieces, same swizzle):
//   T2[row][co] = split(lrelu(acc + bias_mid)),  0 where the column lies outside the utterance (the second
//   conv's "same" zero padding).  The tiles are TRANSPOSED results (hx_prod SWAP): lane (t = lane & 15,
//   g = lane >> 4) owns channels 4g .. 4g+3 of time step t = 8 bytes of a row.  row0 is a multiple of 16, so the
//   swizzle of hx_lds_off depends on the lane only: one address per m, immediate offsets for n.
typedef hx_t hx4 __attribute__((ext_vector_type(4)));
// smid (LDS): [bias of the first conv | its inverse operand scales], per channel, both already multiplied by the
// intermediate tile's own scale (LeakyReLU commutes with a positive factor): v = acc * kinv + kb.  Read here, once
// per tile, instead of living in 8 MW registers across the tile loop.
// RAW_TOO (MODE_UPHEAD): the tile is also stored WITHOUT the LeakyReLU, `raw_bytes` further.
typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
template <int MW, int NA, bool RAW_TOO = false>
__device__ __forceinline__ void hx_chain_store(const f32x4 (&acc)[NA][MW], int ntl, unsigned char* T2, int chunk_bytes,
                                               int lo_off, int mg, int row0, int col0, int T, const float* smid,
                                               int cmidp, int lane, int raw_bytes = 0) {
    const int l15 = lane & 15, g = lane >> 4;
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int co0 = (mg * MW + m) * 16 + 4 * g;
        unsigned char* base = T2 + (co0 >> 5) * chunk_bytes + hx_lds_off(row0 + l15, (co0 & 31) >> 3) + (co0 & 7) * 2;
        const f32x4 kb = *reinterpret_cast<const f32x4*>(smid + co0);
        f32x4 kinv = kb;
        if constexpr (HX_NP == 2) kinv = *reinterpret_cast<const f32x4*>(smid + cmidp + co0);
        #pragma unroll
        for (int n = 0; n < NA; ++n) {
            if (n >= ntl) continue;
            const bool inside = (unsigned)(col0 + n * 16 + l15) < (unsigned)T;
            f32x4 v;
            if constexpr (HX_NP == 2) v = acc[n][m] * kinv + kb; else v = acc[n][m] + kb;
            #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = inside ? v[e] : 0.f;      // (|v| < 2^15 by construction: s_mid comes from a bound of this tensor)
            if constexpr (RAW_TOO) {
                const hx4 hr = __builtin_convertvector(v, hx4);
                *reinterpret_cast<hx4*>(base + raw_bytes + n * 16 * HX_ROW) = hr;
                if constexpr (HX_NP == 2) {
                    const u32x2p hp = __builtin_bit_cast(u32x2p, hr);
                    *reinterpret_cast<u32x2p*>(base + raw_bytes + lo_off + n * 16 * HX_ROW) =
                        u32x2p{hx_lo_pair(hp.x, v.x, v.y), hx_lo_pair(hp.y, v.z, v.w)};
                }
            }
            #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * LRELU_SLOPE);
            const hx4 h = __builtin_convertvector(v, hx4);
            *reinterpret_cast<hx4*>(base + n * 16 * HX_ROW) = h;
            if constexpr (HX_NP == 2) {
                const u32x2p hp = __builtin_bit_cast(u32x2p, h);
                *reinterpret_cast<u32x2p*>(base + lo_off + n * 16 * HX_ROW) =
                    u32x2p{hx_lo_pair(hp.x, v.x, v.y), hx_lo_pair(hp.y, v.z, v.w)};
            }
        }
    }
}

#ifdef FASTSVC_ACT_BF16
// Polyphase epilogue in bfloat16 storage: a lane's 4 S consecutive output samples go out as 16-byte pieces
// (8 samples; one 8-byte piece left over for odd S) instead of S 8-byte ones (see ws_epilogue_poly).
template <int MW, int NW, int EPI, int S, bool TAILK = false, class KT>
__device__ __forceinline__ void hx_epilogue_poly8(const ConvParams& p, const EpiRsrc& R, f32x4 (&acc)[3][NW][MW],
                                                  float (&s1)[MW], float (&s2)[MW],
                                                  int mg, int tcol0, bool active, int lane, const KT& K) {
    if (!active) return;
    const float slope = (p.flags & F_POST_LRELU) ? LRELU_SLOPE : 1.0f;
    const int T_out = p.ldy;
    const int shift_soff = p.COUT * T_out * 2;             // bytes
    constexpr int NP8 = S / 2;                              // 16-byte pieces; S odd: plus one 8-byte piece
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int cot = (mg * MW + m) * 16 + (lane & 15);
        const bool cok = cot < p.COUT;
        const int co = cok ? cot : 0;
        const float bias = K.bias(p, 0, m, cot);
        const int rowoff = co * T_out;
        #pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int t = tcol0 + n * 16 + (lane >> 4) * 4;            // input-rate column (rows are a multiple of 4 long)
            const bool ok = cok && t < p.T;
            const int nvo = TAILK ? (ok ? min(4, p.T - t) * S : 0) : 4 * S;   // output samples of this lane inside the row
            const int boff0 = ok ? (rowoff + t * S) * 2 : OOB_OFF;     // bytes
            const int foff0 = ok ? (rowoff + t * S) * 4 : OOB_OFF;     // the 4-wide helpers take float bytes
            const f32x4 zz = acc[1][n][m] + bias;
            const f32x4 zf = zz + acc[0][n][m];
            const f32x4 zl = zz + acc[2][n][m];
            auto phase_value = [&](int k) -> float {
                const int jj = k / S, ph = k % S;
                return ph == 0 ? zf[jj] : ph == S - 1 ? zl[jj] : zz[jj];
            };
            f32x8 l1[NP8 > 0 ? NP8 : 1], l2[NP8 > 0 ? NP8 : 1];
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_AFF) {                                      // every operand load of the item first
                #pragma unroll
                for (int q = 0; q < NP8; ++q) {
                    l1[q] = act_load8(R.ss, ok ? boff0 + q * 16 : OOB_OFF, 0);
                    l2[q] = act_load8(R.ss, ok ? boff0 + q * 16 : OOB_OFF, shift_soff);
                }
                if (S & 1) {
                    t1 = act_load4(R.ss, ok ? foff0 + (S - 1) * 16 : OOB_OFF, 0);
                    t2 = act_load4(R.ss, ok ? foff0 + (S - 1) * 16 : OOB_OFF, shift_soff * 2);
                }
            }
            #pragma unroll
            for (int q = 0; q < NP8; ++q) {
                f32x8 v;
                #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = phase_value(8 * q + e), c = phase_value(8 * q + 4 + e);
                    v.lo[e] = fmaxf(a, a * slope); v.hi[e] = fmaxf(c, c * slope);
                }
                act_store8(R.y, ok ? boff0 + q * 16 : OOB_OFF, v, ok ? 8 : 0);      // dropped when y is absent
                if (EPI == EPI_AFF) {
                    f32x8 u;
                    u.lo = l1[q].lo * v.lo + l2[q].lo; u.hi = l1[q].hi * v.hi + l2[q].hi;
                    u = TAILK ? keep8_exact(u, nvo - 8 * q) : keep8(u, ok ? 8 : 0);
                    act_store8(R.y2, ok ? boff0 + q * 16 : OOB_OFF, u, ok ? 8 : 0);
                    s1[m] += ((u.lo.x + u.lo.y) + (u.lo.z + u.lo.w)) + ((u.hi.x + u.hi.y) + (u.hi.z + u.hi.w));
                    s2[m] += ((u.lo.x * u.lo.x + u.lo.y * u.lo.y) + (u.lo.z * u.lo.z + u.lo.w * u.lo.w)) +
                             ((u.hi.x * u.hi.x + u.hi.y * u.hi.y) + (u.hi.z * u.hi.z + u.hi.w * u.hi.w));
                }
            }
            if (S & 1) {
                f32x4 v;
                #pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = phase_value(4 * (S - 1) + e); v[e] = fmaxf(a, a * slope); }
                act_store4(R.y, ok ? foff0 + (S - 1) * 16 : OOB_OFF, v);
                if (EPI == EPI_AFF) {
                    f32x4 u = t1 * v + t2;
                    if (!ok) u = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (TAILK) u = keep_first(u, nvo - 4 * (S - 1));
                    act_store4(R.y2, ok ? foff0 + (S - 1) * 16 : OOB_OFF, u);
                    s1[m] += (u.x + u.y) + (u.z + u.w);
                    s2[m] += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
                }
            }
        }
    }
}
// Polyphase + FiLM-affine epilogue through the wave's own LDS patch (bfloat16 storage, round 4).  In hx_epilogue_poly8 a lane
// owns 4 S consecutive output samples of ONE channel row: its 16-byte pieces sit 8 S bytes apart (40 for S = 5: every
// other one misaligned, plus an 8-byte tail), 16 rows per instruction - the stretched FiLM layers ran at 2.2-3.1 TB/s where the
// direct ones reach 4-4.6.  Here the finished (LeakyReLU'd, bfloat16) tile goes to LDS in that layout and comes back
// row-major: four lanes = 64 contiguous bytes of a row, sixteen rows per instruction, for the scale / shift loads, the
// affine, its InstanceNorm sums and both stores.  LDS executes a wave's accesses in order: no barrier, a wave fence.
template <int MW, int NW, int S> constexpr int hx_poly_patch_bytes() { return MW * 16 * (NW * 16 * S * 2 + 16); }
// The staged epilogue runs in two passes and the second one needs nothing but the patch: the consumer wave converts its
// finished tile into the patch (pass 1) and, behind a workgroup barrier, takes the first channel tile of the row-major
// pass while its STAGING partner (wave + 4, idle by then: timeline of up.3.up_stretch - 10.7k of 13.5k cycles per tile in
// this epilogue, the staging waves 76 % of theirs at the barrier) takes the others.
// channel tiles the consumer keeps in pass 2
template <int MW> constexpr int hx_poly_mc() { return 1; }

template <int MW, int NW, int S, class KT>
__device__ __forceinline__ void hx_poly_pass1(const ConvParams& p, const f32x4 (&acc)[3][NW][MW], int mg, int lane, const KT& K,
                                              unsigned char* Pw) {
    constexpr int NC = NW * 16 * S;                         // output samples per row of this wave's tile
    constexpr int PB = NC * 2 + 16;                         // patch row pitch, bytes
    const float slope = (p.flags & F_POST_LRELU) ? LRELU_SLOPE : 1.0f;
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const float bias = K.bias(p, 0, m, (mg * MW + m) * 16 + (lane & 15));
        #pragma unroll
        for (int n = 0; n < NW; ++n) {
            const f32x4 zz = acc[1][n][m] + bias;
            const f32x4 zf = zz + acc[0][n][m];
            const f32x4 zl = zz + acc[2][n][m];
            auto phase_value = [&](int k) -> float {
                const int jj = k / S, ph = k % S;
                const float a = ph == 0 ? zf[jj] : ph == S - 1 ? zl[jj] : zz[jj];
                return fmaxf(a, a * slope);
            };
            unsigned char* dst = Pw + (m * 16 + (lane & 15)) * PB + ((n * 16 + (lane >> 4) * 4) * S) * 2;
            #pragma unroll
            for (int q = 0; q < S; ++q) {                   // 4 S samples = S pieces of 8 bytes
                u32x2v w;
                w.x = bf16_pack2(phase_value(4 * q), phase_value(4 * q + 1));
                w.y = bf16_pack2(phase_value(4 * q + 2), phase_value(4 * q + 3));
                *reinterpret_cast<u32x2v*>(dst + q * 8) = w;
            }
        }
    }
}

// pass 2 over the channel tiles [M0, M1): a lane owns 16 bytes (8 samples) of a row per 64-byte chunk.  request(): the
// scale / shift operands, issued ahead (the consumer's in front of its pass 1, the staging partner's in front of the
// barrier) where they fit the registers, else row tile by row tile inside run().
template <int MW, int NW, int EPI, int S, int M0, int M1>
struct HxPolyPass2 {
    static constexpr int NC = NW * 16 * S, PB = NC * 2 + 16, NCH = (NC + 31) / 32, NM = M1 - M0;
    static constexpr bool HOIST = NM * NCH <= 10;
    u32x4 l1[HOIST ? NM : 1][NCH], l2[HOIST ? NM : 1][NCH];     // (raw words: converted where they are used, see act4_t)
    int off[HOIST ? NM : 1][NCH], nv[HOIST ? NM : 1][NCH];
    __device__ __forceinline__ void fetch(const ConvParams& p, const EpiRsrc& R, int mg, int tcol0, int lane, int m, int slot) {
        const int rsub = lane >> 2, piece = lane & 3;
        const int T_out = p.ldy;
        const int shift_soff = p.COUT * T_out * 2;         // bytes
        const int ncv = max(0, min(NW * 16, p.T - tcol0)) * S;  // valid output samples of a row of this tile
        const int cot = (mg * MW + m) * 16 + rsub;
        const bool cok = cot < p.COUT;
        const int rowb = ((cok ? cot : 0) * T_out + tcol0 * S) * 2;
        #pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 32 + piece * 8;
            nv[slot][c] = (cok && col < NC) ? max(0, min(8, ncv - col)) : 0;
            off[slot][c] = nv[slot][c] > 0 ? rowb + col * 2 : OOB_OFF;
            if (EPI == EPI_AFF) {
                l1[slot][c] = __builtin_amdgcn_raw_buffer_load_b128(R.ss, off[slot][c], 0, 0);
                l2[slot][c] = __builtin_amdgcn_raw_buffer_load_b128(R.ss, off[slot][c], shift_soff, 0);
            }
        }
    }
    __device__ __forceinline__ void request(const ConvParams& p, const EpiRsrc& R, int mg, int tcol0, int lane) {
        if constexpr (HOIST) {
            #pragma unroll
            for (int m = M0; m < M1; ++m) fetch(p, R, mg, tcol0, lane, m, m - M0);
        }
    }
    __device__ __forceinline__ void run(const ConvParams& p, const EpiRsrc& R, const unsigned char* Pw, float (&s1)[MW], float (&s2)[MW],
                                        int mg, int tcol0, int lane) {
        const int rsub = lane >> 2, piece = lane & 3;
        #pragma unroll
        for (int m = M0; m < M1; ++m) {
            const unsigned char* src = Pw + (m * 16 + rsub) * PB + piece * 16;
            float a1 = 0.f, a2 = 0.f;
            const int sl = HOIST ? m - M0 : 0;
            if constexpr (!HOIST) fetch(p, R, mg, tcol0, lane, m, 0);
            // Every chunk's arithmetic first, every store after: act_store8's half-lane branch hides the stores from hipcc's
            // vmcnt count, so a chunk's operand words behind the previous chunk's stores were waited for with counts that
            // ended in vmcnt(0) - the row's last chunk waited for the acknowledgement of all its stores, once per tile and role
            u32x4 yw[EPI == EPI_AFF ? NCH : 1];
            #pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = c * 32 + piece * 8;
                if (EPI == EPI_AFF) {
                    const f32x8 v = bf8_unpack(*reinterpret_cast<const u32x4*>(src + (col < NC ? c * 64 : 0)));
                    f32x8 u;
                    const f32x8 sc = bf8_unpack(l1[sl][c]), sh = bf8_unpack(l2[sl][c]);
                    u.lo = sc.lo * v.lo + sh.lo; u.hi = sc.hi * v.hi + sh.hi;
                    u = keep8_exact(u, nv[sl][c]);
                    yw[c] = bf8_pack(u);
                    a1 += ((u.lo.x + u.lo.y) + (u.lo.z + u.lo.w)) + ((u.hi.x + u.hi.y) + (u.hi.z + u.hi.w));
                    a2 += ((u.lo.x * u.lo.x + u.lo.y * u.lo.y) + (u.lo.z * u.lo.z + u.lo.w * u.lo.w)) +
                          ((u.hi.x * u.hi.x + u.hi.y * u.hi.y) + (u.hi.z * u.hi.z + u.hi.w * u.hi.w));
                }
            }
            // (hipcc sinks a chunk's arithmetic back behind the previous chunk's stores unless the words are pinned here)
            if (EPI == EPI_AFF) {
                #pragma unroll
                for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(yw[c]));
            }
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = c * 32 + piece * 8;
                const int nvs = min(8, (nv[sl][c] + 3) & ~3);    // (ragged rows: the straddling group of 4 is stored whole)
                // (y is the patch itself, already bfloat16; wave-uniform: the FiLM-affine layers write y2 only)
                if (p.y) act_store8_raw(R.y, off[sl][c], *reinterpret_cast<const u32x4*>(src + (col < NC ? c * 64 : 0)), nvs);
                if (EPI == EPI_AFF) act_store8_raw(R.y2, off[sl][c], yw[c], nvs);
            }
            if (EPI == EPI_AFF) {
                // the caller folds s1 / s2 over the lane >> 4 groups and takes channel lane & 15 from lanes 0..15: hand the row
                // sums over in that layout (row r lives in lanes 4 r .. 4 r + 3 here)
                a1 += __shfl_xor(a1, 1); a2 += __shfl_xor(a2, 1);
                a1 += __shfl_xor(a1, 2); a2 += __shfl_xor(a2, 2);
                const float r1 = __shfl(a1, 4 * (lane & 15)), r2 = __shfl(a2, 4 * (lane & 15));
                if (lane < 16) { s1[m] += r1; s2[m] += r2; }
            }
        }
    }
};
// The decimating pair's two outputs (c1 and the 1x1 residual r) the same way: ws_epilogue_dec2 stores 8-byte pieces, sixteen
// rows x four pieces per instruction and tensor - the pairs ran at 8-10 % matrix-pipe occupancy, 5 k of 6 k cycles per unit outside
// the matrix work.  Both finished tiles go to the wave's LDS patch and leave row-major: NW * 32 contiguous bytes per row.
template <int MW, int NW> constexpr int hx_dec2_patch_bytes() { return 2 * MW * 16 * (NW * 32 + 16); }
template <int MW, int NW, class KT>
__device__ __forceinline__ void hx_epilogue_dec2_staged(const ConvParams& p, const EpiRsrc& R, f32x4 (&acc)[2][NW][MW], int sig, int mg,
                                                        int tcol0, bool active, int lane, const KT& K, unsigned char* Pw) {
    if (!active) return;
    constexpr int PB = NW * 32 + 16;                        // patch row pitch, bytes
    constexpr int TB = MW * 16 * PB;                        // one tensor's tile
    constexpr int NP = NW * 2;                              // 16-byte pieces per row
    constexpr int ITEMS = MW * 16 * NP;
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int cot = (mg * MW + m) * 16 + (lane & 15);
        const float bias = K.bias(p, sig, m, cot), bias2 = K.bias2(p, sig, m, cot);
        const float iv = K.inv(m), iv2 = K.inv2(m);
        #pragma unroll
        for (int n = 0; n < NW; ++n) {
            unsigned char* dst = Pw + (m * 16 + (lane & 15)) * PB + (n * 16 + (lane >> 4) * 4) * 2;
            const f32x4 a = KT::finish(acc[0][n][m], iv, bias), b = KT::finish(acc[1][n][m], iv2, bias2);
            u32x2v wa, wb;
            wa.x = bf16_pack2(a.x, a.y); wa.y = bf16_pack2(a.z, a.w);
            wb.x = bf16_pack2(b.x, b.y); wb.y = bf16_pack2(b.z, b.w);
            *reinterpret_cast<u32x2v*>(dst) = wa;
            *reinterpret_cast<u32x2v*>(dst + TB) = wb;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    #pragma unroll
    for (int k = 0; k < (ITEMS + 63) / 64; ++k) {
        const int idx = lane + 64 * k;
        const int row = idx / NP, piece = idx - row * NP;
        const int cot = mg * MW * 16 + row;
        const int col = tcol0 + piece * 8;
        const int nv = (idx < ITEMS && cot < p.COUT) ? max(0, min(8, p.T - col)) : 0;
        const int nvs = min(8, (nv + 3) & ~3);              // (ragged rows: the straddling group of 4 is stored whole)
        const int off = nv > 0 ? (cot * p.ldy + col) * 2 : OOB_OFF;
        const unsigned char* src = Pw + (idx < ITEMS ? row * PB + piece * 16 : 0);
        const u32x4 wa = *reinterpret_cast<const u32x4*>(src), wb = *reinterpret_cast<const u32x4*>(src + TB);
        if (__builtin_amdgcn_ballot_w64(nvs == 4) == 0) {   // wave-uniform: no half piece (nearly always)
            __builtin_amdgcn_raw_buffer_store_b128(wa, R.y, off, 0, FASTSVC_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(wb, R.y2, off, 0, FASTSVC_ST_AUX);
        } else {
            u32x2v a0, a1, b0, b1;
            a0.x = wa.x; a0.y = wa.y; a1.x = wa.z; a1.y = wa.w; b0.x = wb.x; b0.y = wb.y; b1.x = wb.z; b1.y = wb.w;
            __builtin_amdgcn_raw_buffer_store_b64(a0, R.y, nvs >= 4 ? off : OOB_OFF, 0, FASTSVC_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(a1, R.y, nvs >= 8 ? off + 8 : OOB_OFF, 0, FASTSVC_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(b0, R.y2, nvs >= 4 ? off : OOB_OFF, 0, FASTSVC_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(b1, R.y2, nvs >= 8 ? off + 8 : OOB_OFF, 0, FASTSVC_ST_AUX);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int MW, int NW, int MODE> constexpr bool hx_dec2_staged() { return MODE == MODE_DEC2 && hx_dec2_patch_bytes<MW, NW>() <= 16 * 1024; }
// instances that take the staged polyphase epilogue (the patch of a wave: MW * 16 rows of 4 S NW * 8 + 16 bytes)
template <int MW, int NW, int MODE, int EPI, int S>
constexpr bool hx_poly_staged() { return MODE == MODE_POLY && (EPI == EPI_AFF || EPI == EPI_PLAIN) && hx_poly_patch_bytes<MW, NW, S>() <= 16 * 1024; }
// the pair epilogue needs an even number of time tiles per wave; its operands are staged while they fit
template <int MW, int NW, int MODE> constexpr bool hx_pairs_epi() { return MODE == MODE_DIRECT && NW % 2 == 0; }
#else
template <int MW, int NW, int MODE> constexpr bool hx_pairs_epi() { return false; }
template <int MW, int NW, int MODE, int EPI, int S> constexpr bool hx_poly_staged() { return false; }
template <int MW, int NW, int S> constexpr int hx_poly_patch_bytes() { return 0; }
template <int MW, int NW, int MODE> constexpr bool hx_dec2_staged() { return false; }
template <int MW, int NW> constexpr int hx_dec2_patch_bytes() { return 0; }
#endif

// variants that stage their epilogue operands (scale, shift, residual) in LDS ahead of the epilogue with
// LDS-DMA pieces (ws_epilogue_stage, fastsvc_device.inc): fetched inside the epilogue each 16x16 item costs a
// full memory round trip (timeline of up.3.d3 without it: 15.5k of 17k cycles per unit in the epilogue)
template <int MW, int NW, int MODE, int EPI>
constexpr bool hx_estage() {
    return MODE == MODE_DIRECT && (EPI == EPI_AFF || EPI == EPI_RES) && (hx_pairs_epi<MW, NW, MODE>() ? MW * NW <= 12 : MW * NW <= 6);
}

// instances that request the NEXT tile's epilogue operands one tile ahead into a second slot set (EST2, see the
// consumer loop): one K chunk per tile (the caller checks WSTATIC) and slot sets small enough to double
template <int MW, int NW, int MODE, int EPI>
constexpr bool hx_est2() { return hx_estage<MW, NW, MODE, EPI>() && MW * NW <= 4; }

template <int MW, int NW, int MODE, int EPI, int S = 1>
constexpr int hx_min_waves() {
#ifdef FASTSVC_ACT_BF16
    // ... and the C = 24 middle conv with its second operand (bfloat16 storage): it sits at the edge of the budget, and
    // past it the CU holds one workgroup instead of two (measured 948 -> 1147 us at 64 x 240000)
    if (MODE == MODE_DIRECT && MW == 2 && NW == 2 && EPI == EPI_AFF && S > 1) return 4;     // (it fits without a spill: checked)
#endif

    // 128 VGPRs = two 8-wave workgroups per CU (one computes while the other stores: the C = 24 layers are
    // bound by their store phase) where the producers' two register sets (2 x 32), the unit-deep weight ring
    // (12 * MW) and the accumulators fit; else 256
    return (MODE == MODE_DIRECT && MW == 2 && (EPI == EPI_PLAIN || EPI == EPI_RANK1 || EPI == EPI_RES)) ? 4 : 2;
}

// conv_last behind the last block's conv (ConvParams::last_w): acc holds the block's finished output tile of ALL
// C channels (one channel group): y[t] = sum_co last_w[co] * out[co][t] + last_b.  Lanes 0..15 of a row quad hold
// the 16 channels of a tile: a 4-step butterfly over them, lane 0 of the quad stores 4 samples.  PAIRS: acc is in
// the 8-wide pair layout of hx_epilogue8 (tile 2k = samples 0..3, tile 2k+1 = samples 4..7 of the lane's 8).
template <int MW, int NW, bool PAIRS>
__device__ __forceinline__ void hx_last_reduce(const ConvParams& p, const f32x4 (&acc)[NW][MW], const float (&kw)[MW],
                                               float bias, int b, int tcol0, int lane) {
    // (bias = last_b[0], read ONCE by the caller: read here it was a global load per tile and, behind it, a vmcnt(0) in the
    // consumer's chain - an L2 round trip and every younger request of the wave waited for, tile after tile)
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.last_y + (long)b * p.last_y_b, p.ldy);
    #pragma unroll
    for (int n = 0; n < NW; ++n) {
        f32x4 s = acc[n][0] * kw[0];
        #pragma unroll
        for (int m = 1; m < MW; ++m) s += acc[n][m] * kw[m];
        // sum over the 16 lanes of a row (= the 16 channels of a tile) by DPP moves - quad swaps, then the mirrored half row
        // and the mirrored row: every pairing joins disjoint partial sums, all 16 lanes end with the total.  (__shfl_xor is
        // ds_bpermute here: four DEPENDENT LDS-crossbar round trips per component, the fused conv_last cost more than the
        // 737 MB store it saves)
        #pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = s[e];
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
            v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
            s[e] = v;
        }
        const int t = PAIRS ? tcol0 + (n >> 1) * 32 + (lane >> 4) * 8 + (n & 1) * 4 : tcol0 + n * 16 + (lane >> 4) * 4;
        buf_store4(yr, ((lane & 15) == 0 && t < p.T) ? t * 4 : OOB_OFF, s + bias);
    }
}

// MODE: MODE_DIRECT (any dilation <= 28), MODE_POLY (S = stretch factor; tiles walk the INPUT columns) or
// MODE_DEC2 (p.s = decimation; two outputs).  EPI: epilogue kind (DEC2: ignored).
// TAILK: instance for ragged batches whose rows run at the frame rate or twice it - an utterance's own row length
// need not be a multiple of 4 then (the pitch is).  The staging waves zero what lies past the row end AFTER the
// prologue (the conv's zero padding), the epilogue keeps it out of the InstanceNorm sums and the running max and
// stores the straddling float4 whole (between the row end and the pitch lies nobody's data).  Its own instances:
// folded into the others it cost the 128-register variants their spill-free allocation.
// bfloat16 storage, plain windows (a direct conv's or the stretched conv's input): the staging waves request 16 bytes per
// lane - 8 time steps of one channel - instead of 8.  A CU's vector-memory pipe takes a wave's request in about the same
// time whatever its width (tools/micro/cu_pull.hip, profiles/r6_cu_pull_microbench.txt: 8 B per lane requests stop at
// ~13 B/clk per CU even out of L2, 16 B ones reach 19-47), and the windows were the only 8 B requests left in the
// bfloat16 path.  Item = (quad of 4 channels, octet of 8 rows): 4 requests of 8 rows x 128 B per item, the same 16
// registers per set; the window's halo is aligned to 8 rows so that an item lies inside the utterance or outside it
// (its first 4 rows inside: row ends are multiples of 4).
#ifndef FASTSVC_HX_W8
#define FASTSVC_HX_W8 1
#endif
constexpr bool hx_w8(int MODE, bool tailk, int S) {      // (the decimating pair: its compact-input instance, S = 2)
    return FASTSVC_HX_W8 != 0 && HX_NP == 1 && !tailk && (MODE == MODE_DIRECT || MODE == MODE_POLY || (MODE == MODE_DEC2 && S == 2));
}

template <int MW, int NW, int WM, int WN, int MODE, int EPI, int S, bool WSTATIC, bool TAILK = false>
__global__ __launch_bounds__(512, (hx_min_waves<MW, NW, MODE, EPI, S>()))
void conv_hx_kernel(const ConvParams p0) {
    constexpr bool UPH = MODE == MODE_UPHEAD;                          // conv_first -> {stretched residual conv, stretched up conv + FiLM affine}
    constexpr bool POLY = MODE == MODE_POLY, DEC2 = MODE == MODE_DEC2, CHAIN = MODE == MODE_CHAIN || MODE == MODE_CHAIN1 || UPH;
    constexpr bool IN1 = MODE == MODE_CHAIN1;                          // the staging waves compute the stage's first conv
    constexpr bool WLB = MODE == MODE_CHAIN && S == 2;                 // the second conv's weights live in LDS (one channel group)
    // MODE_DIRECT with S > 1: a SECOND operand (ConvParams::x2) - the raw tensor x2 stretched S times along time through
    // its own k=3, d=1 conv - is accumulated with the main conv: units alternate (main chunk, x2 chunk), the main ones
    // in window buffer 0, the x2 ones in buffer 1, one weight unit each.  (Measured against ONE unit per chunk that stages
    // both windows and runs six tap steps - one barrier less per tile: C = 24, bfloat16, 64 x 240000: 1233-1343 us
    // against 954 - the second window's LDS and the twelve resident fragments cost the second workgroup of the CU;
    // C = 48: 555 against 585 us.)
    constexpr bool XR = MODE == MODE_DIRECT && S > 1;
    constexpr int NT = 16 * NW * WN;                                   // (input-rate) columns per workgroup tile
    // MODE_CHAIN: the first conv also produces the second one's halo (<= 4 columns per side): its tile starts 4
    // columns early and is one 16-column MFMA tile longer (computed by the last wave along time)
    constexpr int NTV = CHAIN ? NT + 16 : NT;                          // columns the staged window serves
    constexpr int HB = CHAIN ? 4 : 0;                                  // aligned halo of the second conv
    constexpr int NPROD_T = 256;                                       // producer threads
    constexpr bool W8 = hx_w8(MODE, TAILK, S);                            // 16-byte window requests (see hx_w8)
    constexpr int SPARE = W8 ? 8 : 4;                                  // rows behind the tile where item-less threads park
    constexpr int MAXW = NTV + (MODE == MODE_DIRECT ? (W8 ? 64 : 56) : (W8 ? 16 : 8));   // halo <= 28 rows per side (POLY / DEC2: 1, CHAIN: <= 4)
    constexpr int ITEMS = (MAXW + NPROD_T - 1) / NPROD_T;              // (octet, 4 time steps) items per producer thread
    constexpr int NWS = DEC2 ? 4 : 3;                                  // weight slots per unit and channel tile
    constexpr int NVAR = DEC2 ? 2 : 1;                                 // tile variants: LeakyReLU'd (+ raw)
    constexpr int NSLOT = NWS * MW * HX_NP;
    // this instance's epilogue measures what it writes (ConvParams::amax_out): residual kinds and the fused pair
    constexpr bool TRACKS = AMAX_TRACK && ((CHAIN && MODE != MODE_UPHEAD) || (MODE == MODE_DIRECT && (EPI == EPI_RES || EPI == EPI_RANK1)));
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
    ConvParams p = p0;
    if (p0.lens) {                                                     // ragged batch: this utterance's own lengths
        const int frames = p0.lens[b];
        p.T = frames * p0.len_mul;
        p.x_T = frames * p0.xlen_mul;
    }
    const int mg = blockIdx.y * WM + wave_m;
    const bool active = !producer && mg < p.ngroups;
    const int halo = (MODE == MODE_DIRECT || CHAIN) ? p.dil : 1;
    const int halo_al = W8 ? ((halo + 7) & ~7) : ((halo + 3) & ~3);
    const int W = NTV + 2 * halo_al;                                   // tile rows
    const int nch = p.nch32;
    const int CINp = nch * HX_KC;
    const int flags = p.flags;
    const int ntx = (p.T + NT - 1) / NT;
    const int tile0 = blockIdx.x * p.tpw;
    const int ntiles = min(p.tpw, ntx - tile0);
    if (ntiles <= 0) return;
    const int nunits = ntiles * (XR ? 2 * nch : nch);
#ifdef FASTSVC_TIMELINE
    // diagnostic build (tools/timeline.py): lane 0 of every wave stamps s_memtime at its phase boundaries
    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long* tlw = (p.tl && wg_lin < p.tl_wgs) ? p.tl + ((long)wg_lin * 8 + wave) * 64 : nullptr;
    int tli = 0;
    auto stamp = [&](int tag) {
        if (tlw && lane == 0 && tli < 62) { tlw[tli++] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); }
    };
    stamp(1);
    if (tlw && lane == 0) {
        tlw[62] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        tlw[63] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    }
#else
    auto stamp = [](int) {};
#endif

    // Two workgroups share a CU in the 128-VGPR variants.  Dispatched together they would run their phases
    // in lockstep (both load, both multiply, both store): the second one of a CU (workgroup ids 256 apart
    // in dispatch order) starts about half a unit late, so one stores while the other multiplies.
    if constexpr (hx_min_waves<MW, NW, MODE, EPI, S>() >= 4) {
        const int wg_id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if ((wg_id >> 8) & 1) {
            for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles per step
        }
    }
    double* sstat = reinterpret_cast<double*>(smem_raw);                               // [WM*MW*16][2]
    float2* ncoef = reinterpret_cast<float2*>(smem_raw + sizeof(double) * 2 * 16 * MW * WM);   // [CINp + 8]: the last 8 are (0, 0)
    unsigned char* tiles = reinterpret_cast<unsigned char*>(ncoef + CINp + 8);         // [2][HX_NP][W rows][64 B]
    // small per-workgroup constants (static LDS; HX_STATIC_LDS bounds them for the launcher's size checks)
    __shared__ unsigned s_amax, s_cnt;                 // workgroup's largest |value| written (ConvParams::amax_out), waves done
    __shared__ float s_inv[HX_NP == 2 ? 2 * 16 * MW * WM : 4];        // inverse operand scales [conv | second DEC2 output][channel]
    __shared__ __attribute__((aligned(16))) float s_mid[CHAIN ? 2 * (16 * MW * WM + 32) : 4];   // MODE_CHAIN: hx_chain_store's constants
    // ---- operand scales of the split-binary16 products (ConvParams, "dynamic range"; powers of two) ----
    // Set by setup_shared, i.e. AFTER the staging waves have their first two windows and the consumers their weight
    // stream in flight: the amax row is one more dependent global load, and in front of everything else it cost
    // every launch its latency (+2-3.5 us per kernel, measured).
    float sx = 1.f;                                    // the staging waves multiply what they stage by sx
    float sx2 = 1.f;                                   // XR: ... and the second operand by sx2
    float s_mid_scale = 1.f;                           // MODE_CHAIN: scale of the intermediate tile
    const int ninv = 16 * MW * p.ngroups;
    const float* invtab = (HX_NP == 2 && p.whx_inv) ? p.whx_inv + (long)sig * p.whx_inv_sig : nullptr;
    auto operand_scales = [&]() {
#ifndef FASTSVC_EXP_NOSCALE
        if constexpr (HX_NP == 2) {
            if (p.amax_in) {
                auto bound_of = [&](int entry, int s) -> float {
                    float bd = amax_read(p.amax_in, entry);
                    #pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (p.bnd_path[j]) { const float* q = p.bnd_path[j] + (long)s * p.bnd_sig[j]; bd = bd * q[0] + q[1]; }
                    return bd;
                };
                float bound = bound_of(sig * p.amax_in_sig + b, sig);
                if (p.amax_in2 > 0) bound = fmaxf(bound, bound_of(p.amax_in2 + b, 1));
                // behind an InstanceNorm the row is that of the speaker biases p (spk_proj_kernel fills it): a
                // normalised row of n samples never exceeds sqrt(n - 1) in magnitude, plus its bias
                if (flags & F_PRE_NORM) bound += sqrtf((float)p.x_T);
                if constexpr (CHAIN) {
                    if (invtab) {
                        const float* cst = invtab + (UPH ? 3 : 2) * ninv;      // l1 / largest |bias| of the first conv (and of the 1 -> C conv)
                        if constexpr (IN1) bound = bound * cst[2] + cst[3];
                        s_mid_scale = hx_scale_for(bound * cst[0] + cst[1]);
                    }
                }
                sx = hx_scale_for(bound);
            }
            if constexpr (XR) {
                if (p.amax_x2) {
                    float bd = amax_read(p.amax_x2, b);
                    if (p.bnd_x2) bd = bd * p.bnd_x2[0] + p.bnd_x2[1];
                    sx2 = hx_scale_for(bd);
                }
            }
        }
#endif
    };
    const int lo_off = (W + SPARE) * HX_ROW;                           // hi tile (+ the spare rows), then lo tile
    const int raw_off = HX_NP * (W + SPARE) * HX_ROW;                  // DEC2: the raw tile behind the LeakyReLU'd one
    const int bufsz = NVAR * HX_NP * (W + SPARE) * HX_ROW;
    // MODE_CHAIN: the intermediate tile behind the two window buffers, [32-channel chunk][hi, lo][NT + 16 rows]
    constexpr int T2ROWS = NT + 16;
    constexpr int T2CHUNK = HX_NP * T2ROWS * HX_ROW;
    unsigned char* T2 = tiles + 2 * bufsz;
    const int t2one = CHAIN ? p.nch32b * T2CHUNK : 0;                   // one copy of the intermediate tile
    const int t2bytes = UPH ? 2 * t2one : t2one;                       // MODE_UPHEAD: LeakyReLU'd copy, then the raw one
    unsigned char* Wl = T2 + t2bytes;                                  // WLB: [K chunk][tap][m][piece] fragments of the second conv
    const int wlbytes = WLB ? p.nch32b * NSLOT * HX_FRAG : 0;

    auto setup_shared = [&]() {
        // Every global load of the set-up is ISSUED first - InstanceNorm sums and speaker bias of this thread's input
        // channel (one channel per thread: C_in <= 512), the fused pair's bias / inverse scales of its middle channel,
        // then the scalar loads of operand_scales() - and waited for once: written one after the other they were two
        // to three serialised memory round trips in front of the first tile (+1.5-3 us on every fused launch).
        const int c = tid;
        double q1 = 0.0, q2 = 0.0;
        float pc = 0.f;
        if ((flags & F_PRE_NORM) && c < p.CIN) {
            q1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
            q2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
            pc = p.spk[(long)b * p.CIN + c];
        }
        const int cmidp = CHAIN ? p.nch32b * HX_KC : 0;
        float bmid = 0.f, imid = 1.f;
        if constexpr (CHAIN) {
            if (c < p.CMID) {
                bmid = p.bias_mid[(long)sig * p.bias_mid_sig + c];
                if constexpr (HX_NP == 2) { if (invtab) imid = invtab[c]; }
            }
        }
        operand_scales();
        if constexpr (TRACKS) { if (tid == 0) { s_amax = 0u; s_cnt = 0u; } }
        if (flags & F_STATS) {
            for (int i = tid; i < 2 * 16 * MW * WM; i += 512) sstat[i] = 0.0;
        }
        if constexpr (CHAIN) {
            // channel padding of the intermediate tile is never written: it must read as 0, not as LDS garbage
            for (int i = tid * 16; i < t2bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(T2 + i) = u32x4{0u, 0u, 0u, 0u};
            // first conv's bias and inverse operand scales, times the intermediate tile's scale (hx_chain_store)
            if (c < cmidp) {
                const bool ok = c < p.CMID;
                s_mid[c] = ok ? bmid * s_mid_scale : 0.f;
                if constexpr (HX_NP == 2) s_mid[cmidp + c] = ok ? imid * (s_mid_scale / sx) : 0.f;
            }
        }
        if constexpr (WLB) {
            // the second conv's fragments of this (signal, channel group 0): once per workgroup
            const unsigned char* src = reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                                       (long)nch * NSLOT * HX_FRAG;
            for (int i = tid * 16; i < wlbytes; i += 512 * 16)
                *reinterpret_cast<u32x4*>(Wl + i) = *reinterpret_cast<const u32x4*>(src + i);
        }
        // prologue coefficients of every input channel, applied by the producers as ONE FMA u * A + Bc:
        // InstanceNorm + speaker bias (u - mean) * rstd + p  ->  A = rstd, Bc = p - mean * rstd (fastsvc.py:134-139);
        // no norm: (1, 0); channel padding: (0, 0) - all times sx, the power-of-two scale of the staged tile
        if (c < CINp) {
            float2 ab = make_float2(0.f, 0.f);
            if (c < p.CIN) {
                ab = make_float2(sx, 0.f);
                if (flags & F_PRE_NORM) {
                    const double inv_len = 1.0 / (double)p.x_T;
                    const double mean = q1 * inv_len;
                    double var = q2 * inv_len - mean * mean;      // biased variance (InstanceNorm2d)
                    var = var > 0.0 ? var : 0.0;
                    const double rstd = 1.0 / sqrt(var + IN_EPS);
                    ab.x = (float)rstd * sx;
                    ab.y = (float)((double)pc - mean * rstd) * sx;
                }
            }
            ncoef[c] = ab;
        }
        if (c < 8) ncoef[CINp + c] = make_float2(0.f, 0.f);       // what an item outside the utterance is staged with
        __syncthreads();
    };

    if (producer) {
        // ================================ PRODUCER WAVES ================================
        const int ptid = tid - 256;
        const __amdgpu_buffer_rsrc_t xr = IN1
            ? make_rsrc(p.x + (long)sig * p.x_sig + (long)b * p.x_b, p.T)          // raw signal row: float32 whatever the storage
            : act_rsrc(p.x, (long)sig * p.x_sig + (long)b * p.x_b,
                       p.xsplit > 0 ? p.x_sig + (long)p.xsplit * p.ldx : (long)p.CIN * p.ldx);
        const int xsplit_off = (int)(p.x_sig * 4);                                  // "float bytes" (the host bounds it below 2^31)
        // item = (octet of 8 channels, quad of 4 rows); threads without an item park theirs in the 4 spare rows
        // behind the tile, so that the code below is straight-line (any branch between the loads and their use
        // makes hipcc wait vmcnt(0), i.e. for the NEXT unit's loads as well)
        int it_oct[ITEMS], it_q[ITEMS];
        bool it_in[ITEMS];
        #pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int idx = i * NPROD_T + ptid;
            if constexpr (W8) {                          // (quad of 4 channels, octet of 8 rows)
                it_oct[i] = idx & 7;
                it_in[i] = (idx >> 3) < (W >> 3);
                it_q[i] = it_in[i] ? (idx >> 3) : (W >> 3);
            } else {
                it_oct[i] = idx & 3;
                it_in[i] = (idx >> 2) < (W >> 2);
                it_q[i] = it_in[i] ? (idx >> 2) : (W >> 2);
            }
        }
        const float slope = (DEC2 || (flags & F_PRE_LRELU)) ? LRELU_SLOPE : 1.0f;     // max(v, slope * v): identity for 1
        const bool tail_rows = TAILK && (p.T & 3) != 0;    // this utterance's rows end inside a float4 (ragged batch)
        // MODE_CHAIN1: taps and bias of the first conv for the 8 channels of this thread's item(s) (one octet per
        // thread when ITEMS == 1, which the launcher guarantees)
        float i1w[IN1 ? 8 : 1][3], i1b[IN1 ? 8 : 1];
        if constexpr (IN1) {
            #pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = it_oct[0] * 8 + c;
                const bool ok = ci < p.CIN;
                const float* wp = p.in1_w + (long)sig * p.in1_w_sig + (ok ? ci : 0) * 3;
                i1w[c][0] = ok ? wp[0] : 0.f; i1w[c][1] = ok ? wp[1] : 0.f; i1w[c][2] = ok ? wp[2] : 0.f;
                i1b[c] = ok ? p.in1_b[(long)sig * p.in1_b_sig + ci] : 0.f;
            }
        }
        // unconditional loads of unit `un`: 8 channels x 4 time steps per item; whatever lies outside the
        // tensor or the utterance reads as 0 through the descriptor (offset pushed out of range)
        // (tile, chunk) of the unit each of the lambdas below is called with next: they are called once per unit, in
        // unit order, and count along - `un / nch` and `un % nch` inside them were scalar divisions that hipcc moved
        // into skip-if-no-lane regions between the window loads and their use, and behind such a region it waits
        // vmcnt(0): for the loads just issued as well (the two-deep prefetch ran one deep in every second unit)
        int l_tl = 0, l_ch = 0, c_ch = 0, xl_tl = 0, xl_ch = 0, xc_tl = 0, xc_ch = 0, e_tl = 0, e_ch = 0;
        auto pos_next = [&](int& tl_, int& ch_) { const bool wrap = ch_ + 1 == nch; ch_ = wrap ? 0 : ch_ + 1; tl_ += wrap ? 1 : 0; };
        typedef typename HxWin<IN1>::type pwin_t;
        auto pload = [&](int un, pwin_t (&px)[ITEMS][8], unsigned& tokmask) {
            const int tl = l_tl, ch = l_ch;
            pos_next(l_tl, l_ch);
            const int t_start = (tile0 + tl) * NT - HB - halo_al;
            const int soff = ch * HX_KC * p.ldx * 4;
            const int rows_left = p.CIN - ch * HX_KC;
            tokmask = 0;
#ifdef FASTSVC_ACT_BF16
            if constexpr (W8) {
                #pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int t = t_start + 8 * it_q[i];                       // a multiple of 8: the item starts inside the row or not at all
                    const bool live = it_in[i] & ((unsigned)t < (unsigned)p.T) & (un < nunits) & !(FASTSVC_DBG_ON(p, DBG_NO_LOAD));
                    tokmask |= (live ? (unsigned)min(8, p.T - t) : 0u) << (4 * i);      // 8, or 4 at the row's end
                    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = it_oct[i] * 4 + c;
                        typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
                        const u32x4w w = __builtin_amdgcn_raw_buffer_load_b128(xr, (live & (r < rows_left)) ? (r * p.ldx + t) * 2 : OOB_OFF, soff >> 1, FASTSVC_LD_AUX);
                        px[i][2 * c] = act4_t{w.x, w.y};
                        px[i][2 * c + 1] = act4_t{w.z, w.w};
                    }
                }
                return;
            }
#endif
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int t = t_start + 4 * it_q[i];
                // (bitwise: the short-circuit form compiles to a lane-divergent region, and behind one hipcc waits vmcnt(0)
                // in front of the commit - for the loads just issued as well: the two-deep prefetch ran one deep)
                const bool tok = it_in[i] & ((unsigned)t < (unsigned)p.T) & (un < nunits) & !(FASTSVC_DBG_ON(p, DBG_NO_LOAD));
                // how many of the item's 4 time steps lie inside the row (a ragged batch's own row lengths need not
                // be a multiple of 4: the float4 that straddles the row end also holds whatever the pitch holds)
                if constexpr (TAILK) tokmask |= (tok ? (unsigned)min(4, p.T - t) : 0u) << (3 * i);
                else tokmask |= (tok ? 1u : 0u) << i;
                if constexpr (IN1) {
                    // six signal samples t-1 .. t+4 (a negative offset is out of range like any other: 0 = the
                    // first conv's own zero padding)
                    px[i][0] = buf_load4(xr, tok ? t * 4 : OOB_OFF, 0);
                    px[i][1].x = buf_load1(xr, tok ? (t - 1) * 4 : OOB_OFF, 0);
                    px[i][1].y = buf_load1(xr, tok ? (t + 4) * 4 : OOB_OFF, 0);
                } else {
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int r = it_oct[i] * 8 + c;
                    if constexpr (DEC2 && S == 2) {
                        // the input IS the compact decimated copy h[..., ::s] a whole-stage launch wrote (fastsvc_cond.hip):
                        // unit stride, one vector load per channel like any direct window
                        px[i][c] = act_load4_raw(xr, (tok & (r < rows_left)) ? (r * p.ldx + t) * 4 : OOB_OFF, soff);
                    } else if constexpr (DEC2) {
                        // x[..., ::s] (Squeeze2d): four strided elements; negative t lies before the tensor -> 0
                        const int o = (tok & (r < rows_left)) ? (r * p.ldx + t * p.s) * 4 : OOB_OFF;
                        px[i][c] = act_pack4_raw(act_load1_raw(xr, o, soff), act_load1_raw(xr, o + 4 * p.s, soff),
                                                 act_load1_raw(xr, o + 8 * p.s, soff), act_load1_raw(xr, o + 12 * p.s, soff));
                    } else if constexpr (CHAIN) {
                        // (general addressing: the channel may live in the second signal's tensor, ConvParams::xsplit)
                        const int cc = ch * HX_KC + r;
                        const bool second = (p.xsplit > 0) & (cc >= p.xsplit);
                        const int row = second ? cc - p.xsplit : cc;
                        px[i][c] = act_load4_raw(xr, (tok & (r < rows_left)) ? (row * p.ldx + t) * 4 + (second ? xsplit_off : 0) : OOB_OFF, 0);
                    } else {
                        px[i][c] = act_load4_raw(xr, (tok & (r < rows_left)) ? (r * p.ldx + t) * 4 : OOB_OFF, soff);
                    }
                }
                }
            }
        };
        // prologue transform (one FMA: InstanceNorm-apply + speaker bias; LeakyReLU), split, transpose, LDS write
        auto pcommit = [&](int un, const pwin_t (&pw)[ITEMS][8], unsigned tokmask, unsigned char* tile) {
            if (FASTSVC_DBG_ON(p, DBG_NO_COMMIT)) return;
            const int ch = c_ch;
            { int dummy = 0; pos_next(dummy, c_ch); }
            if constexpr (IN1) {
                const pwin_t (&px)[ITEMS][8] = pw;
                #pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const float keep = ((tokmask >> i) & 1u) ? 1.f : 0.f;      // rows outside the utterance: the NEXT conv's zero padding
                    float xv[6] = {px[i][1].x, px[i][0].x, px[i][0].y, px[i][0].z, px[i][0].w, px[i][1].y};
                    #pragma unroll
                    for (int q = 0; q < 6; ++q) xv[q] = fmaxf(xv[q], xv[q] * LRELU_SLOPE);
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float e[8];
                        #pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            // (same association as in1_conv_kernel)
                            const float u = (i1b[c] + (i1w[c][0] * xv[j] + i1w[c][1] * xv[j + 1]) + i1w[c][2] * xv[j + 2]) * keep;
                            e[c] = fmaxf(u, u * slope);
                        }
                        hx_commit_slot(tile, hx_lds_off(4 * it_q[i] + j, it_oct[i]), lo_off, e);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                return;
            }
#ifdef FASTSVC_ACT_BF16
            if constexpr (W8) {
                const bool row_end = (p.T & 7) != 0;           // (wave-uniform) the last item of a row holds 4 rows of it
                #pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int nvi = (int)((tokmask >> (4 * i)) & 15u);
                    const f32x4* cf = reinterpret_cast<const f32x4*>(ncoef + (nvi != 0 ? ch * HX_KC + it_oct[i] * 4 : CINp));
                    const f32x4 c0 = cf[0], c1 = cf[1];
                    const float A[4] = {c0.x, c0.z, c1.x, c1.z};
                    const float Bc[4] = {c0.y, c0.w, c1.y, c1.w};
                    f32x4 px[8];
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        act4_t w = pw[i][c];
                        asm volatile("" : "+v"(w));            // (the conversion stays behind the unit barrier, see below)
                        px[c] = act_unpack4(w);
                    }
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float e[4];
                        #pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = px[2 * c + (j >> 2)][j & 3] * A[c] + Bc[c];
                        if (j >= 4 && row_end) {
                            #pragma unroll
                            for (int c = 0; c < 4; ++c) e[c] = nvi > 4 ? e[c] : 0.f;
                        }
                        const int off = hx_lds_off(8 * it_q[i] + j, it_oct[i] >> 1) + (it_oct[i] & 1) * 8;
                        if constexpr (DEC2) {                  // the raw copy for the 1x1 residual conv (see below)
                            u32x2v hr;
                            hr.x = bf16_pack2(e[0], e[1]);
                            hr.y = bf16_pack2(e[2], e[3]);
                            *reinterpret_cast<u32x2v*>(tile + raw_off + off) = hr;
                        }
                        #pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = fmaxf(e[c], e[c] * slope);
                        u32x2v h;
                        h.x = bf16_pack2(e[0], e[1]);
                        h.y = bf16_pack2(e[2], e[3]);
                        *reinterpret_cast<u32x2v*>(tile + off) = h;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                return;
            }
#endif
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                // (A, Bc) of the item's 8 channels: 64 contiguous bytes
                // rows outside the utterance are the conv's zero padding AFTER the prologue: their loads returned 0, so
                // only the additive term has to go - such an item reads the (0, 0) coefficients behind the table (one
                // select on the address instead of eight on the values: the staging waves' instruction count bounds
                // the narrow layers)
                const int nvi = TAILK ? (int)((tokmask >> (3 * i)) & 7u) : (int)((tokmask >> i) & 1u);
                const bool tok = nvi != 0;
                const f32x4* cf = reinterpret_cast<const f32x4*>(ncoef + (tok ? ch * HX_KC + it_oct[i] * 8 : CINp));
                const f32x4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
                const float A[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
                const float Bc[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
                // one time step (= one 16-byte slot of 8 channels per piece) at a time: the transformed values
                // never exist as a second copy of the register set (two steps at a time - the prologue FMA and the
                // LeakyReLU multiply as packed float32 operations on (t, t+1) - saves 0.75 instructions per value and was
                // measured SLOWER in float32 storage: cfg2 1.412 vs 1.388 ms; in bfloat16 storage (round 5) it took 1 % off
                // the C = 24 layers and made up.0.d27 differ from run to run - v_pk_fma_f32 / v_pk_mul_f32 behind the
                // window loads, cause not found - and was dropped)
                f32x4 px[8];                                   // (the conversion of bfloat16 words happens HERE, a unit after the request)
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if constexpr (IN1) px[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    else {
                        // (pinned HERE: hipcc otherwise hoists the conversion - pure VALU on the loaded registers - above the
                        // unit barrier into the half that issued the request, and waits for the request there)
                        act4_t w = pw[i][c];
                        asm volatile("" : "+v"(w));
                        px[c] = act_unpack4(w);
                    }
                }
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float e[8];
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) e[c] = px[c][j] * A[c] + Bc[c];
                    if (TAILK && tail_rows) {                  // (wave-uniform; rows a multiple of 4 long skip it)
                        #pragma unroll
                        for (int c = 0; c < 8; ++c) e[c] = j < nvi ? e[c] : 0.f;
                    }
                    // the raw copy for the 1x1 residual conv: no prologue at all, i.e. (A, Bc) = (sx, 0) and the FMA's
                    // result IS the scaled raw value (a decimating stage never sits behind a norm); committed first,
                    // the LeakyReLU then runs in place
                    if constexpr (DEC2) hx_commit_slot(tile + raw_off, hx_lds_off(4 * it_q[i] + j, it_oct[i]), lo_off, e);
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) e[c] = fmaxf(e[c], e[c] * slope);
                    hx_commit_slot(tile, hx_lds_off(4 * it_q[i] + j, it_oct[i]), lo_off, e);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // ---- XR, second operand: Stretch_S(x2), raw.  Item = (octet of 8 channels, group of XJ input columns): XJ * 8
        // element loads, every column written S times = XROWS consecutive rows of the second window (row r <-> output
        // column t_start + r, the same alignment as the main window; the x2 conv's taps are rows -1 / 0 / +1).  64 groups
        // per tile cover window + halo for every shape (64 * XROWS - XROWS + 1 >= 253 rows); rows that fall outside the
        // window go to the spare row behind it, columns outside the utterance read 0 = the conv's zero padding.
        constexpr int XJ = (XR && S == 2) ? 2 : 1, XROWS = XJ * (XR ? S : 1);
        const __amdgpu_buffer_rsrc_t x2r = XR ? act_rsrc(p.x2, (long)b * p.x2_b, (long)p.CIN * p.ldx2) : xr;
        const int x2T = !XR ? 0 : p0.lens ? p0.lens[b] * p0.x2len_mul : p.x2_T;
        const int xoct = ptid & 3, xg = ptid >> 2;
        // 16-byte-window instances (hx_w8), S = 2: item = (quad of 4 channels, group of 4 input columns) - FOUR 8-byte requests
        // per item where the element loads below are 8 two-byte ones per column (a wave's request costs the CU's memory pipe
        // about the same whatever its width: this operand, half of the main one's bytes, took two thirds of the staging
        // waves' requests); 32 groups per tile cover window + halo for every shape.  The words are bfloat16 already: the
        // commit is two byte-permutes per column and S 8-byte LDS writes; rows outside the window go to the lane's own
        // 8 bytes of the spare rows.  Measured at cfg3 (round 6): up.0.d3x (S = 2, C = 192) 210 -> 186 us; S = 4: 259 -> 260
        // (C = 96), 508 -> 518 (C = 48); S = 5 (C = 24): 885 -> 910 - there ONE wave commits 4 S rows per lane where four
        // shared S each, and the staging waves' instruction count is what paces the narrow layers: S = 2 only.
        constexpr bool XW = XR && W8 && S == 2;
        const int xq4 = ptid & 7, xg4 = ptid >> 3;
        auto ploadX = [&](int un, act1_t (&px)[XJ][8]) {
#ifdef FASTSVC_ACT_BF16
            if constexpr (XW) {
                const int tl = xl_tl, ch = xl_ch;
                pos_next(xl_tl, xl_ch);
                const int t_start = (tile0 + tl) * NT - halo_al;                  // >= -8
                const int j = ((((t_start + 8 * S) / S) - 8) & ~3) + 4 * xg4;     // floor(t_start / S), down to a multiple of 4
                // (row ends are multiples of 4 in these instances: a group lies inside the utterance or outside it)
                const bool ok = ((unsigned)j < (unsigned)x2T) & (un < nunits);
                const int rows_left = p.CIN - ch * HX_KC;
                #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = xq4 * 4 + c;
                    const u32x2v w = __builtin_amdgcn_raw_buffer_load_b64(x2r, (ok & (r < rows_left)) ? (r * p.ldx2 + j) * 2 : OOB_OFF,
                                                                         ch * HX_KC * p.ldx2 * 2, FASTSVC_LD_AUX);
                    px[0][2 * c] = w.x;
                    px[0][2 * c + 1] = w.y;
                }
                return;
            }
#endif
            if constexpr (XR) {
                const int tl = xl_tl, ch = xl_ch;        // (the odd units)
                pos_next(xl_tl, xl_ch);
                const int t_start = (tile0 + tl) * NT - halo_al;
                const int g0 = (t_start + 8 * XROWS) / XROWS - 8;                 // floor(t_start / XROWS), t_start >= -28
                const int soff = ch * HX_KC * p.ldx2 * 4;
                const int rows_left = p.CIN - ch * HX_KC;
                #pragma unroll
                for (int jj = 0; jj < XJ; ++jj) {
                    const int j = (g0 + xg) * XJ + jj;
                    const bool ok = ((unsigned)j < (unsigned)x2T) & (un < nunits);
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int r = xoct * 8 + c;
                        px[jj][c] = act_load1_raw(x2r, (ok & (r < rows_left)) ? (r * p.ldx2 + j) * 4 : OOB_OFF, soff);
                    }
                }
            }
        };
        auto pcommitX = [&](int un, const act1_t (&pw)[XJ][8], unsigned char* tile) {
#ifdef FASTSVC_ACT_BF16
            if constexpr (XW) {
                const int tl = xc_tl;
                pos_next(xc_tl, xc_ch);
                const int t_start = (tile0 + tl) * NT - halo_al;
                const int j = ((((t_start + 8 * S) / S) - 8) & ~3) + 4 * xg4;
                const int r0 = j * S - t_start;
                const int park = W * HX_ROW + (ptid & 63) * 8;
                #pragma unroll
                for (int k = 0; k < 4; ++k) {
                    u32x2v h;
                    h.x = __builtin_amdgcn_perm(pw[0][2 + (k >> 1)], pw[0][0 + (k >> 1)], (k & 1) ? 0x07060302u : 0x05040100u);
                    h.y = __builtin_amdgcn_perm(pw[0][6 + (k >> 1)], pw[0][4 + (k >> 1)], (k & 1) ? 0x07060302u : 0x05040100u);
                    #pragma unroll
                    for (int ph = 0; ph < S; ++ph) {
                        const int r = r0 + k * S + ph;
                        const int off = (unsigned)r < (unsigned)W ? hx_lds_off(r, xq4 >> 1) + (xq4 & 1) * 8 : park;
                        *reinterpret_cast<u32x2v*>(tile + off) = h;
                    }
                }
                return;
            }
#endif
            if constexpr (XR) {
                const int tl = xc_tl;
                pos_next(xc_tl, xc_ch);
                const int t_start = (tile0 + tl) * NT - halo_al;
                const int g0 = (t_start + 8 * XROWS) / XROWS - 8;
                const int r0 = (g0 + xg) * XROWS - t_start;
                #pragma unroll
                for (int jj = 0; jj < XJ; ++jj) {
                    float px[XJ][8];
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) px[jj][c] = act_unpack1(pw[jj][c]);
                    hx8 h;
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) h[c] = (hx_t)(HX_NP == 2 ? px[jj][c] * sx2 : px[jj][c]);
                    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
                    u32x4w lp = {0u, 0u, 0u, 0u};
                    if constexpr (HX_NP == 2) {
                        const u32x4w hp = __builtin_bit_cast(u32x4w, h);
                        #pragma unroll
                        for (int k = 0; k < 4; ++k) lp[k] = hx_lo_pair(hp[k], px[jj][2 * k] * sx2, px[jj][2 * k + 1] * sx2);
                    }
                    #pragma unroll
                    for (int ph = 0; ph < S; ++ph) {
                        const int r = r0 + jj * S + ph;
                        const int off = hx_lds_off((unsigned)r < (unsigned)W ? r : W, xoct);
                        *reinterpret_cast<hx8*>(tile + off) = h;
                        if constexpr (HX_NP == 2) *reinterpret_cast<u32x4w*>(tile + lo_off + off) = lp;
                    }
                }
            }
        };
        if constexpr (XR) {
            pwin_t pa[ITEMS][8];
            act1_t xb[XJ][8];
            unsigned oka = 0;
            // one register set per operand, each re-requested right after its commit: the main window of tile t + 1 is
            // requested under tile t's x2 unit (the long one: it carries the consumers' epilogue) and committed under tile
            // t + 1's - a whole tile of lead (requested in the half BEFORE its commit it had the main unit's 1-2k cycles)
            // (C >= 48: measured 558 -> 539 us on up.2.d3x at cfg3 bfloat16.  C = 24 keeps the request in the half before the
            // commit: with the whole-tile lead it ran 940 -> 967 us - its window requests then fall into the consumers' epilogue)
            pload(0, pa, oka);
            ploadX(1, xb);
            stamp(2);
            setup_shared();
            pcommit(0, pa, oka, tiles);
            unsigned okn = 0;
            if constexpr (!WSTATIC) pload(2, pa, okn);
            stamp(3);
            __syncthreads();                           // unit 0 staged
            stamp(4);
            for (int u = 0; u < nunits; u += 2) {       // (an even number of units: every main unit has its x2 unit)
                if constexpr (WSTATIC) pload(u + 2, pa, oka);
                pcommitX(u + 1, xb, tiles + bufsz);
                if constexpr (!WSTATIC) ploadX(u + 3, xb);
                stamp(5);
                __syncthreads();                       // end of unit u
                stamp(6);
                if constexpr (WSTATIC) ploadX(u + 3, xb); else oka = okn;
                pcommit(u + 2, pa, oka, tiles);
                if constexpr (!WSTATIC) pload(u + 4, pa, okn);
                stamp(5);
                __syncthreads();                       // end of unit u + 1
                stamp(6);
            }
        } else {
        // staged polyphase epilogue: this staging wave runs pass 2 for the channel tiles its consumer partner (wave - 4)
        // leaves it (HxPolyPass2) - descriptors, group and statistics as the consumer has them
        constexpr bool PSPLIT = POLY && hx_poly_staged<MW, NW, MODE, EPI, S>();
        EpiRsrc Rp;
        const bool pact = mg < p.ngroups;
        if constexpr (PSPLIT) {
            const long ct = (long)p.COUT * p.ldy;
            const float* nul = p.bias;
            Rp.y = act_rsrc(p.y ? p.y : nul, p.y ? (long)sig * p.y_sig + (long)b * p.y_b : 0, p.y ? ct : 0);
            const bool has_y2 = (flags & F_AFF_OUT) != 0;
            Rp.y2 = act_rsrc(has_y2 ? p.y2 : nul, has_y2 ? (long)sig * p.y2_sig + (long)b * p.y2_b : 0, has_y2 ? ct : 0);
            const bool has_ss = (flags & (F_STATS | F_AFF_OUT)) != 0;
            Rp.ss = act_rsrc(has_ss ? p.ss_out : nul, has_ss ? (long)b * p.ss_out_b : 0, has_ss ? 2 * ct : 0);
            Rp.res = Rp.y; Rp.r1x = Rp.y;
        }
        auto ppass2 = [&](int un) {
#ifdef FASTSVC_ACT_BF16
            if constexpr (PSPLIT) {
                const int etl = e_tl, ech = e_ch;
                pos_next(e_tl, e_ch);
                if (un < nunits && ech == nch - 1) {            // (wave-uniform) this unit ends a tile
                    const int tcolw = (tile0 + etl) * NT + wave_n * (NW * 16);
                    const unsigned char* Pw = tiles + 2 * bufsz + cw * hx_poly_patch_bytes<MW, NW, S>();
                    HxPolyPass2<MW, NW, EPI, S, hx_poly_mc<MW>(), MW> P2;
                    float s1p[MW], s2p[MW];
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) { s1p[m] = 0.f; s2p[m] = 0.f; }
                    if (pact) P2.request(p, Rp, mg, tcolw, lane);
                    __syncthreads();                           // the consumers' patches are complete
                    if (pact) {
                        P2.run(p, Rp, Pw, s1p, s2p, mg, tcolw, lane);
                        if (flags & F_STATS) {
                            #pragma unroll
                            for (int m = hx_poly_mc<MW>(); m < MW; ++m) {
                                if (lane < 16) {               // (run() left the row sums in lanes 0..15)
                                    const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                                    atomicAdd(&sstat[slot + 0], (double)s1p[m]);
                                    atomicAdd(&sstat[slot + 1], (double)s2p[m]);
                                }
                            }
                        }
                    }
                }
            }
#endif
        };
        pwin_t pa[ITEMS][8], pb[ITEMS][8];
        unsigned oka = 0, okb = 0;                       // per-item "rows inside the utterance" bits
        // No branch may sit between a load and its use (hipcc then counts vmcnt for the path WITHOUT the
        // newer loads and so waits for them too - a full memory latency per unit): loads and commits are
        // unconditional; past the last unit they fetch nothing (offsets out of range) and stage zeros into the
        // buffer nobody reads.
        pload(0, pa, oka);
        pload(1, pb, okb);
        stamp(2);
        setup_shared();
        if constexpr (IN1 && HX_NP == 2) {
            // sx: the scale of the 1 -> C conv's OUTPUT, which is what gets split; a power of two, so the scaled taps
            // give exactly sx times the unscaled result
            #pragma unroll
            for (int c = 0; c < 8; ++c) { i1w[c][0] *= sx; i1w[c][1] *= sx; i1w[c][2] *= sx; i1b[c] *= sx; }
        }
        pcommit(0, pa, oka, tiles);
        stamp(3);
        __syncthreads();                               // unit 0 staged
        stamp(4);
        for (int u = 0; u < nunits; u += 2) {
            pload(u + 2, pa, oka);
            stamp(9);
            pcommit(u + 1, pb, okb, tiles + bufsz);
            stamp(5);
            ppass2(u);
            __syncthreads();                           // end of unit u
            if (CHAIN && (u % nch) == nch - 1) __syncthreads();        // the consumers wrote the intermediate tile
            stamp(6);
            // NO `if (u + 1 >= nunits) break;` here: hipcc folds that exit into the loop latch, so the state "first
            // half only" (this half's loads in flight, their destination registers about to be reused for addresses)
            // reaches the loop head in its bookkeeping and it puts `s_waitcnt vmcnt(0)` in front of the loads above -
            // the two-deep prefetch ran one deep in every instance.  With an odd unit count the second half runs on
            // a phantom unit (loads out of range, zeros into the buffer nobody reads) and the consumers add the
            // matching barrier at their end.
            pload(u + 3, pb, okb);
            stamp(9);
            pcommit(u + 2, pa, oka, tiles);
            stamp(5);
            ppass2(u + 1);
            __syncthreads();                           // end of unit u+1 (or the phantom one)
            if (CHAIN && u + 1 < nunits && ((u + 1) % nch) == nch - 1) __syncthreads();
            stamp(6);
        }
        }
    } else {
        // ================================ CONSUMER WAVES ================================
        f32x4 acc[(POLY || DEC2 || UPH) ? 1 : NW][MW];
        f32x4 acc3[3][(POLY || UPH) ? NW : 1][MW];    // polyphase: a / z / c accumulator sets
        f32x4 acc2[2][DEC2 ? NW : 1][MW];             // decimating pair: k=3 / 1x1
        constexpr bool XSPLIT = XR && HX_NP == 2;     // second operand, float32 storage: its own accumulator set (other operand scales)
        f32x4 accX[XSPLIT ? NW : 1][MW];
        float s1[MW], s2[MW];
        // small tiles: the lane's InstanceNorm sums of ALL tiles stay in registers and the rows are joined once, after the
        // loop (larger tiles have no registers to spare: per tile, through LDS atomics)
        constexpr bool GSTAT = MODE == MODE_DIRECT && (MW * NW <= 6 || (MW == 3 && NW == 4));   // (3 x 4: 202 -> 214 registers of 256)
        double g1[GSTAT ? MW : 1], g2[GSTAT ? MW : 1];
        #pragma unroll
        for (int m = 0; m < (GSTAT ? MW : 1); ++m) g1[m] = g2[m] = 0.0;
        HxWeightStream<NSLOT> wst;
        const int wunits = UPH ? nch + 2 * p.nch32b : CHAIN ? nch + p.nch32b : XR ? 2 * nch : nch;    // weight units per tile (CHAIN: first conv's, then the second's; XR: interleaved)
        wst.init(reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                     (long)(active ? mg : 0) * wunits * NSLOT * HX_FRAG, WLB ? nch : wunits, lane);
        int aoff[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            aoff[tap] = hx_lds_off((halo_al - halo) + tap * halo + wave_n * (NW * 16) + (lane & 15), lane >> 4);
        int aoffX[3];                                  // XR: taps of the second operand's conv (dilation 1, same window alignment)
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            aoffX[tap] = XR ? hx_lds_off((halo_al - 1) + tap + wave_n * (NW * 16) + (lane & 15), lane >> 4) : 0;
        // XR with one K chunk (WSTATIC): the second conv's fragments stay in registers like the first one's
        HxWeightStream<NSLOT> wstX_static;
        if constexpr (XR && WSTATIC)
            wstX_static.init(reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                                 ((long)(active ? mg : 0) * wunits + 1) * NSLOT * HX_FRAG, 1, lane);
        HxWeightStream<NSLOT>& wstX = (XR && WSTATIC) ? wstX_static : wst;
        EpiRsrc R;
        {
            const long ct = (long)p.COUT * p.ldy;
            const float* nul = p.bias;
            R.y = act_rsrc(p.y ? p.y : nul, p.y ? (long)sig * p.y_sig + (long)b * p.y_b : 0, p.y ? ct : 0);
            const bool has_y2 = DEC2 || (flags & F_AFF_OUT) != 0;                     // (MODE_UPHEAD: y = xr, y2 = u1)
            R.y2 = act_rsrc(has_y2 ? p.y2 : nul, has_y2 ? (long)sig * p.y2_sig + (long)b * p.y2_b : 0, has_y2 ? ct : 0);
            R.res = act_rsrc(p.res ? p.res : nul, p.res ? (long)sig * p.res_sig + (long)b * p.res_b : 0, p.res ? ct : 0);
            const bool has_ss = (flags & (F_STATS | F_AFF_OUT)) != 0;
            R.ss = act_rsrc(has_ss ? p.ss_out : nul, has_ss ? (long)b * p.ss_out_b : 0, has_ss ? 2 * ct : 0);
            R.r1x = make_rsrc(p.r1x ? p.r1x + (long)sig * p.r1x_sig + (long)b * p.r1x_b : nul, p.r1x ? p.ldy : 0);
        }
        float k_bias[MW], k_bias2[MW], k_r1w[MW], k_r1b[MW];
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int cot = (mg * MW + m) * 16 + (lane & 15);
            const int co = cot < p.COUT ? cot : 0;
            const bool cok = active && cot < p.COUT;
            k_bias[m] = cok ? p.bias[(long)sig * p.bias_sig + co] : 0.f;
            k_bias2[m] = 0.f; k_r1w[m] = 0.f; k_r1b[m] = 0.f;
            if constexpr (DEC2 || UPH) k_bias2[m] = cok ? p.bias2[(long)sig * p.bias2_sig + co] : 0.f;
            if constexpr (XR) k_bias[m] += cok ? p.bias2[co] : 0.f;        // both convs' biases join the one accumulator
            if constexpr (EPI == EPI_RANK1) {
                k_r1w[m] = cok ? p.r1w[(long)sig * p.r1_sig + co] : 0.f;
                k_r1b[m] = cok ? p.r1b[(long)sig * p.r1_sig + co] : 0.f;
            }
        }
        // (the inverse operand scales of this lane's channels live in LDS, s_inv, filled after setup_shared: the
        // activation scale of a normalised edge is only known then.  MODE_CHAIN: they belong to the SECOND conv.)
        const EpiConst<MW, true, HX_NP == 2> K{k_bias, k_bias2, k_r1w, k_r1b, s_inv + wave_m * (MW * 16) + (lane & 15), 16 * MW * WM};
        float t_inv[MW], t_inv2[MW];                   // (requested here with the other constants; dead after setup)
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int cot = (mg * MW + m) * 16 + (lane & 15);
            const bool cok = active && cot < p.COUT;
            t_inv[m] = cok ? 1.f : 0.f; t_inv2[m] = t_inv[m];
            if constexpr (HX_NP == 2) {
                if (invtab) {
                    t_inv[m] = cok ? invtab[(CHAIN ? ninv : 0) + cot] : 0.f;
                    if constexpr (DEC2) t_inv2[m] = cok ? invtab[ninv + cot] : 0.f;
                    if constexpr (UPH) t_inv2[m] = cok ? invtab[2 * ninv + cot] : 0.f;
                    if constexpr (XR) t_inv2[m] = cok ? invtab[ninv + cot] : 0.f;
                }
            }
        }
        constexpr bool LAST_OK = MODE == MODE_DIRECT && EPI == EPI_RES && WM == 1;     // conv_last may ride on this instance
        float k_last[MW];
        // (wave-uniform: kept in a scalar register - the float32 <2,3> instance has no vector register to spare)
        const float b_last = (LAST_OK && p.last_w) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.last_b[0]))) : 0.f;
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int cot = (mg * MW + m) * 16 + (lane & 15);
            k_last[m] = (LAST_OK && p.last_w && active && cot < p.COUT) ? p.last_w[cot] : 0.f;
        }
        constexpr bool EST = hx_estage<MW, NW, MODE, EPI>();
        // this wave's epilogue-operand slots, behind the tile buffers (hx_launch_direct sizes them)
        constexpr bool PAIRS = hx_pairs_epi<MW, NW, MODE>();
        constexpr int EST_WAVE_FLOATS = EST ? (PAIRS ? MW * (NW / 2) * 256 : MW * NW * EST_ITEM_FLOATS) : 0;   // per operand
        const int est_ops = EPI == EPI_RES ? 1 : p.res ? 3 : 2;
        const float* Ew = reinterpret_cast<const float*>(tiles + 2 * bufsz) + cw * (est_ops * EST_WAVE_FLOATS);
        // EST2 (one K chunk per tile: WSTATIC): the operands of tile t + 1 are requested BEFORE tile t's products into a
        // second slot set - requested in front of their own tile's products they had one short matrix loop (200-600
        // cycles) to cover a memory round trip, i.e. every tile's epilogue started with an exposed HBM latency
        constexpr bool EST2 = hx_est2<MW, NW, MODE, EPI>() && WSTATIC && !XR;
        const int est_set = 4 * est_ops * EST_WAVE_FLOATS;         // floats of one slot set (all four waves)
        // bfloat16 pair epilogue: this wave's re-layout patch behind every wave's operand slots
        float* Xw = const_cast<float*>(reinterpret_cast<const float*>(tiles + 2 * bufsz)) + (EST2 ? 2 : 1) * est_set +
                    cw * (16 * 36);
        (void)Xw;
        stamp(2);
        setup_shared();
        if constexpr (HX_NP == 2) {
            const float isx = 1.0f / (CHAIN ? s_mid_scale : sx);   // exact: a power of two
            if (lane < 16) {
                #pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int li = (wave_m * MW + m) * 16 + lane;
                    s_inv[li] = t_inv[m] * isx;
                    if constexpr (DEC2 || UPH) s_inv[16 * MW * WM + li] = t_inv2[m] * isx;
                }
            }
        }
        // XR, float32 storage: the second conv accumulates in its own set (its operands carry other power-of-two scales);
        // the set joins the main one before the epilogue, times (inverse scales of x2's conv) / (those of the main conv) -
        // a power of two, from the exponent fields
        float xratio[MW];
        #pragma unroll
        for (int m = 0; m < MW; ++m) {
            xratio[m] = 0.f;
            if constexpr (XR && HX_NP == 2) {
                auto ex = [](float v) { return (int)((__builtin_bit_cast(unsigned, v) >> 23) & 0xffu); };
                const int e = ex(t_inv2[m]) - ex(t_inv[m]) + ex(sx) - ex(sx2) + 127;
                if (t_inv[m] != 0.f) xratio[m] = __builtin_bit_cast(float, (unsigned)min(254, max(1, e)) << 23);
            }
        }
        __syncthreads();                               // unit 0 staged (and s_inv visible)
        stamp(4);
        int u = 0;
        if constexpr (CHAIN) {
            f32x4 accA[NW + 1][MW];                    // the first conv's tile; time tile NW only in the last wave along time
            const int cmidp = p.nch32b * HX_KC;
            // terms that join the SECOND conv's accumulator directly (rank-1 / tensor residual) are multiplied by
            // 1 / inverse scale: exact for a power of two, from its bits (exponent field e -> 254 - e)
            auto fwd = [&](int m) -> float {
                if constexpr (HX_NP == 2) return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, K.inv(m)));
                else return 1.f;
            };
            // WSTATIC (one K chunk per conv: C <= 32): both convs' fragments stay in registers for the whole
            // workgroup - `wst` holds the first conv's, `wstB` the second's; nothing is re-requested
            HxWeightStream<NSLOT> wstB_static;
            if constexpr (WSTATIC)
                wstB_static.init(reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                                     ((long)(active ? mg : 0) * wunits + 1) * NSLOT * HX_FRAG, 1, lane);
            HxWeightStream<NSLOT>& wstB = WSTATIC ? wstB_static : wst;
            int aoffB[3];
            #pragma unroll
            for (int tap = 0; tap < 3; ++tap)
                aoffB[tap] = hx_lds_off((HB - p.dil2) + tap * p.dil2 + wave_n * (NW * 16) + (lane & 15), lane >> 4);
            constexpr int lo_offB = T2ROWS * HX_ROW;
            const bool extra = wave_n == WN - 1;
            for (int tl = 0; tl < ntiles; ++tl) {
                const int tcol0 = (tile0 + tl) * NT + wave_n * (NW * 16);
                // rank-1 residual (1-channel stage input): its row is fetched before the first conv, so that it
                // is OLDER than every weight request the second conv waits for anyway
                f32x4 rx[EPI == EPI_RANK1 ? NW : 1];
                if constexpr (EPI == EPI_RANK1) {
                    #pragma unroll
                    for (int n = 0; n < NW; ++n) {
                        const int t = tcol0 + n * 16 + (lane >> 4) * 4;
                        rx[n] = buf_load4(R.r1x, (active && t < p.T) ? t * 4 : OOB_OFF, 0);
                    }
                }
                #pragma unroll
                for (int n = 0; n <= NW; ++n)
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) accA[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                // (no branch around the units and do-while loops: any path on which the loops' weight requests are
                // not issued makes hipcc treat the residual loads as the YOUNGEST requests where they are used, i.e.
                // wait for every weight request behind them - a memory latency per tile; FASTSVC_DBG has no
                // no-MFMA switch in this mode for the same reason)
                int ch = 0;
                do {
                    hx_unit_direct<MW, NW + 1, !WSTATIC, true>(accA, tiles + (u & 1) * bufsz, aoff, lo_off, wst, extra);
                    stamp(7);
                    __syncthreads();                   // end of unit u
                    stamp(6);
                    ++u;
                } while (++ch < nch);
                // every wave is past the previous tile's second conv (the barriers above): its tile may be overwritten
                if (active)
                    hx_chain_store<MW, NW + 1, UPH>(accA, extra ? NW + 1 : NW, T2, T2CHUNK, lo_offB, mg, wave_n * (NW * 16),
                                                    (tile0 + tl) * NT - HB + wave_n * (NW * 16), p.T, s_mid, cmidp, lane, t2one);
                if constexpr (UPH) {
                    __syncthreads();                   // intermediate tiles complete
                    stamp(5);
                    // residual branch: xr = poly(a) + bias  (plain epilogue, no LeakyReLU, no FiLM)
                    ConvParams pr = p;
                    pr.flags = 0;
                    EpiRsrc Rr = R;
                    Rr.y2 = R.res;                     // (absent: zero-length descriptor)
                    #pragma unroll
                    for (int k = 0; k < 3; ++k)
                        #pragma unroll
                        for (int n = 0; n < NW; ++n)
                            #pragma unroll
                            for (int m = 0; m < MW; ++m) acc3[k][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                    int cb = 0;
                    do { hx_unit_poly<MW, NW, true>(acc3, T2 + t2one + cb * T2CHUNK, aoffB, lo_offB, wst); } while (++cb < p.nch32b);
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                    ws_epilogue_poly<MW, NW, EPI_PLAIN, S, 0>(pr, Rr, acc3, s1, s2, mg, tcol0, active, lane, K);
                    // up branch: u1 = scale * lrelu(poly(lrelu(a)) + bias2) + shift, InstanceNorm partial sums
                    EpiRsrc Ru = R;
                    Ru.y = R.res;                      // (the pre-affine tensor is not written)
                    const EpiConst<MW, true, HX_NP == 2> K2{k_bias2, k_bias2, k_r1w, k_r1b,
                                                            s_inv + 16 * MW * WM + wave_m * (MW * 16) + (lane & 15), 0};
                    #pragma unroll
                    for (int k = 0; k < 3; ++k)
                        #pragma unroll
                        for (int n = 0; n < NW; ++n)
                            #pragma unroll
                            for (int m = 0; m < MW; ++m) acc3[k][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                    cb = 0;
                    do { hx_unit_poly<MW, NW, true>(acc3, T2 + cb * T2CHUNK, aoffB, lo_offB, wst); } while (++cb < p.nch32b);
                    stamp(7);
                    ws_epilogue_poly<MW, NW, EPI_AFF, S, 0>(p, Ru, acc3, s1, s2, mg, tcol0, active, lane, K2);
                    if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
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
                    stamp(8);
                } else {
                // the stage's residual tensor (the 1x1 conv's output) is fetched HERE, into registers the first
                // conv's tile has just left: it lands under the second conv instead of costing the epilogue a
                // memory round trip per item
                f32x4 rres[EPI == EPI_RES ? NW : 1][EPI == EPI_RES ? MW : 1];
                float kf[MW];
                #pragma unroll
                for (int m = 0; m < MW; ++m) kf[m] = (EPI == EPI_RANK1 || EPI == EPI_RES) ? fwd(m) : 1.f;
                #pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const int t = tcol0 + n * 16 + (lane >> 4) * 4;
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) {
                        const int cot = (mg * MW + m) * 16 + (lane & 15);
                        const bool ok = active && cot < p.COUT && t < p.T;
                        if constexpr (EPI == EPI_RES) rres[n][m] = act_load4(R.res, ok ? (cot * p.ldy + t) * 4 : OOB_OFF, 0);
                        if constexpr (EPI == EPI_RANK1) acc[n][m] = (rx[n] * k_r1w[m] + k_r1b[m]) * kf[m];
                        else acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                __syncthreads();                       // intermediate tile complete
                stamp(5);
                int cb = 0;
                do {
                    if constexpr (WLB) hx_unit_direct_wl<MW, NW>(acc, T2 + cb * T2CHUNK, aoffB, lo_offB, Wl + cb * (NSLOT * HX_FRAG), lane);
                    else hx_unit_direct<MW, NW, !WSTATIC>(acc, T2 + cb * T2CHUNK, aoffB, lo_offB, wstB);
                } while (++cb < p.nch32b);
                stamp(7);
                if constexpr (EPI == EPI_RES) {
                    #pragma unroll
                    for (int n = 0; n < NW; ++n)
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) acc[n][m] += rres[n][m] * kf[m];
                }
                #pragma unroll
                for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                ws_epilogue_kind<MW, NW, EPI_PLAIN, false, 0, 1>(p, R, acc, s1, s2, sig, mg, tcol0, active, lane, K, Ew);
                if constexpr (TRACKS) { if (p.amax_out) amax_tile_flush(R); }
                stamp(8);
                }
            }
        } else if constexpr (XR) {
            for (int tl = 0; tl < ntiles; ++tl) {
                #pragma unroll
                for (int n = 0; n < NW; ++n)
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) {
                        acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (XSPLIT) accX[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                const int tcolw = (tile0 + tl) * NT + wave_n * (NW * 16);
                for (int ch = 0; ch < nch; ++ch) {
                    if constexpr (EST) {
                        // the epilogue's scale / shift pieces, two units ahead of it
                        if (active && ch + 1 == nch) {
#ifdef FASTSVC_ACT_BF16
                            if constexpr (PAIRS) hx_epilogue8_stage<MW, NW / 2, EPI>(p, R, Ew, mg, tcolw, lane);
                            else
#endif
                            ws_epilogue_stage<MW, NW, EPI>(p, R, Ew, mg, tcolw, lane);
                        }
                    }
                    if (active) hx_unit_direct<MW, NW, !WSTATIC>(acc, tiles, aoff, lo_off, wst);
                    stamp(7);
                    __syncthreads();                   // end of the main unit
                    stamp(6);
                    ++u;
                    if (active) {
                        if constexpr (XSPLIT) hx_unit_direct<MW, NW, !WSTATIC>(accX, tiles + bufsz, aoffX, lo_off, wstX);
                        else hx_unit_direct<MW, NW, !WSTATIC>(acc, tiles + bufsz, aoffX, lo_off, wstX);
                    }
                    stamp(7);
                    if (ch + 1 == nch) {
                        if constexpr (XSPLIT) {
                            #pragma unroll
                            for (int n = 0; n < NW; ++n)
                                #pragma unroll
                                for (int m = 0; m < MW; ++m) acc[n][m] += accX[n][m] * xratio[m];
                        }
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                        if constexpr (EST) { if (active) ws_epilogue_stage_wait<NSLOT>(!WSTATIC); }
#ifdef FASTSVC_ACT_BF16
                        if constexpr (PAIRS) hx_epilogue8<MW, NW / 2, EPI, EST, false, false>(p, R, acc, s1, s2, sig, mg, tcolw, active, lane, K, Ew, Xw);
                        else
#endif
                        ws_epilogue_kind<MW, NW, EPI, EST, 0, -1>(p, R, acc, s1, s2, sig, mg, tcolw, active, lane, K, Ew);
                        if (flags & F_STATS) {
                            #pragma unroll
                            for (int m = 0; m < MW; ++m) {
                                if constexpr (GSTAT) { g1[m] += (double)s1[m]; g2[m] += (double)s2[m]; }
                                else {
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
                    }
                    stamp(8);
                    __syncthreads();                   // end of the x2 unit
                    stamp(6);
                    ++u;
                }
            }
        } else
        for (int tl = 0; tl < ntiles; ++tl) {
            auto est_stage = [&](const float* slots, int tile) {
                if constexpr (EST) {
#ifdef FASTSVC_ACT_BF16
                    if constexpr (PAIRS) hx_epilogue8_stage<MW, NW / 2, EPI>(p, R, slots, mg, tile * NT + wave_n * (NW * 16), lane);
                    else
#endif
                    ws_epilogue_stage<MW, NW, EPI>(p, R, slots, mg, tile * NT + wave_n * (NW * 16), lane);
                }
            };
            #pragma unroll
            for (int n = 0; n < NW; ++n)
                #pragma unroll
                for (int m = 0; m < MW; ++m) {
                    if constexpr (POLY) { acc3[0][n][m] = acc3[1][n][m] = acc3[2][n][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    else if constexpr (DEC2) { acc2[0][n][m] = acc2[1][n][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    else acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            const float* EwT = EST2 ? Ew + (tl & 1) * est_set : Ew;       // this tile's slot set
            for (int ch = 0; ch < nch; ++ch, ++u) {
                if constexpr (EST2) {
                    if (active && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
                        if (tl == 0) est_stage(Ew, tile0);
                        if (tl + 1 < ntiles) est_stage(Ew + ((tl + 1) & 1) * est_set, tile0 + tl + 1);
                    }
                } else if constexpr (EST) {
                    // one unit earlier when the tile has several K chunks: more time to land
                    if (active && ch == max(nch - 2, 0) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) est_stage(Ew, tile0 + tl);
                }
                if (active && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA))) {
                    if constexpr (POLY) hx_unit_poly<MW, NW, !WSTATIC>(acc3, tiles + (u & 1) * bufsz, aoff, lo_off, wst);
                    else if constexpr (DEC2) hx_unit_dec2<MW, NW, !WSTATIC>(acc2, tiles + (u & 1) * bufsz, aoff, lo_off, raw_off, wst);
                    else hx_unit_direct<MW, NW, !WSTATIC>(acc, tiles + (u & 1) * bufsz, aoff, lo_off, wst);
                }
                stamp(7);
                if (ch + 1 == nch) {
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                    // the staged pieces are older than the NSLOT ring re-requests of this unit
                    if constexpr (EST2) {
                        // This tile's pieces have landed when everything but the wave's YOUNGER requests is back: the next
                        // tile's pieces and - issued behind this tile's pieces, a tile ago - the previous tile's STORES.  Counting
                        // only the pieces made every tile wait for those stores' acknowledgement (timeline of up.3.d9: 1.4k of 6.4k
                        // cycles per tile between the products and the epilogue).  The store count is the guaranteed minimum
                        // (a straddling row end stores twice: more younger requests only make the wait stricter).
                        if (active) {
                            constexpr int PER_OP = PAIRS ? MW * (NW / 2) : MW * NW * EST_DMA_PER_ITEM;
                            constexpr int EITEMS = PAIRS ? MW * (NW / 2) : MW * NW;
                            const int ndma = tl + 1 < ntiles ? (EPI == EPI_RES ? 1 : p.res ? 3 : 2) * PER_OP : 0;
                            int nst = 0;
                            if (tl > 0 && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
                                if (EPI == EPI_AFF) nst = EITEMS * (p.y ? 2 : 1);
                                else nst = (p.y ? EITEMS : 0) + ((LAST_OK && p.last_w) ? NW : 0);
                            }
                            hx_wait_vmcnt(ndma + nst);
                        }
                    } else
                    if constexpr (EST) { if (active) ws_epilogue_stage_wait<NSLOT>(!WSTATIC && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA))); }
                    stamp(10);                         // (timeline build: the staged operands have landed)
                    if constexpr (POLY) {
#ifdef FASTSVC_ACT_BF16
                        if constexpr (hx_poly_staged<MW, NW, MODE, EPI, S>()) {
                            // pass 1, workgroup barrier, then this wave's share of pass 2 (the staging partner runs the rest)
                            const int tcolw = (tile0 + tl) * NT + wave_n * (NW * 16);
                            unsigned char* Pw = tiles + 2 * bufsz + cw * hx_poly_patch_bytes<MW, NW, S>();
                            HxPolyPass2<MW, NW, EPI, S, 0, hx_poly_mc<MW>()> P2;
                            if (active) {
                                P2.request(p, R, mg, tcolw, lane);
                                hx_poly_pass1<MW, NW, S>(p, acc3, mg, lane, K, Pw);
                            }
                            __syncthreads();               // every wave's patch is complete
                            if (active) P2.run(p, R, Pw, s1, s2, mg, tcolw, lane);
                        } else
                            hx_epilogue_poly8<MW, NW, EPI, S, TAILK>(p, R, acc3, s1, s2, mg, (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
#else
                        ws_epilogue_poly<MW, NW, EPI, S, TAILK ? 2 : 0>(p, R, acc3, s1, s2, mg, (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
#endif
                    }
                    else if constexpr (DEC2)
                    {
#ifdef FASTSVC_ACT_BF16
                        if constexpr (hx_dec2_staged<MW, NW, MODE>())
                            hx_epilogue_dec2_staged<MW, NW>(p, R, acc2, sig, mg, (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K,
                                                            tiles + 2 * bufsz + cw * hx_dec2_patch_bytes<MW, NW>());
                        else
#endif
                        ws_epilogue_dec2<MW, NW>(p, R, acc2, sig, mg, (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
                    }
                    else {
#ifdef FASTSVC_ACT_BF16
                        if constexpr (PAIRS) {
                            if (!(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE)))
                                hx_epilogue8<MW, NW / 2, EPI, EST, TAILK>(p, R, acc, s1, s2, sig, mg,
                                                                   (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K, EwT, Xw);
                        } else
#endif
                        ws_epilogue_kind<MW, NW, EPI, EST, TAILK ? 2 : 0, -1>(p, R, acc, s1, s2, sig, mg,
                                                              (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K, EwT);
                        if constexpr (LAST_OK) {
                            if (p.last_w && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE)))
                                hx_last_reduce<MW, NW, PAIRS>(p, acc, k_last, b_last, b, (tile0 + tl) * NT + wave_n * (NW * 16), lane);
                        }
                    }
                    stamp(11);                         // (timeline build: tile epilogue issued)
                    if constexpr (TRACKS) { if (p.amax_out) amax_tile_flush(R); }
                    if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) {
                            if constexpr (GSTAT) { g1[m] += (double)s1[m]; g2[m] += (double)s2[m]; }
                            else {
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
                }
                stamp(8);
                __syncthreads();                       // end of unit u
                stamp(6);
            }
        }
        if constexpr (TRACKS) amax_flush(p, R, &s_amax, &s_cnt, 4, sig, b, lane, blockIdx.x);   // (float32 storage: the next conv's split-binary16 scale)
        if constexpr (GSTAT) {
            // the lanes' InstanceNorm partial sums of all tiles: rows joined once per workgroup (per tile that was two
            // cross-row exchanges and two LDS atomics per channel tile inside a divergent region)
            if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {
                #pragma unroll
                for (int m = 0; m < MW; ++m) {
                    double a1 = g1[m], a2 = g2[m];
                    a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
                    a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
                    if (active && lane < 16) {
                        const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                        atomicAdd(&sstat[slot + 0], a1);
                        atomicAdd(&sstat[slot + 1], a2);
                    }
                }
            }
        }
        if (nunits & 1) __syncthreads();               // the staging waves' loop runs in pairs of units
    }
    if ((flags & F_STATS) && !(FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE))) {   // one f64 global atomic per channel per workgroup
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += 512) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}

constexpr size_t HX_STATIC_LDS = 4096;                  // upper bound of conv_hx_kernel's static LDS (s_inv, s_mid)
template <auto KERNEL>
static hipError_t hx_launch_instance(dim3 grid, size_t smem, hipStream_t stream, const ConvParams& p) {
    if (smem + HX_STATIC_LDS > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - HX_STATIC_LDS);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(KERNEL, grid, dim3(512), smem, stream, p);
    return hipGetLastError();
}

// instances compiled with the row-end handling (TAILK): the layers that run at the frame rate or twice it - C >= 96
// in the recipe's configuration, i.e. MW = 3 - in both storages
constexpr bool hx_tail_instance(int MW, int MODE, int EPI, int S) {
    return MW == 3 && ((MODE == MODE_DIRECT && EPI != EPI_RANK1) || (MODE == MODE_POLY && S != 5) || MODE == MODE_DEC2);
}

template <int MW, int NW, int WM, int WN, int MODE, int EPI, int S>
static hipError_t hx_launch_kind(dim3 grid, size_t smem, hipStream_t stream, const ConvParams& p) {
    if (p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0)) {
        if constexpr (hx_tail_instance(MW, MODE, EPI, S))
            return hx_launch_instance<&conv_hx_kernel<MW, NW, WM, WN, MODE, EPI, S, false, true>>(grid, smem, stream, p);
        else return hipErrorInvalidValue;              // (run_conv asks conv_hx_tail_ok first)
    }
    // the ring holds the whole layer when there is a single K chunk: nothing to re-request (MW = 2 layers: C_in <= 32)
    if constexpr (MW == 2 && MODE != MODE_UPHEAD) {
        if (p.nch32 == 1 && ((MODE != MODE_CHAIN && MODE != MODE_CHAIN1) || p.nch32b == 1)) return hx_launch_instance<&conv_hx_kernel<MW, NW, WM, WN, MODE, EPI, S, true>>(grid, smem, stream, p);
    }
    return hx_launch_instance<&conv_hx_kernel<MW, NW, WM, WN, MODE, EPI, S, false>>(grid, smem, stream, p);
}

template <int MW, int NW, int WM, int WN, int MODE>
static hipError_t hx_launch_shape(const ConvParams& p, int nsig, hipStream_t stream) {
    constexpr int NT = 16 * NW * WN;
    const int ntx = (p.T + NT - 1) / NT;
    const int tpw = p.tpw > 0 ? p.tpw : 1;
    dim3 grid((ntx + tpw - 1) / tpw, (p.ngroups + WM - 1) / WM, nsig * p.B);
    constexpr bool CHAIN = MODE == MODE_CHAIN || MODE == MODE_CHAIN1 || MODE == MODE_UPHEAD;
    const bool w8 = hx_w8(MODE, p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0),     // (the instance hx_launch_kind picks)
                          MODE == MODE_DEC2 ? ((p.s == 1 && (p.ldx & 3) == 0) ? 2 : 1) : 0);
    const int halo = (MODE == MODE_DIRECT || CHAIN) ? p.dil : 1;
    const int halo_al = w8 ? ((halo + 7) & ~7) : ((halo + 3) & ~3);
    const int W = NT + (CHAIN ? 16 : 0) + 2 * halo_al;
    const int spare = w8 ? 8 : 4;
    const size_t smem = sizeof(double) * 2 * 16 * MW * WM + sizeof(float) * 2 * ((size_t)p.nch32 * HX_KC + 8) +
                        (size_t)2 * (MODE == MODE_DEC2 ? 2 : 1) * HX_NP * (W + spare) * HX_ROW +
                        (CHAIN ? (size_t)(MODE == MODE_UPHEAD ? 2 : 1) * p.nch32b * HX_NP * (NT + 16) * HX_ROW : 0);
    const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
    if constexpr (MODE == MODE_UPHEAD) {
#ifndef FASTSVC_ACT_BF16
        if (!(p.flags & F_AFF_OUT) || !p.y || !p.y2 || !p.bias2 || smem + HX_STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
#define FASTSVC_HXU(sv) if (p.s == sv) return hx_launch_kind<MW, NW, WM, WN, MODE_UPHEAD, EPI_AFF, sv>(grid, smem, stream, p);
        FASTSVC_HXU(2) FASTSVC_HXU(4) FASTSVC_HXU(5)
#undef FASTSVC_HXU
#endif
        return hipErrorInvalidValue;
    } else if constexpr (MODE == MODE_CHAIN1) {
        if (aff || !p.r1x || smem + HX_STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
        return hx_launch_kind<MW, NW, WM, WN, MODE_CHAIN1, EPI_RANK1, 1>(grid, smem, stream, p);
    } else if constexpr (MODE == MODE_CHAIN) {
        if (aff || smem + HX_STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
        const int kind = p.r1x ? EPI_RANK1 : p.res ? EPI_RES : EPI_PLAIN;
        // S = 2: the second conv's weights in LDS - one channel group, more than one K chunk (a single chunk stays
        // in registers), room next to the tiles, and no residual tensor: with one the phase the weights' L2 latency
        // stretched was ALSO what hid the residual loads' latency (4-7k cycles under load; fetched earlier they
        // block the first conv's weight ring, the queue being in order) - down.1.c23: 73.8 µs either way
        const size_t wl = (size_t)p.nch32b * 3 * MW * HX_NP * HX_FRAG;
        if constexpr (WM == 1 && MW == 3) {
            if (kind == EPI_PLAIN && p.ngroups == 1 && p.nch32b > 1 && smem + wl + HX_STATIC_LDS <= 160 * 1024)
                return hx_launch_kind<MW, NW, WM, WN, MODE_CHAIN, EPI_PLAIN, 2>(grid, smem + wl, stream, p);
        }
#define FASTSVC_HXC(k) if (kind == k) return hx_launch_kind<MW, NW, WM, WN, MODE_CHAIN, k, 1>(grid, smem, stream, p);
        FASTSVC_HXC(EPI_PLAIN) FASTSVC_HXC(EPI_RES) FASTSVC_HXC(EPI_RANK1)
#undef FASTSVC_HXC
        return hipErrorInvalidValue;
    } else if constexpr (MODE == MODE_DEC2) {
        // S = 2: compact input (p.s == 1, rows a multiple of 4 long, 16-byte aligned) - vector window loads
        const size_t patch = hx_dec2_staged<MW, NW, MODE_DEC2>() ? 4 * (size_t)hx_dec2_patch_bytes<MW, NW>() : 0;   // (bfloat16 storage)
        if (p.s == 1 && (p.ldx & 3) == 0) return hx_launch_kind<MW, NW, WM, WN, MODE_DEC2, EPI_PLAIN, 2>(grid, smem + patch, stream, p);
        return hx_launch_kind<MW, NW, WM, WN, MODE_DEC2, EPI_PLAIN, 1>(grid, smem + patch, stream, p);
    } else if constexpr (MODE == MODE_POLY) {
#define FASTSVC_HXP(sv) \
        if (p.s == sv) return aff ? hx_launch_kind<MW, NW, WM, WN, MODE_POLY, EPI_AFF, sv>(grid, smem + (hx_poly_staged<MW, NW, MODE_POLY, EPI_AFF, sv>() ? 4 * hx_poly_patch_bytes<MW, NW, sv>() : 0), stream, p) \
                                  : hx_launch_kind<MW, NW, WM, WN, MODE_POLY, EPI_PLAIN, sv>(grid, smem + (hx_poly_staged<MW, NW, MODE_POLY, EPI_PLAIN, sv>() ? 4 * hx_poly_patch_bytes<MW, NW, sv>() : 0), stream, p);
        FASTSVC_HXP(2) FASTSVC_HXP(4) FASTSVC_HXP(5)
#undef FASTSVC_HXP
        return hipErrorInvalidValue;
    } else {
        const int kind = aff ? EPI_AFF : p.r1x ? EPI_RANK1 : p.res ? EPI_RES : EPI_PLAIN;
        size_t est = 0;
        constexpr bool PAIRS = hx_pairs_epi<MW, NW, MODE_DIRECT>();
        if ((kind == EPI_AFF && hx_estage<MW, NW, MODE_DIRECT, EPI_AFF>()) || (kind == EPI_RES && hx_estage<MW, NW, MODE_DIRECT, EPI_RES>()))
            est = sizeof(float) * 4 * (size_t)(aff ? (p.res ? 3 : 2) : 1) * (PAIRS ? MW * (NW / 2) * 256 : MW * NW * EST_ITEM_FLOATS);
        if (MW == 2 && p.nch32 == 1 && !p.x2 &&
            ((kind == EPI_AFF && hx_est2<MW, NW, MODE_DIRECT, EPI_AFF>()) || (kind == EPI_RES && hx_est2<MW, NW, MODE_DIRECT, EPI_RES>())))
            est *= 2;                                                              // one K chunk: two slot sets (EST2, the next tile's operands)
        if (PAIRS) est += sizeof(float) * 4 * 16 * 36;                             // the waves' re-layout patches
        if (p.x2) {
            // second operand (ConvParams::x2): FiLM-affine epilogue, no residual tensor, rows a multiple of 4 long; the
            // stretch factors of the recipe's blocks per channel-tile count (C = 24: x5, one K chunk; C >= 48: x2 / x4)
            if (kind != EPI_AFF || p.res || p.x2_T * p.s2 != p.T || (p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0)))
                return hipErrorInvalidValue;
            if (w8 && p.s2 == 2 && W / p.s2 + 5 > 128) return hipErrorInvalidValue;   // (the second operand's 32 groups of 4 columns per tile)
#define FASTSVC_HXX(sv, stat) if (p.s2 == sv) return hx_launch_instance<&conv_hx_kernel<MW, NW, WM, WN, MODE_DIRECT, EPI_AFF, sv, stat>>(grid, smem + est, stream, p);
            if constexpr (MW * NW <= 12) {                                         // (larger tiles spill with this epilogue)
                if constexpr (MW == 2) { if (p.nch32 == 1) { FASTSVC_HXX(5, true) } }
                else { FASTSVC_HXX(2, false) FASTSVC_HXX(4, false) }
            }
#undef FASTSVC_HXX
            return hipErrorInvalidValue;
        }
#define FASTSVC_HX(k) if (kind == k) return hx_launch_kind<MW, NW, WM, WN, MODE_DIRECT, k, 1>(grid, smem + est, stream, p);
        FASTSVC_HX(EPI_PLAIN) FASTSVC_HX(EPI_RES) FASTSVC_HX(EPI_RANK1) FASTSVC_HX(EPI_AFF)
#undef FASTSVC_HX
        return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_hx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
    if (p.ntaps != 3 || (p.T & 3) || !p.whx || p.nch32 * HX_KC > 512) return hipErrorInvalidValue;   // (one input channel per thread in the set-up)
#define FASTSVC_HXS(md, mw, nw, wm, wn) \
    if (cfg.MW == mw && cfg.NW == nw && cfg.WM == wm && cfg.WN == wn) \
        return hx_launch_shape<mw, nw, wm, wn, md>(p, cfg.nsig, stream);
    if (p.mode == MODE_DIRECT) {
        if (p.dil < 1 || p.dil > 28) return hipErrorInvalidValue;
        // workgroup tiles of 128 or 192 output columns: with the halo (<= 2 x 28) the window stays within 256 rows
        // = one item per producer thread
        FASTSVC_HXS(MODE_DIRECT, 2, 2, 1, 4) FASTSVC_HXS(MODE_DIRECT, 2, 3, 1, 4)
        FASTSVC_HXS(MODE_DIRECT, 3, 2, 1, 4) FASTSVC_HXS(MODE_DIRECT, 3, 3, 1, 4)
        FASTSVC_HXS(MODE_DIRECT, 3, 4, 2, 2) FASTSVC_HXS(MODE_DIRECT, 3, 6, 2, 2)
        FASTSVC_HXS(MODE_DIRECT, 3, 8, 4, 1)
    } else if (p.mode == MODE_POLY) {
        if (p.dil != 1) return hipErrorInvalidValue;
        FASTSVC_HXS(MODE_POLY, 2, 2, 1, 4) FASTSVC_HXS(MODE_POLY, 2, 3, 1, 4)
        FASTSVC_HXS(MODE_POLY, 3, 2, 1, 4) FASTSVC_HXS(MODE_POLY, 3, 2, 2, 2)
    } else if (p.mode == MODE_CHAIN) {
        // every workgroup holds the WHOLE intermediate tensor slice: one workgroup row of channel groups
        if (p.dil < 1 || p.dil > 4 || p.dil2 < 1 || p.dil2 > 4 || !p.bias_mid || p.CMID != p.COUT) return hipErrorInvalidValue;
        if (p.ngroups > cfg.WM) return hipErrorInvalidValue;
        FASTSVC_HXS(MODE_CHAIN, 2, 2, 1, 4) FASTSVC_HXS(MODE_CHAIN, 2, 3, 1, 4)
        FASTSVC_HXS(MODE_CHAIN, 3, 2, 1, 4) FASTSVC_HXS(MODE_CHAIN, 3, 3, 1, 4)
        FASTSVC_HXS(MODE_CHAIN, 3, 4, 2, 2) FASTSVC_HXS(MODE_CHAIN, 3, 6, 2, 2)
    } else if (p.mode == MODE_UPHEAD) {
        // (every workgroup holds the whole C-channel intermediate tile, twice: one workgroup row of channel groups)
        if (p.dil != 1 || p.dil2 != 1 || !p.bias_mid || p.CMID != p.COUT || p.ngroups > cfg.WM) return hipErrorInvalidValue;
        FASTSVC_HXS(MODE_UPHEAD, 3, 4, 4, 1) FASTSVC_HXS(MODE_UPHEAD, 3, 2, 2, 2)
        FASTSVC_HXS(MODE_UPHEAD, 3, 2, 1, 4) FASTSVC_HXS(MODE_UPHEAD, 2, 2, 1, 4)
    } else if (p.mode == MODE_CHAIN1) {
        // (tiles of <= 192 columns: one staging item per thread, which the first conv's per-thread taps rely on)
        if (p.dil < 1 || p.dil > 4 || p.dil2 < 1 || p.dil2 > 4 || !p.bias_mid || p.CMID != p.COUT || p.CIN > 32 ||
            !p.in1_w || !p.in1_b || p.ngroups > cfg.WM) return hipErrorInvalidValue;
        FASTSVC_HXS(MODE_CHAIN1, 2, 2, 1, 4) FASTSVC_HXS(MODE_CHAIN1, 2, 3, 1, 4)
    } else if (p.mode == MODE_DEC2) {
        if (p.dil != 1 || !p.bias2 || !p.y2) return hipErrorInvalidValue;
        FASTSVC_HXS(MODE_DEC2, 3, 2, 1, 4) FASTSVC_HXS(MODE_DEC2, 3, 3, 1, 4) FASTSVC_HXS(MODE_DEC2, 3, 2, 2, 2)
        // every channel group of a C_out = 96 / 192 stage in ONE workgroup: the two staged windows (LeakyReLU'd and raw)
        // serve 2 / 4 times the matrix work, and the input is fetched once instead of once per channel group
        FASTSVC_HXS(MODE_DEC2, 3, 4, 2, 2) FASTSVC_HXS(MODE_DEC2, 3, 4, 4, 1)
    }
#undef FASTSVC_HXS
    return hipErrorInvalidValue;
}

#ifndef FASTSVC_ACT_BF16      // storage-independent host query: defined once
bool conv_hx_x2_ok(int MW, int nch32, int s2) { return MW == 2 ? (nch32 == 1 && s2 == 5) : MW == 3 ? (s2 == 2 || s2 == 4) : false; }
bool conv_hx_tail_ok(int mode, int MW, int epi_kind, int S) { return hx_tail_instance(MW, mode, epi_kind, S); }

bool conv_hx_shape(int mode, int MW, int NW, int WM, int WN) {
    if (mode == MODE_POLY)
        return (MW == 2 && WM == 1 && WN == 4 && (NW == 2 || NW == 3)) ||
               (MW == 3 && NW == 2 && ((WM == 1 && WN == 4) || (WM == 2 && WN == 2)));
    if (mode == MODE_DEC2)
        return MW == 3 && ((WM == 1 && WN == 4 && (NW == 2 || NW == 3)) || (WM == 2 && WN == 2 && (NW == 2 || NW == 4)) ||
                           (WM == 4 && WN == 1 && NW == 4));
    if (mode == MODE_CHAIN1) return MW == 2 && WM == 1 && WN == 4 && (NW == 2 || NW == 3);
    if (mode == MODE_UPHEAD)
        return (MW == 3 && ((NW == 4 && WM == 4 && WN == 1) || (NW == 2 && WM == 2 && WN == 2) || (NW == 2 && WM == 1 && WN == 4))) ||
               (MW == 2 && NW == 2 && WM == 1 && WN == 4);
    if (mode == MODE_CHAIN)
        return (WM == 1 && WN == 4 && (MW == 2 || MW == 3) && (NW == 2 || NW == 3)) ||
               (MW == 3 && WM == 2 && WN == 2 && (NW == 4 || NW == 6));
    if (mode != MODE_DIRECT) return false;
    if (MW == 2) return WM == 1 && WN == 4 && (NW == 2 || NW == 3);
    if (MW != 3) return false;
    if (WM == 1 && WN == 4) return NW == 2 || NW == 3;
    if (WM == 2 && WN == 2) return NW == 4 || NW == 6;
    return WM == 4 && WN == 1 && NW == 8;
}
#endif

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
