// fastsvc_hx.hip - gfx950 (CDNA4 / MI355X) half-precision-MFMA convolution kernels of the FastSVC
// generator forward: the same k=3 dilated "same" convolutions, fused prologues and epilogues as
// fastsvc_kernels.hip (reference: Conv1d1x3 / Conv2d1x3, harana/layers/upsample.py:76-83,99-106, used by
// harana/models/fastsvc.py:56-75,164-178,209-218), with the implicit GEMM on the 16x16x32 matrix
// instruction instead of the f32-input one (1/16 of its rate):
//
//   float32 storage (namespace fastsvc):  SPLIT-HALF products, fp32-class results.  Every operand is split
//       exactly into two binary16 pieces, x = xh + xl with xh = f16(x), xl = f16(x - xh) (22 significand
//       bits; the pieces' products are exact in the fp32 accumulator), and
//           x * w  ~=  xh*wh + xh*wl + xl*wh          (the dropped xl*wl term is < 2^-22 |x w|)
//       costs three v_mfma_f32_16x16x32_f16 per 32 input channels = 3/16 of the f32-input MFMA time.
//       End to end the generator output moves by 4e-6 against the fp32 path (oracle simulation and
//       tests/test_parity_gpu.py), i.e. it stays at the reference's own fp32 noise floor (1e-5).
//   bfloat16 storage (namespace fastsvc::bf16, -DFASTSVC_ACT_BF16):  ONE v_mfma_f32_16x16x32_bf16 product
//       of the bf16-rounded operands - BASELINE config 3's "bf16 generator forward".
//
// Data movement (what bounds these kernels: with the matrix work this cheap every layer is HBM-side):
//   * workgroup = 4 consumer + 4 producer waves, one barrier per unit = (time tile, 32-channel K chunk),
//     double-buffered LDS tile, p.tpw consecutive tiles per workgroup - the pipeline of conv_mfma_ws_kernel;
//   * LDS tile = TIME-major rows of 32 channels (64 B of f16 / bf16): a lane's MFMA A-fragment - 8
//     consecutive input channels at one time step - is ONE ds_read_b128, and a conv tap is a ROW offset
//     (always 16-byte aligned whatever the dilation).  16-byte slot s of row r sits at slot s ^ ((r >> 1) & 2):
//     conflict-free for the four 16-lane service groups of ds_read_b128 at any row offset;
//   * producers load float4 (4 time steps) of 8 channels per thread - 256 B contiguous per channel row and
//     wave instruction -, apply InstanceNorm / LeakyReLU, split, transpose 4 x 8 in registers and write
//     4 (+4) ds_write_b128;
//   * weights: pre-split fragments in MFMA operand order ([chunk][tap][16-channel tile][hi,lo][lane][8]),
//     streamed from L2 into a unit-deep register ring with buffer_load_dwordx4 (resident when C_in <= 32);
//   * epilogues: shared with the f32 kernels (fastsvc_device.inc).
// Rows must be a multiple of 4 long (float4 everywhere); run_conv falls back to conv_mfma_ws_kernel otherwise.
#include "fastsvc_kernels.h"

namespace fastsvc {
#ifdef FASTSVC_ACT_BF16
namespace bf16 {
#endif

#include "fastsvc_device.inc"

#ifdef FASTSVC_ACT_BF16
typedef __bf16 hx_t;
constexpr int HX_NP = 1;                 // operand pieces: bf16 product of the rounded operands
#else
typedef _Float16 hx_t;
constexpr int HX_NP = 2;                 // hi + lo binary16 pieces, three products
#endif
typedef hx_t hx8 __attribute__((ext_vector_type(8)));
typedef hx_t hx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 hx_mfma(hx8 a, hx8 b, f32x4 c) {
#ifdef FASTSVC_ACT_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

constexpr int HX_KC = 32;                // input channels per K chunk = one MFMA K step per tap
constexpr int HX_ROW = 64;               // bytes of one LDS tile row (32 channels)
constexpr int HX_FRAG = 1024;            // bytes of one packed weight fragment (64 lanes x 8 halves)

// LDS byte offset of 16-byte slot `oct` of tile row `row`
__device__ __forceinline__ int hx_lds_off(int row, int oct) { return row * HX_ROW + ((oct ^ ((row >> 1) & 2)) << 4); }

// Unit-deep weight ring: NSLOT fragments = every fragment of one (tile, chunk) unit of this wave's channel
// group, statically indexed; slot s is re-requested with the NEXT unit's fragment s right after its last use.
template <int NSLOT>
struct HxWeightStream {
    u32x4 wr[NSLOT];
    __amdgpu_buffer_rsrc_t rsrc;
    int voff, next, total;
    __device__ __forceinline__ void init(const unsigned char* group_base, int nunits, int lane) {
        total = nunits * NSLOT * HX_FRAG;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(group_base), 0, total, 0x00020000);
        voff = lane * 16;
        #pragma unroll
        for (int s = 0; s < NSLOT; ++s) wr[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s * HX_FRAG, 0);
        next = NSLOT * HX_FRAG;
        if (next >= total) next = 0;
    }
    __device__ __forceinline__ void request(int s) { wr[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, next + s * HX_FRAG, 0); }
    __device__ __forceinline__ void advance() { next += NSLOT * HX_FRAG; if (next >= total) next = 0; }
};

struct HxFrag { hx8 p[HX_NP]; };

__device__ __forceinline__ HxFrag hx_read(const unsigned char* tile, int off, int lo_off) {
    HxFrag f;
    f.p[0] = *reinterpret_cast<const hx8*>(tile + off);
    if constexpr (HX_NP == 2) f.p[1] = *reinterpret_cast<const hx8*>(tile + lo_off + off);
    return f;
}

// acc += a (.) w for one 16x16 tile and 32 input channels: hh + hl + lh (split) or one product (bf16).
// Callers interleave independent accumulators between the products of one (the loops below run the product
// index outermost) so that no MFMA waits for the previous one's result.
template <int PROD>
__device__ __forceinline__ f32x4 hx_prod(const HxFrag& a, const u32x4 (&w)[HX_NP], f32x4 acc) {
    if constexpr (HX_NP == 1) return hx_mfma(a.p[0], __builtin_bit_cast(hx8, w[0]), acc);
    else if constexpr (PROD == 0) return hx_mfma(a.p[0], __builtin_bit_cast(hx8, w[0]), acc);
    else if constexpr (PROD == 1) return hx_mfma(a.p[0], __builtin_bit_cast(hx8, w[1]), acc);
    else return hx_mfma(a.p[1], __builtin_bit_cast(hx8, w[0]), acc);
}
constexpr int HX_NPROD = HX_NP == 2 ? 3 : 1;

// DIRECT unit: acc[n][m] += sum_tap X[t + (tap-1) d] W[tap].  Steps run tap-major so that a tap's weight
// slots are re-requested (for the next unit) as early as possible; the A fragments of step s+1 are read
// from LDS before the MFMAs of step s.  aoff[tap]: this lane's byte offset of (row of time tile 0, its octet).
template <int MW, int NW, bool RELOAD>
__device__ __forceinline__ void hx_unit_direct(f32x4 (&acc)[NW][MW], const unsigned char* tile, const int (&aoff)[3],
                                               int lo_off, HxWeightStream<3 * MW * HX_NP>& ws) {
    constexpr int NG = (MW >= 3 || NW == 1) ? 1 : 2;          // time tiles per step (>= 3 independent accumulators)
    constexpr int NSTEP = 3 * NW / NG;
    HxFrag a[2][NG];
    #pragma unroll
    for (int g = 0; g < NG; ++g) a[0][g] = hx_read(tile, aoff[0] + g * 16 * HX_ROW, lo_off);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int tap = s / (NW / NG), n0 = (s % (NW / NG)) * NG;
        if (s + 1 < NSTEP) {
            const int tap1 = (s + 1) / (NW / NG), n1 = ((s + 1) % (NW / NG)) * NG;
            #pragma unroll
            for (int g = 0; g < NG; ++g) a[(s + 1) & 1][g] = hx_read(tile, aoff[tap1] + (n1 + g) * 16 * HX_ROW, lo_off);
            __builtin_amdgcn_sched_barrier(0);                 // reads stay ahead of the MFMAs
        }
        #pragma unroll
        for (int pr = 0; pr < HX_NPROD; ++pr)
            #pragma unroll
            for (int g = 0; g < NG; ++g)
                #pragma unroll
                for (int m = 0; m < MW; ++m) {
                    u32x4 w[HX_NP];
                    #pragma unroll
                    for (int q = 0; q < HX_NP; ++q) w[q] = ws.wr[(tap * MW + m) * HX_NP + q];
                    if (pr == 0) acc[n0 + g][m] = hx_prod<0>(a[s & 1][g], w, acc[n0 + g][m]);
                    else if (pr == 1) acc[n0 + g][m] = hx_prod<1>(a[s & 1][g], w, acc[n0 + g][m]);
                    else acc[n0 + g][m] = hx_prod<2>(a[s & 1][g], w, acc[n0 + g][m]);
                }
        if (RELOAD && n0 + NG == NW) {                          // last use of this tap's fragments
            #pragma unroll
            for (int q = 0; q < MW * HX_NP; ++q) ws.request(tap * MW * HX_NP + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (RELOAD) ws.advance();
}

// convert / split 8 channel values of one time step into the tile's 16-byte slot(s)
__device__ __forceinline__ void hx_commit_slot(unsigned char* tile, int off, int lo_off, const float (&e)[8]) {
    hx8 h;
    #pragma unroll
    for (int c = 0; c < 8; ++c) h[c] = (hx_t)e[c];
    *reinterpret_cast<hx8*>(tile + off) = h;
    if constexpr (HX_NP == 2) {
        hx8 l;
        #pragma unroll
        for (int c = 0; c < 8; ++c) l[c] = (hx_t)(e[c] - (float)h[c]);
        *reinterpret_cast<hx8*>(tile + lo_off + off) = l;
    }
}

template <int MW, int NW, int EPI>
constexpr int hx_min_waves() {
    // 128 VGPRs (two workgroups per CU) when ring + accumulators + the producers' two register sets fit
    return (MW * NW <= 4) ? 4 : 2;
}

template <int MW, int NW, int WM, int WN, int EPI, bool WSTATIC>
__global__ __launch_bounds__(512, (hx_min_waves<MW, NW, EPI>()))
void conv_hx_direct_kernel(const ConvParams p0) {
    constexpr int NT = 16 * NW * WN;                                   // output columns per workgroup tile
    constexpr int NPROD_T = 256;                                       // producer threads
    constexpr int MAXW = NT + 56;                                      // halo <= 28 rows per side
    constexpr int ITEMS = (MAXW + NPROD_T - 1) / NPROD_T;              // (octet, 4 time steps) items per producer thread
    constexpr int NSLOT = 3 * MW * HX_NP;
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
    const int halo = p.dil;
    const int halo_al = (halo + 3) & ~3;
    const int W = NT + 2 * halo_al;                                    // tile rows
    const int nch = p.nch32;
    const int CINp = nch * HX_KC;
    const int flags = p.flags;
    const int ntx = (p.T + NT - 1) / NT;
    const int tile0 = blockIdx.x * p.tpw;
    const int ntiles = min(p.tpw, ntx - tile0);
    if (ntiles <= 0) return;
    const int nunits = ntiles * nch;

    double* sstat = reinterpret_cast<double*>(smem_raw);                               // [WM*MW*16][2]
    float2* ncoef = reinterpret_cast<float2*>(smem_raw + sizeof(double) * 2 * 16 * MW * WM);   // [CINp]
    unsigned char* tiles = reinterpret_cast<unsigned char*>(ncoef + CINp);             // [2][HX_NP][W rows][64 B]
    const int lo_off = W * HX_ROW;                                     // hi tile, then lo tile
    const int bufsz = HX_NP * W * HX_ROW;

    auto setup_shared = [&]() {
        if (flags & F_STATS) {
            for (int i = tid; i < 2 * 16 * MW * WM; i += 512) sstat[i] = 0.0;
        }
        if (flags & F_PRE_NORM) {
            // (u - mean) * rstd + p  ==  u * A + Bc  with A = rstd, Bc = p - mean * rstd   (fastsvc.py:134-139)
            const double inv_len = 1.0 / (double)p.x_T;
            for (int c = tid; c < CINp; c += 512) {
                float2 ab = make_float2(0.f, 0.f);
                if (c < p.CIN) {
                    const double q1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
                    const double q2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
                    const double mean = q1 * inv_len;
                    double var = q2 * inv_len - mean * mean;          // biased variance (InstanceNorm2d)
                    var = var > 0.0 ? var : 0.0;
                    const double rstd = 1.0 / sqrt(var + IN_EPS);
                    ab.x = (float)rstd;
                    ab.y = (float)((double)p.spk[(long)b * p.CIN + c] - mean * rstd);
                }
                ncoef[c] = ab;
            }
        }
        __syncthreads();
    };

    if (producer) {
        // ================================ PRODUCER WAVES ================================
        const int ptid = tid - 256;
        const __amdgpu_buffer_rsrc_t xr =
            act_rsrc(p.x, (long)sig * p.x_sig + (long)b * p.x_b, (long)p.CIN * p.ldx);
        int it_oct[ITEMS], it_q[ITEMS];
        #pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int idx = i * NPROD_T + ptid;
            it_oct[i] = idx & 3;
            it_q[i] = (idx >> 2) < (W >> 2) ? (idx >> 2) : -1;          // quad of rows 4q .. 4q+3; -1: no item
        }
        // unconditional loads of unit `un`: 8 channels x 4 time steps per item; whatever lies outside the
        // tensor or the utterance reads as 0 through the descriptor (offset pushed out of range)
        auto pload = [&](int un, f32x4 (&px)[ITEMS][8], unsigned& okmask) {
            const int tl = un / nch;
            const int ch = un - tl * nch;
            const int t_start = (tile0 + tl) * NT - halo_al;
            const int soff = ch * HX_KC * p.ldx * 4;
            const int rows_left = p.CIN - ch * HX_KC;
            okmask = 0;
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int t = t_start + 4 * it_q[i];
                const bool tok = it_q[i] >= 0 && (unsigned)t < (unsigned)p.T;
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int r = it_oct[i] * 8 + c;
                    const bool ok = tok && r < rows_left;
                    okmask |= (ok ? 1u : 0u) << (i * 8 + c);
                    px[i][c] = act_load4(xr, ok ? (r * p.ldx + t) * 4 : OOB_OFF, soff);
                }
            }
        };
        // prologue transform (InstanceNorm-apply + speaker bias as one FMA, LeakyReLU), split, transpose, LDS write
        auto pcommit = [&](int un, const f32x4 (&px)[ITEMS][8], unsigned okmask, unsigned char* tile) {
            const int ch = un % nch;
            #pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                if (it_q[i] < 0) continue;
                f32x4 v[8];
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    v[c] = f32x4{0.f, 0.f, 0.f, 0.f};                  // zero "same" padding / channel padding
                    if (okmask & (1u << (i * 8 + c))) {
                        v[c] = px[i][c];
                        if (flags & F_PRE_NORM) {
                            const float2 ab = ncoef[ch * HX_KC + it_oct[i] * 8 + c];
                            v[c] = v[c] * ab.x + ab.y;
                        }
                        if (flags & F_PRE_LRELU) {
                            v[c].x = lrelu(v[c].x); v[c].y = lrelu(v[c].y); v[c].z = lrelu(v[c].z); v[c].w = lrelu(v[c].w);
                        }
                    }
                }
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // odd quads write their rows in the order 1 0 3 2: the 8 lanes of a ds_write_b128 service
                    // group (two quads) then cover both 64-byte halves of the 32 write banks
                    const int jj = j ^ (it_q[i] & 1);
                    float e[8];
                    #pragma unroll
                    for (int c = 0; c < 8; ++c) e[c] = jj == 0 ? v[c].x : jj == 1 ? v[c].y : jj == 2 ? v[c].z : v[c].w;
                    hx_commit_slot(tile, hx_lds_off(4 * it_q[i] + jj, it_oct[i]), lo_off, e);
                }
            }
        };

        f32x4 pa[ITEMS][8], pb[ITEMS][8];
        unsigned oka = 0, okb = 0;
        pload(0, pa, oka);
        if (nunits > 1) pload(1, pb, okb);
        setup_shared();
        pcommit(0, pa, oka, tiles);
        __syncthreads();                               // unit 0 staged
        for (int u = 0; u < nunits; u += 2) {
            if (u + 1 < nunits) {
                if (u + 2 < nunits) pload(u + 2, pa, oka);
                pcommit(u + 1, pb, okb, tiles + bufsz);
            }
            __syncthreads();                           // end of unit u
            if (u + 1 >= nunits) break;
            if (u + 2 < nunits) {
                if (u + 3 < nunits) pload(u + 3, pb, okb);
                pcommit(u + 2, pa, oka, tiles);
            }
            __syncthreads();                           // end of unit u+1
        }
    } else {
        // ================================ CONSUMER WAVES ================================
        f32x4 acc[NW][MW];
        float s1[MW], s2[MW];
        HxWeightStream<NSLOT> wst;
        wst.init(reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                     (long)(active ? mg : 0) * nch * NSLOT * HX_FRAG, nch, lane);
        int aoff[3];
        #pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            aoff[tap] = hx_lds_off((halo_al - halo) + tap * halo + wave_n * (NW * 16) + (lane & 15), lane >> 4);
        EpiRsrc R;
        {
            const long ct = (long)p.COUT * p.ldy;
            const float* nul = p.bias;
            R.y = act_rsrc(p.y ? p.y : nul, p.y ? (long)sig * p.y_sig + (long)b * p.y_b : 0, p.y ? ct : 0);
            const bool has_y2 = (flags & F_AFF_OUT) != 0;
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
            if constexpr (EPI == EPI_RANK1) {
                k_r1w[m] = cok ? p.r1w[(long)sig * p.r1_sig + co] : 0.f;
                k_r1b[m] = cok ? p.r1b[(long)sig * p.r1_sig + co] : 0.f;
            }
        }
        const EpiConst<MW, true> K{k_bias, k_bias2, k_r1w, k_r1b};
        setup_shared();
        __syncthreads();                               // unit 0 staged
        int u = 0;
        for (int tl = 0; tl < ntiles; ++tl) {
            #pragma unroll
            for (int n = 0; n < NW; ++n)
                #pragma unroll
                for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < nch; ++ch, ++u) {
                if (active) hx_unit_direct<MW, NW, !WSTATIC>(acc, tiles + (u & 1) * bufsz, aoff, lo_off, wst);
                if (ch + 1 == nch) {
                    #pragma unroll
                    for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
                    ws_epilogue_kind<MW, NW, EPI, false, 0>(p, R, acc, s1, s2, sig, mg,
                                                            (tile0 + tl) * NT + wave_n * (NW * 16), active, lane, K);
                    if (flags & F_STATS) {
                        #pragma unroll
                        for (int m = 0; m < MW; ++m) {
                            float a1 = s1[m], a2 = s2[m];
                            a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
                            a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
                            if (active && lane < 16) {
                                const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                                atomicAdd(&sstat[slot + 0], (double)a1);
                                atomicAdd(&sstat[slot + 1], (double)a2);
                            }
                        }
                    }
                }
                __syncthreads();                       // end of unit u
            }
        }
    }
    if (flags & F_STATS) {                             // one f64 global atomic per channel per workgroup
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += 512) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}

template <auto KERNEL>
static hipError_t hx_launch_instance(dim3 grid, size_t smem, hipStream_t stream, const ConvParams& p) {
    if (smem > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(KERNEL, grid, dim3(512), smem, stream, p);
    return hipGetLastError();
}

template <int MW, int NW, int WM, int WN>
static hipError_t hx_launch_direct(const ConvParams& p, int nsig, hipStream_t stream) {
    constexpr int NT = 16 * NW * WN;
    const int ntx = (p.T + NT - 1) / NT;
    const int tpw = p.tpw > 0 ? p.tpw : 1;
    dim3 grid((ntx + tpw - 1) / tpw, (p.ngroups + WM - 1) / WM, nsig * p.B);
    const int halo_al = (p.dil + 3) & ~3;
    const int W = NT + 2 * halo_al;
    const int nbuf = (p.nch32 > 1 || tpw > 1) ? 2 : 1;
    const size_t smem = sizeof(double) * 2 * 16 * MW * WM + sizeof(float) * 2 * (size_t)p.nch32 * HX_KC +
                        (size_t)nbuf * HX_NP * W * HX_ROW;
    const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
    const int kind = aff ? EPI_AFF : p.r1x ? EPI_RANK1 : p.res ? EPI_RES : EPI_PLAIN;
    const bool wstatic = p.nch32 == 1;                 // the ring holds the whole layer: nothing to re-request
#define FASTSVC_HX(k) \
    if (kind == k) return wstatic ? hx_launch_instance<&conv_hx_direct_kernel<MW, NW, WM, WN, k, true>>(grid, smem, stream, p) \
                                  : hx_launch_instance<&conv_hx_direct_kernel<MW, NW, WM, WN, k, false>>(grid, smem, stream, p);
    FASTSVC_HX(EPI_PLAIN) FASTSVC_HX(EPI_RES) FASTSVC_HX(EPI_RANK1) FASTSVC_HX(EPI_AFF)
#undef FASTSVC_HX
    return hipErrorInvalidValue;
}

hipError_t launch_conv_hx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
    if (p.mode != MODE_DIRECT || p.ntaps != 3 || p.dil < 1 || p.dil > 28 || (p.T & 3) || !p.whx) return hipErrorInvalidValue;
#define FASTSVC_HXS(mw, nw, wm, wn) \
    if (cfg.MW == mw && cfg.NW == nw && cfg.WM == wm && cfg.WN == wn) return hx_launch_direct<mw, nw, wm, wn>(p, cfg.nsig, stream);
    FASTSVC_HXS(2, 2, 1, 4) FASTSVC_HXS(2, 4, 1, 4)
    FASTSVC_HXS(3, 2, 1, 4) FASTSVC_HXS(3, 4, 1, 4)
    FASTSVC_HXS(3, 2, 2, 2) FASTSVC_HXS(3, 4, 2, 2)
    FASTSVC_HXS(3, 4, 4, 1)
#undef FASTSVC_HXS
    return hipErrorInvalidValue;
}

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
