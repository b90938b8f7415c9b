// fastsvc_wx.hip - the WIDE-layer (C >= 96) implicit-GEMM convolution kernel of the FastSVC generator forward for gfx950
// (CDNA4 / MI355X): the same k=3 dilated "same" convolutions, prologues and epilogues as fastsvc_hx.hip (reference:
// Conv1d1x3 / Conv2d1x3, harana/layers/upsample.py:76-83,99-106, used by harana/models/fastsvc.py:94-112,164-178,209-232),
// laid out for the layers whose roofline is the MATRIX pipe rather than HBM.
//
// What differs from conv_hx_kernel (4 consumer + 4 staging waves, weights in a per-wave register ring):
//   * EVERY wave multiplies.  A workgroup is 8 waves = WM channel groups (of 16 MW = 48 output channels) x WN time slices
//     (of 16 NW columns); two waves share each SIMD's matrix pipe, so one wave's LDS round trips, staging work and
//     epilogue sit under the other's products.  conv_hx's consumer wave is alone on its SIMD's pipe: its serial
//     fragment-read -> product chain kept the pipe 21-32 % busy on these layers (profiles/r5c_cfg3_bfloat16_sq_counters.csv).
//   * Weights go through LDS ONCE per workgroup: a unit's fragments of all WM groups ([group][tap][tile][piece] KB-sized,
//     already in MFMA operand order in the blob) arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, no registers, no VALU)
//     one unit ahead, double-buffered; every wave reads its group's with conflict-free lane-linear ds_read_b128.
//     conv_hx streams them from L2 into registers per wave: WN times the L2 traffic, and the ring's depth ties the
//     products to the L2 latency.
//   * The workgroup tile is 256-384 columns (conv_hx: 128), so a staged window and a unit's weights serve 2-3 x the
//     products.
//   * Activation windows are staged by the first waves of the workgroup - as many as the window has (octet of 8 channels,
//     quad of 4 rows) items - with the SAME item code as conv_hx's staging waves (prologue FMA = InstanceNorm + speaker bias,
//     LeakyReLU, convert, 4 x 8 transpose in registers, ds_write_b128 into the swizzled time-major tile).  A unit's window
//     is requested at the end of the previous unit and committed at the end of its own: one register set, a whole unit
//     (>= 2k cycles of products) of lead.
//   * Epilogue of tile t runs AFTER the barrier that ends its last unit and after the next unit's weight DMA is issued.
// bfloat16 storage only for now (float32 storage: hi + lo planes and fragments need twice the LDS).
#include "fastsvc_kernels.h"

namespace fastsvc {
#ifdef FASTSVC_ACT_BF16
namespace bf16 {
#endif

#include "fastsvc_device.inc"
#include "fastsvc_hx_common.inc"

constexpr int WX_NWAVES = 8;
constexpr int WX_NT = WX_NWAVES * 64;

// One unit = (32-channel K chunk) x 3 taps on this wave's NW time tiles and MW channel tiles.
//   tile: the staged window (time-major rows of 32 channels), aoff[tap]: this lane's fragment offset of time tile 0
//   wl:   this wave's group's fragments of the unit in LDS, [tap][m][piece] x 1 KB, lane-linear
// A fragments are read PF steps ahead, a tap's weight fragments while the previous tap's last steps run.
// on_step(s): called in front of step s's products (the waves that request the next unit's weights issue one LDS-DMA piece
// per step there: issued in one burst at the top of the unit, twelve pieces cost the wave ~1.8k cycles before its first product)
template <int MW, int NW, class F>
__device__ __forceinline__ void wx_unit(f32x4 (&acc)[NW][MW], const unsigned char* tile, const int (&aoff)[3], int lo_off,
                                        const unsigned char* wl, int lane, F&& on_step) {
    constexpr int NSTEP = 3 * NW;
#ifndef WX_PF
#define WX_PF 2
#endif
    constexpr int PF = HX_NP == 1 ? WX_PF : 1;
    constexpr int NF = MW * HX_NP;
    HxFrag a[PF + 1];
    u32x4 w[2][NF];
    #pragma unroll
    for (int q = 0; q < NF; ++q) w[0][q] = *reinterpret_cast<const u32x4*>(wl + q * HX_FRAG + lane * 16);
    #pragma unroll
    for (int q = 0; q < PF && q < NSTEP; ++q) a[q] = hx_read(tile, aoff[q / NW] + (q % NW) * 16 * HX_ROW, lo_off);
    #pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int tap = s / NW, n = s % NW;
        if (s + PF < NSTEP) a[(s + PF) % (PF + 1)] = hx_read(tile, aoff[(s + PF) / NW] + ((s + PF) % NW) * 16 * HX_ROW, lo_off);
        if (n == (NW > 2 ? NW - 3 : 0) && tap + 1 < 3) {
            #pragma unroll
            for (int q = 0; q < NF; ++q)
                w[(tap + 1) & 1][q] = *reinterpret_cast<const u32x4*>(wl + ((tap + 1) * NF + q) * HX_FRAG + lane * 16);
        }
        on_step(s);
        __builtin_amdgcn_sched_barrier(0);                     // reads stay ahead of the products
        hx_step<MW>(acc[n], a[s % (PF + 1)], &w[tap & 1][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int V> struct WxSet { static constexpr int value = V; };      // a compile-time register-set index

// MW x NW: 16-channel x 16-column result tiles per wave; WM x WN = 8 waves; EPI: epilogue kind (fastsvc_device.inc)
template <int MW, int NW, int WM, int WN, int EPI>
__global__ __launch_bounds__(WX_NT, 2)
void conv_wx_kernel(const ConvParams p0) {
    static_assert(WM * WN == WX_NWAVES, "eight waves");
    constexpr int NT = 16 * NW * WN;                                   // columns per workgroup tile
    constexpr int MAXW = NT + 56;                                      // window rows: halo <= 28 per side
    constexpr int ITEMS = (MAXW + WX_NT - 1) / WX_NT;                  // (octet, quad of rows) items per thread
    constexpr int NSLOT = 3 * MW * HX_NP;                              // fragments of one (group, chunk)
    constexpr int WUNIT = WM * NSLOT * HX_FRAG;                        // bytes of one unit's weights (all groups)
    constexpr bool TRACKS = AMAX_TRACK && (EPI == EPI_RES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN;
    const int wave_n = wave - wave_m * WN;
    const int z = blockIdx.z;
    const int sig = z / p0.B;
    const int b = z - sig * p0.B;
    ConvParams p = p0;
    if (p0.lens) {                                                     // ragged batch: this utterance's own lengths
        const int frames = p0.lens[b];
        p.T = frames * p0.len_mul;
        p.x_T = frames * p0.xlen_mul;
    }
    const int mg = blockIdx.y * WM + wave_m;                           // (ngroups is a multiple of WM: every wave is active)
    const int halo = p.dil;
    const int halo_al = (halo + 3) & ~3;
    const int W = NT + 2 * halo_al;                                    // window rows
    const int nch = p.nch32;
    const int CINp = nch * HX_KC;
    const int flags = p.flags;
    const int ntx = (p.T + NT - 1) / NT;
    const int tile0 = blockIdx.x * p.tpw;
    const int ntiles = min(p.tpw, ntx - tile0);
    if (ntiles <= 0) return;
    const int nunits = ntiles * nch;

    double* sstat = reinterpret_cast<double*>(smem_raw);                               // [WM*MW*16][2]
    float2* ncoef = reinterpret_cast<float2*>(smem_raw + sizeof(double) * 2 * 16 * MW * WM);   // [CINp + 8]: the last 8 are (0, 0)
    unsigned char* tiles = reinterpret_cast<unsigned char*>(ncoef + CINp + 8);         // [2][HX_NP][W + 4 rows][64 B]
    const int lo_off = (W + 4) * HX_ROW;
    const int bufsz = HX_NP * (W + 4) * HX_ROW;
    unsigned char* wbuf = tiles + 2 * bufsz;                                           // [2][WM][tap][m][piece] fragments
    float* Xw = reinterpret_cast<float*>(wbuf + 2 * WUNIT) + wave * (16 * 36);         // bfloat16 pair epilogue: this wave's patch
    (void)Xw;
    __shared__ unsigned s_amax, s_cnt;
    __shared__ float s_inv[HX_NP == 2 ? 16 * MW * WM : 4];

#ifdef FASTSVC_TIMELINE
    // diagnostic build (tools/timeline.py): lane 0 of every wave stamps s_memtime at its phase boundaries
    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long* tlw = (p.tl && wg_lin < p.tl_wgs) ? p.tl + ((long)wg_lin * 8 + wave) * 64 : nullptr;
    int tli = 0;
    auto stamp = [&](int tag) __attribute__((always_inline)) {
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
    // ---- weight units by LDS-DMA: unit un -> buffer un & 1.  Piece j of the unit = fragment j % NSLOT of group j / NSLOT;
    // the pieces are dealt to the waves from the LAST wave down (the first waves stage the windows).
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                                (long)(blockIdx.y * WM) * nch * NSLOT * HX_FRAG;
    // one descriptor over the WM groups' fragments (group stride nch * NSLOT KB)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wsrc), 0, WM * nch * NSLOT * HX_FRAG, 0x00020000);
    const unsigned wbuf_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wbuf;
    constexpr int NPIECE = WM * NSLOT;
    // the waves WITHOUT a staging item issue them (a staging wave's window requests must stay in flight across the unit's
    // end, and a wait for weight pieces would include every older request of the wave)
    const int nstage = min(WX_NWAVES - 1, (W + 63) >> 6);              // staging waves: 0 .. nstage - 1
    const bool stager = wave < nstage;                                 // (wave-uniform)
    auto dma_piece = [&](int un, int ch, int j) __attribute__((always_inline)) {
        const int g = j / NSLOT, f = j - g * NSLOT;
        lds_dma16(wr, wbuf_lds + (un & 1) * WUNIT + j * HX_FRAG, lane * 16, ((g * nch + ch) * NSLOT + f) * HX_FRAG);
    };
    auto dma_unit = [&](int un) __attribute__((always_inline)) {       // (the first unit's: in one go)
        for (int j = wave - nstage; j < NPIECE; j += WX_NWAVES - nstage) dma_piece(un, un % nch, j);   // (wave-uniform; never entered by a staging wave)
    };

    // ---- window staging: item = (octet of 8 channels, quad of 4 rows); waves without an item skip the code ----
    const __amdgpu_buffer_rsrc_t xr = act_rsrc(p.x, (long)sig * p.x_sig + (long)b * p.x_b, (long)p.CIN * p.ldx);
    int it_oct[ITEMS], it_q[ITEMS];
    bool it_in[ITEMS];
    #pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = i * WX_NT + tid;
        it_oct[i] = idx & 3;
        it_in[i] = (idx >> 2) < (W >> 2);
        it_q[i] = it_in[i] ? (idx >> 2) : (W >> 2);                    // parked in the 4 spare rows behind the tile
    }
    const float slope = (flags & F_PRE_LRELU) ? LRELU_SLOPE : 1.0f;    // max(v, slope * v): identity for 1
    int l_tl = 0, l_ch = 0, c_ch = 0;
    // window w is requested into register set w & 1 two units before its commit (one unit of products is shorter than a
    // memory round trip under load: with one set, requested at the top of the unit before, every unit waited for its window)
    act4_t pxs[2][ITEMS][8];
    unsigned tokmasks[2] = {0u, 0u};
    auto pload = [&](auto SETC, int un) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETC)::value;
        act4_t (&px)[ITEMS][8] = pxs[SET];
        unsigned& tokmask = tokmasks[SET];
        const int tl = l_tl, ch = l_ch;
        { const bool wrap = l_ch + 1 == nch; l_ch = wrap ? 0 : l_ch + 1; l_tl += wrap ? 1 : 0; }
        const int t_start = (tile0 + tl) * NT - halo_al;
        const int soff = ch * HX_KC * p.ldx * 4;
        const int rows_left = p.CIN - ch * HX_KC;
        tokmask = 0;
        #pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int t = t_start + 4 * it_q[i];
            // (C_in is a multiple of 8: an octet lies inside the tensor or behind it as a whole)
            const bool tok = it_in[i] & ((unsigned)t < (unsigned)p.T) & (un < nunits) & (it_oct[i] * 8 < rows_left);
            tokmask |= (tok ? 1u : 0u) << i;
            // ONE per-lane offset per item; the channel of the octet rides in the scalar offset (eight per-lane offsets,
            // kept across the loop, were what pushed the 3 x 8 instance past its 256 registers)
            const int voff = tok ? (it_oct[i] * 8 * p.ldx + t) * 4 : OOB_OFF;
            #pragma unroll
            for (int c = 0; c < 8; ++c) px[i][c] = act_load4_raw(xr, voff, soff + c * p.ldx * 4);
        }
    };
    auto pcommit = [&](auto SETC, unsigned char* tile) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETC)::value;
        const act4_t (&px)[ITEMS][8] = pxs[SET];
        const unsigned tokmask = tokmasks[SET];
        const int ch = c_ch;
        c_ch = c_ch + 1 == nch ? 0 : c_ch + 1;
        #pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            // rows outside the utterance are the conv's zero padding AFTER the prologue: their loads returned 0, so only the
            // additive term has to go - such an item reads the (0, 0) coefficients behind the table
            const bool tok = ((tokmask >> i) & 1u) != 0;
            const f32x4* cf = reinterpret_cast<const f32x4*>(ncoef + (tok ? ch * HX_KC + it_oct[i] * 8 : CINp));
            const f32x4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
            const float A[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
            const float Bc[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
            // one time step at a time, unpacked where it is used: the 32 converted values never exist together (the unit's
            // 96 accumulators and both register sets are live here)
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                float e[8];
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    act4_t w = px[i][c];
                    asm volatile("" : "+v"(w));                // (keeps hipcc from unpacking all four steps up front)
                    e[c] = act_unpack4(w)[j] * A[c] + Bc[c];
                    e[c] = fmaxf(e[c], e[c] * slope);
                }
                hx_commit_slot(tile, hx_lds_off(4 * it_q[i] + j, it_oct[i]), lo_off, e);
            }
        }
    };

    // ---- epilogue descriptors and per-lane constants ----
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
        R.r1x = make_rsrc(nul, 0);
    }
    float k_bias[MW];
    #pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int cot = (mg * MW + m) * 16 + (lane & 15);
        k_bias[m] = cot < p.COUT ? p.bias[(long)sig * p.bias_sig + cot] : 0.f;
    }
    // (second bias / rank-1 constants: kinds this kernel does not have)
    const EpiConst<MW, true, HX_NP == 2> K{k_bias, k_bias, k_bias, k_bias, s_inv + wave_m * (MW * 16) + (lane & 15), 16 * MW * WM};

    // ---- set-up: first window and first weights in flight, then the tables ----
    typedef WxSet<0> Set0;
    typedef WxSet<1> Set1;
    // Workgroups dispatched together would run their tiles in lockstep: every CU multiplies at the same time (HBM idle) and
    // every CU stores its tile at the same time (the epilogues of film.2.heads at 64 x 12000: 25 MB per tile period, 5.8 us of
    // HBM time during which no CU multiplied - 30 % of the launch).  Workgroup i of an XCD (dispatch order: i / 8) starts
    // (i % 8) / 8 of a tile late, so that at any time an eighth of the chip stores.
    {
        const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int late = ((wg >> 3) & 7) * nch * p.stagger;    // (p.stagger: an eighth of a unit's time in 64-cycle steps)
        for (int k = 0; k < late; ++k) __builtin_amdgcn_s_sleep(1);
    }
    if (stager) { pload(Set0{}, 0); pload(Set1{}, 1); }
    dma_unit(0);
    {
        // prologue coefficients of every input channel, applied by the staging code as ONE FMA u * A + Bc:
        // InstanceNorm + speaker bias (u - mean) * rstd + p  ->  A = rstd, Bc = p - mean * rstd (fastsvc.py:134-139);
        // no norm: (1, 0); channel padding: (0, 0)
        const int c = tid;
        double q1 = 0.0, q2 = 0.0;
        float pc = 0.f;
        if ((flags & F_PRE_NORM) && c < p.CIN) {
            q1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
            q2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
            pc = p.spk[(long)b * p.CIN + c];
        }
        if constexpr (TRACKS) { if (tid == 0) { s_amax = 0u; s_cnt = 0u; } }
        if (flags & F_STATS) {
            for (int i = tid; i < 2 * 16 * MW * WM; i += WX_NT) sstat[i] = 0.0;
        }
        if (c < CINp) {
            float2 ab = make_float2(0.f, 0.f);
            if (c < p.CIN) {
                ab = make_float2(1.f, 0.f);
                if (flags & F_PRE_NORM) {
                    const double inv_len = 1.0 / (double)p.x_T;
                    const double mean = q1 * inv_len;
                    double var = q2 * inv_len - mean * mean;      // biased variance (InstanceNorm2d)
                    var = var > 0.0 ? var : 0.0;
                    const double rstd = 1.0 / sqrt(var + IN_EPS);
                    ab.x = (float)rstd;
                    ab.y = (float)((double)pc - mean * rstd);
                }
            }
            ncoef[c] = ab;
        }
        if (c < 8) ncoef[CINp + c] = make_float2(0.f, 0.f);       // what an item outside the utterance is staged with
        __syncthreads();
    }
    if (stager) { pcommit(Set0{}, tiles); pload(Set0{}, 2); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // unit 0's weights have landed
    __syncthreads();

    int aoff[3];
    #pragma unroll
    for (int tap = 0; tap < 3; ++tap)
        aoff[tap] = hx_lds_off((halo_al - halo) + tap * halo + wave_n * (NW * 16) + (lane & 15), lane >> 4);
    f32x4 acc[NW][MW];
    float s1[MW], s2[MW];
    int ch = 0, tl = 0;
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        if (FASTSVC_DBG_ON(p, DBG_NO_EPILOGUE)) return;
        const int tcolw = (tile0 + tile) * NT + wave_n * (NW * 16);
        #pragma unroll
        for (int m = 0; m < MW; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
#ifdef FASTSVC_ACT_BF16
        if constexpr (NW % 2 == 0) hx_epilogue8<MW, NW / 2, EPI, false>(p, R, acc, s1, s2, sig, mg, tcolw, true, lane, K, nullptr, Xw);
        else
#endif
        ws_epilogue_kind<MW, NW, EPI, false, 0, -1>(p, R, acc, s1, s2, sig, mg, tcolw, true, lane, K, nullptr);
        if constexpr (TRACKS) { if (p.amax_out) amax_tile_flush(R); }
        if (flags & F_STATS) {
            #pragma unroll
            for (int m = 0; m < MW; ++m) {
                float a1 = row_xsum(s1[m]), a2 = row_xsum(s2[m]);
                if (lane < 16) {
                    const int slot = ((wave_m * MW + m) * 16 + lane) * 2;
                    atomicAdd(&sstat[slot + 0], (double)a1);
                    atomicAdd(&sstat[slot + 1], (double)a2);
                }
            }
        }
    };
    // One unit.  ROLE (compile time: each role runs its own copy of the loop, straight-line in what concerns its memory
    // requests - a branch between a window request and its commit makes hipcc count the requests of the path WITHOUT the
    // younger set and wait for both sets: the two-deep prefetch then runs one deep, see fastsvc_hx.hip):
    //   0  no staging item: requests the next unit's weights (LDS-DMA, in one go: a piece per product step was measured slower)
    //   1  stages EARLY (waves 0-3, one per SIMD): first commits the NEXT unit's window and requests the one two units on - the
    //      partner on the SIMD multiplies meanwhile, and this wave multiplies while the partner commits or waits at the barrier
    //      (committing after the products in every wave left the matrix pipe idle for the length of the commit, every unit)
    //   2  stages LATE (waves 4..: windows of more than 256 rows): products first - both waves of a SIMD committing first was
    //      what paced the 264-row windows (3.9k cycles per unit on SIMD 0 against 2.4k of products)
    // Past the last unit the requests fetch nothing (offsets out of range) and the commit stages zeros into the buffer nobody
    // reads.  SETC: the register set that holds window u + 1.
    auto unit = [&](auto ROLE, auto SETC, int u) __attribute__((always_inline)) {
        constexpr int ROLE_ID = decltype(ROLE)::value;
        if constexpr (ROLE_ID == 1) {
            if (!(FASTSVC_DBG_ON(p, DBG_NO_COMMIT))) { pcommit(SETC, tiles + ((u + 1) & 1) * bufsz); pload(SETC, u + 3); }
            asm volatile("" ::: "memory");                     // (the requests go out HERE: hipcc otherwise sinks them below the products)
            stamp(5);
        }
        if constexpr (ROLE_ID == 0) {
            if (u + 1 < nunits && !(FASTSVC_DBG_ON(p, DBG_NO_WEIGHTS))) dma_unit(u + 1);   // (buffer (u + 1) & 1 was last read in unit u - 1: behind a barrier)
            stamp(9);
        }
        if (ch == 0) {
            if (tl > 0) { epilogue(tl - 1); stamp(8); }        // the tile that ended with the last barrier
            #pragma unroll
            for (int n = 0; n < NW; ++n)
                #pragma unroll
                for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (u < nunits && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA)))
            wx_unit<MW, NW>(acc, tiles + (u & 1) * bufsz, aoff, lo_off, wbuf + (u & 1) * WUNIT + wave_m * (NSLOT * HX_FRAG), lane, [](int) {});
        if constexpr (ROLE_ID == 2) {
            // late: commit the next unit's window and request the one two units on AFTER the products (the early wave of this
            // SIMD committed while this one multiplied)
            stamp(7);
            if (!(FASTSVC_DBG_ON(p, DBG_NO_COMMIT))) { pcommit(SETC, tiles + ((u + 1) & 1) * bufsz); pload(SETC, u + 3); }
            stamp(5);
        } else stamp(7);
        // unit u + 1's weights have landed (the waves that requested them hold no other request but an epilogue's stores)
        if constexpr (ROLE_ID == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(10); }
        __syncthreads();
        stamp(6);
        if (++ch == nch) { ch = 0; ++tl; }
    };
    // the loop always runs both halves (an odd count ends with a phantom unit: no products, but the last tile's epilogue)
    const int nun2 = (nunits + 1) & ~1;
    const int role = !stager ? 0 : wave < 4 ? 1 : 2;
    if (role == 1) { for (int u = 0; u < nun2; u += 2) { unit(WxSet<1>{}, Set1{}, u); unit(WxSet<1>{}, Set0{}, u + 1); } }
    else if (role == 2) { for (int u = 0; u < nun2; u += 2) { unit(WxSet<2>{}, Set1{}, u); unit(WxSet<2>{}, Set0{}, u + 1); } }
    else { for (int u = 0; u < nun2; u += 2) { unit(WxSet<0>{}, Set1{}, u); unit(WxSet<0>{}, Set0{}, u + 1); } }
    if (!(nunits & 1)) epilogue(ntiles - 1);
    if constexpr (TRACKS) amax_flush(p, R, &s_amax, &s_cnt, WX_NWAVES, sig, b, lane, blockIdx.x);
    if (flags & F_STATS) {                                     // one f64 global atomic per channel per workgroup
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += WX_NT) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}

constexpr size_t WX_STATIC_LDS = 1024;
template <int MW, int NW, int WM, int WN>
static size_t wx_smem(const ConvParams& p) {
    constexpr int NT = 16 * NW * WN;
    const int halo_al = (p.dil + 3) & ~3;
    const int W = NT + 2 * halo_al;
    size_t s = sizeof(double) * 2 * 16 * MW * WM + sizeof(float) * 2 * ((size_t)p.nch32 * HX_KC + 8) +
               (size_t)2 * HX_NP * (W + 4) * HX_ROW + (size_t)2 * WM * 3 * MW * HX_NP * HX_FRAG;
#ifdef FASTSVC_ACT_BF16
    s += sizeof(float) * WX_NWAVES * 16 * 36;
#endif
    return s;
}

template <auto KERNEL>
static hipError_t wx_launch_instance(dim3 grid, size_t smem, hipStream_t stream, const ConvParams& p) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - WX_STATIC_LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(KERNEL, grid, dim3(WX_NT), smem, stream, p);
    return hipGetLastError();
}

template <int MW, int NW, int WM, int WN>
static hipError_t wx_launch_shape(const ConvParams& p, int nsig, hipStream_t stream) {
    constexpr int NT = 16 * NW * WN;
    if (p.ngroups % WM != 0) return hipErrorInvalidValue;
    const size_t smem = wx_smem<MW, NW, WM, WN>(p);
    if (smem + WX_STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
    const int ntx = (p.T + NT - 1) / NT;
    const int tpw = p.tpw > 0 ? p.tpw : 1;
    dim3 grid((ntx + tpw - 1) / tpw, p.ngroups / WM, nsig * p.B);
    const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
    const int kind = aff ? EPI_AFF : p.res ? EPI_RES : EPI_PLAIN;
#define FASTSVC_WX(k) if (kind == k) return wx_launch_instance<&conv_wx_kernel<MW, NW, WM, WN, k>>(grid, smem, stream, p);
    FASTSVC_WX(EPI_PLAIN) FASTSVC_WX(EPI_RES) FASTSVC_WX(EPI_AFF)
#undef FASTSVC_WX
    return hipErrorInvalidValue;
}

hipError_t launch_conv_wx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
#ifndef FASTSVC_ACT_BF16
    return hipErrorInvalidValue;                               // (float32 storage: not built yet)
#else
    if (p.mode != MODE_DIRECT || p.ntaps != 3 || (p.T & 3) || !p.whx || p.nch32 * HX_KC > WX_NT || p.dil < 1 || p.dil > 28 ||
        p.x2 || p.r1x || p.last_w || p.xsplit || (p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0)) || cfg.MW != 3)
        return hipErrorInvalidValue;
    if (cfg.NW == 8 && cfg.WM == 4 && cfg.WN == 2) return wx_launch_shape<3, 8, 4, 2>(p, cfg.nsig, stream);
    if (cfg.NW == 6 && cfg.WM == 2 && cfg.WN == 4) return wx_launch_shape<3, 6, 2, 4>(p, cfg.nsig, stream);
    if (cfg.NW == 4 && cfg.WM == 4 && cfg.WN == 2) return wx_launch_shape<3, 4, 4, 2>(p, cfg.nsig, stream);
    return hipErrorInvalidValue;
#endif
}

#ifndef FASTSVC_ACT_BF16      // storage-independent host query: defined once
bool conv_wx_shape(int mode, int MW, int NW, int WM, int WN) {
    if (mode != MODE_DIRECT || MW != 3) return false;
    return (NW == 8 && WM == 4 && WN == 2) || (NW == 6 && WM == 2 && WN == 4) || (NW == 4 && WM == 4 && WN == 2);
}
#endif

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
