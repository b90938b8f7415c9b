// fastsvc_wx.hip - the WIDE-layer (C_out = 192 / 384) implicit-GEMM convolution kernel of the FastSVC generator forward for
// gfx950 (CDNA4 / MI355X): the same k=3 dilated "same" convolutions, prologues and epilogues as fastsvc_hx.hip (reference:
// Conv1d1x3 / Conv2d1x3, harana/layers/upsample.py:76-83,99-106, used by harana/models/fastsvc.py:94-112,164-178,209-232),
// laid out for the layers whose roofline is the MATRIX pipe as much as HBM.
//
// What differs from conv_hx_kernel (4 consumer + 4 staging waves, weights in a per-wave register ring):
//   * EVERY wave multiplies.  A workgroup is 8 waves = 4 channel groups (of 48 output channels) x 2 time slices (of 96
//     columns): 192 channels x 192 columns per tile; two waves share each SIMD's matrix pipe, so one wave's LDS round trips,
//     staging work and epilogue sit under the other's products.  conv_hx's consumer wave is alone on its SIMD's pipe: its serial
//     fragment-read -> product chain kept the pipe 21-32 % busy on these layers (profiles/r5c_cfg3_bfloat16_sq_counters.csv).
//   * Roles by wave, one of each per SIMD: waves 0-3 STAGE the activation window of the next unit (a 192-column tile plus its
//     halo, aligned to 8 rows, is at most 256 rows: one (quad of 4 channels, octet of 8 rows) item per lane = four 16-byte
//     requests; until the end of round 6 (octet, quad) items of eight 8-byte requests: a CU takes 8-byte requests at ~13 B/clk
//     at most, even out of L2 - tools/micro/cu_pull.hip -, film.2.heads 217 -> 196 us), waves 4-7 bring the next unit's
//     WEIGHTS - wave 4 + g the nine KB-sized fragments of channel group g: global -> registers a unit ahead, registers -> LDS
//     (lane-linear ds_write_b128) at the top of the unit.  Every wave reads its group's fragments back with conflict-free
//     ds_read_b128: the weights cross L2 -> CU ONCE per workgroup (conv_hx: once per consumer wave and tile).
//     (First version: LDS-DMA pieces - `buffer_load_dwordx4 ... lds` - measured ~195 cycles of issue PER PIECE inside this loop,
//     2.3k cycles per unit for the three waves that issued them, profiles/r6a_timeline_conv_wx_v1.txt.)
//   * Window requests run TWO units ahead through two register sets (one unit of products is shorter than a memory round trip
//     under load), each role's loop is straight-line in what concerns its memory requests (a branch between a request and
//     its use makes hipcc wait for the younger set as well: fastsvc_hx.hip's lesson).
//   * What it is bound by (profiles/r6b_*): the ablations ADD UP - products 110 us + window requests 68 + weight
//     requests 33 + epilogue 50-60 for film.2.heads at 64 x 12000 - i.e. the waves wait for their requests instead of
//     multiplying: an 8-byte-per-lane window request (4 rows x 128 bytes) cost ~47 cycles, a weight request (1 KB contiguous)
//     ~20, wherever in the unit it stood (top of the unit or between the product steps), ~11 B/clk per CU of window data -
//     hence the 16-byte requests above.  Tried against
//     it and dropped: the layer's weights RESIDENT in LDS (96 output channels per workgroup, windows requested four units
//     ahead: both channel halves of a C = 192 layer then read every window - 240 us), LDS-DMA for the weights (~195 cycles
//     of issue per 1 KB piece).
//   * The staging commit is specialised by prologue (PRO): 0 = none (the tensor is already what the conv reads: film.conv,
//     the FiLM heads, conv_first) - a 4 x 8 bfloat16 transpose by 16 byte-permutes per item; 1 = LeakyReLU only (c2 / c3 of a
//     conditioning stage); 2 = InstanceNorm + speaker bias + LeakyReLU as one FMA per value (the up blocks' d = 9 / 27 convs).
//   * Epilogue of tile t runs AFTER the barrier that ends its last unit and behind the next unit's requests: bias / LeakyReLU /
//     residual / FiLM affine in the MFMA layout with ALL of the tile's operands requested up front, bfloat16 pairs through a
//     wave-private LDS patch of 16 channel rows, out in 16-byte pieces of 192-byte row runs (one LDS hand-over per channel
//     tile instead of two round trips per 16 x 32 item).
// bfloat16 storage only (float32 storage: hi + lo planes and fragments need twice the LDS).
#include "fastsvc_kernels.h"

namespace fastsvc {
#ifdef FASTSVC_ACT_BF16
namespace bf16 {
#endif

#include "fastsvc_device.inc"
#include "fastsvc_hx_common.inc"

constexpr int WX_NWAVES = 8;
constexpr int WX_NT = WX_NWAVES * 64;
constexpr int WX_MW = 3, WX_NW = 6, WX_WM = 4, WX_WN = 2;          // the one shape: 48 channels x 96 columns per wave
constexpr int WX_TILE = 16 * WX_NW * WX_WN;                        // 192 columns per workgroup tile
constexpr int WX_PATCH_PITCH = 208;                                // bytes of a patch row: 96 bfloat16 + 16 (16-byte aligned pieces; the 8-byte writes of a 16-lane group are 2-way: 8 cycles for a 6-cycle issue)
constexpr int WX_PATCH = 16 * WX_PATCH_PITCH;                      // one wave's epilogue patch

// One unit = (32-channel K chunk) x 3 taps on this wave's NW time tiles and MW channel tiles.
//   tile: the staged window (time-major rows of 32 channels), aoff[tap]: this lane's fragment offset of time tile 0
//   wl:   this wave's group's fragments of the unit in LDS, [tap][m][piece] x 1 KB, lane-linear
// A fragments are read PF steps ahead, a tap's weight fragments while the previous tap's last steps run.
// on_step(s): the wave's memory work of this unit, a slice per step in front of the step's products (see conv_wx_kernel).
template <int MW, int NW, class F>
__device__ __forceinline__ void wx_unit(f32x4 (&acc)[NW][MW], const unsigned char* tile, const int (&aoff)[3], int lo_off,
                                        const unsigned char* wl, int lane, F&& on_step) {
    constexpr int NSTEP = 3 * NW;
    constexpr int PF = HX_NP == 1 ? 2 : 1;
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

template <int V> struct WxC { static constexpr int value = V; };        // a compile-time index (register set, role)

#ifdef FASTSVC_ACT_BF16
// Plain / residual epilogue of a wave's 48-channel x 96-column tile (see the file header).  acc: [n][m] result tiles in the
// MFMA layout (lane (co = lane & 15, q = lane >> 4): channel co, columns 16 n + 4 q ..+3).
//   v = lrelu?(acc + bias) [+ residual]  ->  bfloat16, 8 bytes per tile into the patch row of the channel;
//   per channel tile: 16 rows x 192 bytes = 192 16-byte pieces = 3 per lane, stored as row runs.
// The residual is fetched in the MFMA layout (8 bytes per lane: 16 rows x 32-byte segments per request - a poor shape for
// HBM, but it keeps the float32 sum in the order conv_hx computes it: bit-identical results).
// EPI_AFF (the middle convs of an up block, fastsvc.py:98-104 + 115-140): v = lrelu?(acc + bias) -> y (when the block needs the
// pre-affine tensor), u = scale * v + shift -> y2, InstanceNorm partial sums of u into s1 / s2 (per lane; the caller joins them).
// The operands - residual, or scale and shift, 8 bytes per lane and result tile - are requested a channel tile (6 - 12 requests)
// ahead of their use: one exposed memory round trip per tile (fetched item by item, as conv_hx's pair epilogue does with
// its smaller tiles, each of the 9-12 items waited out its own: 23-37k cycles per tile, profiles/r6a_timeline_conv_wx_v1.txt).
template <int EPI>
__device__ __forceinline__ void wx_epilogue_rows(const ConvParams& p, const EpiRsrc& R, f32x4 (&acc)[WX_NW][WX_MW], const float (&k_bias)[WX_MW],
                                                 float (&s1)[WX_MW], float (&s2)[WX_MW], int mg, int tcol0, int lane, unsigned char* patch) {
    const float slope = (p.flags & F_POST_LRELU) ? LRELU_SLOPE : 1.0f;     // max(v, slope * v): identity for 1
    const int co = lane & 15, q = lane >> 4;
    unsigned char* wrow = patch + co * WX_PATCH_PITCH + q * 8;
    // this lane's three pieces of a channel tile: piece id = 64 k + lane -> (row, 16-byte piece of the row's 12)
    int prow[3], pcol[3];
    #pragma unroll
    for (int k = 0; k < 3; ++k) { const int pid = 64 * k + lane; prow[k] = pid / 12; pcol[k] = pid - prow[k] * 12; }
    const int shift_soff = p.COUT * p.ldy * 4;                 // ("float bytes": the helpers halve them)
    constexpr int NOP = EPI == EPI_AFF ? 2 : EPI == EPI_RES ? 1 : 0;
    // (a channel tile's operands are requested while the tile before it is finished: two sets; all three tiles' at once -
    // 72 registers for the FiLM kind - spilled)
    constexpr int NSET = EPI == EPI_AFF ? 1 : 2;               // (FiLM kind: 24 registers per set - two sets spill; its tiles wait out a round trip each)
    act4_t ops[NSET][NOP > 0 ? NOP : 1][WX_NW];
    auto request = [&](int m, act4_t (&o)[NOP > 0 ? NOP : 1][WX_NW]) __attribute__((always_inline)) {
        if constexpr (NOP > 0) {
            const int cot = (mg * WX_MW + m) * 16 + co;
            #pragma unroll
            for (int n = 0; n < WX_NW; ++n) {
                const int t = tcol0 + n * 16 + q * 4;
                const int off = (cot < p.COUT && t < p.T) ? (cot * p.ldy + t) * 4 : OOB_OFF;
                if constexpr (EPI == EPI_RES) o[0][n] = act_load4_raw(R.res, off, 0);
                if constexpr (EPI == EPI_AFF) { o[0][n] = act_load4_raw(R.ss, off, 0); o[1][n] = act_load4_raw(R.ss, off, shift_soff); }
            }
        }
    };
    request(0, ops[0]);
    auto rows_out = [&](__amdgpu_buffer_rsrc_t dst, int m) __attribute__((always_inline)) {
        // lanes read what OTHER lanes of the wave wrote (a wave's LDS operations execute in order: no barrier, but hipcc
        // must not move the reads above the writes - per thread the addresses never overlap)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        #pragma unroll
        for (int k = 0; k < 3; ++k) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(patch + prow[k] * WX_PATCH_PITCH + pcol[k] * 16);
            const int cr = (mg * WX_MW + m) * 16 + prow[k];
            const int t = tcol0 + pcol[k] * 8;
            // (rows are a multiple of 4 long: a piece is whole, half, or outside)
            const int nv = (cr < p.COUT && t < p.T) ? min(8, p.T - t) : 0;
            act_store8_raw(dst, (cr * p.ldy + t) * 2, w, nv);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    #pragma unroll
    for (int m = 0; m < WX_MW; ++m) {
        const bool cok = (mg * WX_MW + m) * 16 + co < p.COUT;
        if (NSET == 2 && m + 1 < WX_MW) request(m + 1, ops[(m + 1) % NSET]);
        if (NSET == 1 && m > 0) request(m, ops[0]);
        #pragma unroll
        for (int n = 0; n < WX_NW; ++n) {
            f32x4 v = acc[n][m] + k_bias[m];
            v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
            v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
            if constexpr (EPI == EPI_RES) v += act_unpack4(ops[m % NSET][0][n]);
            acc[n][m] = v;
            u32x2v w;
            w.x = bf16_pack2(v.x, v.y);
            w.y = bf16_pack2(v.z, v.w);
            *reinterpret_cast<u32x2v*>(wrow + n * 32) = w;
        }
        if (EPI != EPI_AFF || p.y) rows_out(R.y, m);            // (wave-uniform)
        if constexpr (EPI == EPI_AFF) {
            #pragma unroll
            for (int n = 0; n < WX_NW; ++n) {
                const int t = tcol0 + n * 16 + q * 4;
                f32x4 u = act_unpack4(ops[m % NSET][0][n]) * acc[n][m] + act_unpack4(ops[m % NSET][1][n]);
                if (!(cok && t < p.T)) u = f32x4{0.f, 0.f, 0.f, 0.f};      // (rows are a multiple of 4 long)
                s1[m] += (u.x + u.y) + (u.z + u.w);
                s2[m] += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
                u32x2v w;
                w.x = bf16_pack2(u.x, u.y);
                w.y = bf16_pack2(u.z, u.w);
                *reinterpret_cast<u32x2v*>(wrow + n * 32) = w;
            }
            rows_out(R.y2, m);
        }
    }
}
#endif

#ifdef FASTSVC_ACT_BF16           // (bfloat16 storage only: the float32 unit keeps the host queries below)
// PRO: staging prologue (0 none, 1 LeakyReLU, 2 InstanceNorm + speaker bias + LeakyReLU); EPI: epilogue kind (fastsvc_device.inc)
template <int PRO, int EPI>
__global__ __launch_bounds__(WX_NT, 2)
void conv_wx_kernel(const ConvParams p0) {
    constexpr int MW = WX_MW, NW = WX_NW, WM = WX_WM, WN = WX_WN;
    constexpr int NT = WX_TILE;
    constexpr int NSLOT = 3 * MW * HX_NP;                              // fragments of one (group, chunk)
    constexpr int WUNIT = WM * NSLOT * HX_FRAG;                        // bytes of one unit's weights (all groups)
    constexpr bool ROWS_EPI = HX_NP == 1;                              // (bfloat16 storage: the row-run epilogue for every kind)
    static_assert(WM * WN == WX_NWAVES, "eight waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = wave < 4;                                      // (wave-uniform) waves 0-3 stage windows, 4-7 bring weights
    const int wave_m = wave & 3;                                       // waves w and w + 4 share a SIMD: one of each role per SIMD,
    const int wave_n = wave >> 2;                                      //   channel group w & 3, time slice w >> 2
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
    const int halo_al = (halo + 7) & ~7;                               // (8-row items, see the staging)
    const int W = NT + 2 * halo_al;                                    // window rows (<= 256)
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
    unsigned char* tiles = reinterpret_cast<unsigned char*>(ncoef + CINp + 8);         // [2][W + 8 rows][64 B]
    const int lo_off = (W + 8) * HX_ROW;
    const int bufsz = HX_NP * (W + 8) * HX_ROW;
    unsigned char* wbuf = tiles + 2 * bufsz;                                           // [2][WM][tap][m] fragments
    unsigned char* patch = wbuf + 2 * WUNIT + wave * WX_PATCH;                         // this wave's epilogue patch (>= hx_pair's 16 x 36 floats)
    float* Xw = reinterpret_cast<float*>(patch);
    (void)Xw;
    __shared__ float s_inv[4];

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

    // ---- weights (waves 4-7): group g = wave - 4, unit un -> its nine fragments [tap][m], contiguous in the blob ----
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.whx) + (long)sig * p.whx_sig +
                                (long)(blockIdx.y * WM + wave_m) * nch * NSLOT * HX_FRAG;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wsrc), 0, nch * NSLOT * HX_FRAG, 0x00020000);
    u32x4 wreg[NSLOT];
    int w_ch = 0;
    int w_soff = 0;                                                    // scalar offset of the unit being requested
    auto wload_begin = [&](int un) __attribute__((always_inline)) {    // (called once per unit, in unit order)
        w_soff = un < nunits ? w_ch * (NSLOT * HX_FRAG) : nch * NSLOT * HX_FRAG;   // past the last unit: out of range, nothing fetched
        w_ch = w_ch + 1 == nch ? 0 : w_ch + 1;
    };
    auto wload_frag = [&](int f) __attribute__((always_inline)) {
        wreg[f] = __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, w_soff + f * HX_FRAG, 0);
    };
    auto wcommit_frag = [&](unsigned char* dst, int f) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(dst + f * HX_FRAG + lane * 16) = wreg[f];
    };
    auto wload = [&](int un) __attribute__((always_inline)) {
        wload_begin(un);
        #pragma unroll
        for (int f = 0; f < NSLOT; ++f) wload_frag(f);
    };
    auto wcommit = [&](unsigned char* dst) __attribute__((always_inline)) {
        #pragma unroll
        for (int f = 0; f < NSLOT; ++f) wcommit_frag(dst, f);
    };

    // ---- window staging (waves 0-3): item = (quad of 4 channels, octet of 8 rows), one per lane: four 16-byte requests
    // (8 time steps of one channel each, 8 rows x 128 B per wave request).  A CU's vector-memory pipe takes a wave's
    // request in about the same time whatever its width (tools/micro/cu_pull.hip: 8 B per lane stops at ~13 B/clk per CU
    // even out of L2, 16 B per lane reaches 19-47): half the requests for the same window.  The halo is aligned to 8 rows,
    // so an item starts inside the utterance or lies outside it; at a row's end (a multiple of 4) its last 4 rows may.
    const __amdgpu_buffer_rsrc_t xr = act_rsrc(p.x, (long)sig * p.x_sig + (long)b * p.x_b, (long)p.CIN * p.ldx);
    const int it_cq = tid & 7;
    const bool it_in = (tid >> 3) < (W >> 3);
    const int it_ro = it_in ? (tid >> 3) : (W >> 3);                   // (lanes without an item: parked in the 8 spare rows behind the tile)
    const float slope = (PRO >= 1 && (flags & F_PRE_LRELU)) ? LRELU_SLOPE : 1.0f;    // max(v, slope * v): identity for 1
    const bool row_end = (p.T & 7) != 0;                               // (uniform) the last item of a row holds 4 rows of it
    int l_tl = 0, l_ch = 0, c_ch = 0;
    // window w is requested into register set w & 1 two units before its commit
    u32x4 pxs[2][4];
    int nvs[2] = {0, 0};                                               // rows of the item inside the utterance: 0, 4 or 8
    int p_voff[2] = {0, 0}, p_soff[2] = {0, 0};
    auto pload_begin = [&](auto SETC, int un) __attribute__((always_inline)) {     // (called once per unit, in unit order)
        constexpr int SET = decltype(SETC)::value;
        const int tl_ = l_tl, ch_ = l_ch;
        { const bool wrap = l_ch + 1 == nch; l_ch = wrap ? 0 : l_ch + 1; l_tl += wrap ? 1 : 0; }
        const int t = (tile0 + tl_) * NT - halo_al + 8 * it_ro;
        const int rows_left = p.CIN - ch_ * HX_KC;
        // (C_in is a multiple of 8: a quad lies inside the tensor or behind it as a whole)
        const bool tok = it_in & ((unsigned)t < (unsigned)p.T) & (un < nunits) & (it_cq * 4 < rows_left);
        nvs[SET] = tok ? min(8, p.T - t) : 0;
        // ONE per-lane offset per item (bytes); the channel of the quad rides in the scalar offset
        p_voff[SET] = tok ? (it_cq * 4 * p.ldx + t) * 2 : OOB_OFF;
        p_soff[SET] = ch_ * HX_KC * p.ldx * 2;
    };
    auto pload_chan = [&](auto SETC, int c) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETC)::value;
        pxs[SET][c] = __builtin_amdgcn_raw_buffer_load_b128(xr, p_voff[SET], p_soff[SET] + c * p.ldx * 2, FASTSVC_LD_AUX);
    };
    auto pload = [&](auto SETC, int un) __attribute__((always_inline)) {
        pload_begin(SETC, un);
        #pragma unroll
        for (int c = 0; c < 4; ++c) pload_chan(SETC, c);
    };
    // commit of one row (time step j of the item's octet) of the window in set SET
    float cA[4], cB[4];                                                // (PRO 2: the item's (A, Bc) coefficients, read once per commit)
    auto pcommit_begin = [&](auto SETC) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETC)::value;
        const int ch_ = c_ch;
        c_ch = c_ch + 1 == nch ? 0 : c_ch + 1;
        if constexpr (PRO == 2) {
            // (A, Bc) of the item's 4 channels: InstanceNorm + speaker bias as ONE FMA u * A + Bc; rows outside the utterance
            // are the conv's zero padding AFTER the prologue: their loads returned 0, so only the additive term has to go - such
            // an item reads the (0, 0) coefficients behind the table
            const f32x4* cf = reinterpret_cast<const f32x4*>(ncoef + (nvs[SET] != 0 ? ch_ * HX_KC + it_cq * 4 : CINp));
            const f32x4 c0 = cf[0], c1 = cf[1];
            cA[0] = c0.x; cA[1] = c0.z; cA[2] = c1.x; cA[3] = c1.z;
            cB[0] = c0.y; cB[1] = c0.w; cB[2] = c1.y; cB[3] = c1.w;
        }
    };
    auto pcommit_row = [&](auto SETC, unsigned char* tile, int j) __attribute__((always_inline)) {
        constexpr int SET = decltype(SETC)::value;
        const u32x4 (&px)[4] = pxs[SET];
        unsigned char* d = tile + hx_lds_off(8 * it_ro + j, it_cq >> 1) + (it_cq & 1) * 8;
        u32x2v o;
        if constexpr (PRO == 0) {
            // the tensor is what the conv reads: rows outside the utterance and channel padding were fetched as zeros.
            // 4 x 8 transpose of 16-bit values: dword k of row j = (channel 2k | channel 2k + 1 << 16) at time step j
            #pragma unroll
            for (int k = 0; k < 2; ++k)
                o[k] = __builtin_amdgcn_perm(px[2 * k + 1][j >> 1], px[2 * k][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
        } else {
            float e[4];
            #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned w = px[c][j >> 1];
                float v = __builtin_bit_cast(float, (j & 1) ? (w & 0xffff0000u) : (w << 16));
                if constexpr (PRO == 2) v = v * cA[c] + cB[c];
                e[c] = fmaxf(v, v * slope);
            }
            o.x = bf16_pack2(e[0], e[1]);
            o.y = bf16_pack2(e[2], e[3]);
        }
        if (j >= 4 && row_end) {                                       // (what lies behind the row's end is nobody's data)
            o.x = nvs[SET] > 4 ? o.x : 0u;
            o.y = nvs[SET] > 4 ? o.y : 0u;
        }
        *reinterpret_cast<u32x2v*>(d) = o;
    };
    auto pcommit = [&](auto SETC, unsigned char* tile) __attribute__((always_inline)) {
        pcommit_begin(SETC);
        #pragma unroll
        for (int j = 0; j < 8; ++j) pcommit_row(SETC, tile, j);
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
    const EpiConst<MW, true, false> K{k_bias, k_bias, k_bias, k_bias, s_inv, 0};

    // ---- set-up: the first requests in flight, then the tables ----
    typedef WxC<0> Set0;
    typedef WxC<1> Set1;
    if (stager) { pload(Set0{}, 0); pload(Set1{}, 1); }
    else wload(0);
    {
        // prologue coefficients of every input channel (PRO 2): InstanceNorm + speaker bias (u - mean) * rstd + p  ->
        // A = rstd, Bc = p - mean * rstd (fastsvc.py:134-139); channel padding: (0, 0)
        const int c = tid;
        if constexpr (PRO == 2) {
            double q1 = 0.0, q2 = 0.0;
            float pc = 0.f;
            if ((flags & F_PRE_NORM) && c < p.CIN) {
                q1 = p.st_in[((long)b * p.CIN + c) * 2 + 0];
                q2 = p.st_in[((long)b * p.CIN + c) * 2 + 1];
                pc = p.spk[(long)b * p.CIN + c];
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
        }
        if (flags & F_STATS) {
            for (int i = tid; i < 2 * 16 * MW * WM; i += WX_NT) sstat[i] = 0.0;
        }
        __syncthreads();
    }
    if (stager) { pcommit(Set0{}, tiles); pload(Set0{}, 2); }
    else { wcommit(wbuf + wave_m * (NSLOT * HX_FRAG)); wload(1); }
    __syncthreads();
    stamp(4);

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
        wx_epilogue_rows<EPI>(p, R, acc, k_bias, s1, s2, mg, tcolw, lane, patch);
#else
        ws_epilogue_kind<MW, NW, EPI, false, 0, -1>(p, R, acc, s1, s2, sig, mg, tcolw, true, lane, K, nullptr);
#endif
        if (EPI == EPI_AFF && (flags & F_STATS)) {
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
    // requests): 1 = stages windows, 0 = brings weights.  At the top of unit u the wave hands over what unit u + 1 needs
    // (buffers (u + 1) & 1 were last read in unit u - 1: behind a barrier) and requests what comes after; past the last unit
    // the requests fetch nothing (offsets out of range) and what is handed over lands in the buffers nobody reads.
    // SETC: the register set that holds window u + 1.
    auto unit = [&](auto ROLE, auto SETC, int u) __attribute__((always_inline)) {
        constexpr int ROLE_ID = decltype(ROLE)::value;
        // At the top of unit u the wave hands over what unit u + 1 needs (buffers (u + 1) & 1 were last read in unit u - 1: behind
        // a barrier) and requests what comes after.  (Riding between the product steps instead, a slice per step, the same work
        // cost MORE: 245 against 213 us for film.2.heads at 64 x 12000 - what a wave waits for are its requests, wherever they
        // stand: skipping the window requests alone took 68 us off, the weight requests 33, the LDS stores nothing.)
        if constexpr (ROLE_ID == 1) {
            if (!(FASTSVC_DBG_ON(p, DBG_NO_COMMIT))) { pcommit(SETC, tiles + ((u + 1) & 1) * bufsz); pload(SETC, u + 3); }
        } else {
            if (!(FASTSVC_DBG_ON(p, DBG_NO_WEIGHTS))) { wcommit(wbuf + ((u + 1) & 1) * WUNIT + wave_m * (NSLOT * HX_FRAG)); wload(u + 2); }
        }
        asm volatile("" ::: "memory");                         // (the requests go out HERE: hipcc otherwise sinks them below the products)
        stamp(5);
        if (ch == 0) {
            if (tl > 0) { epilogue(tl - 1); stamp(8); }        // the tile that ended with the last barrier
            #pragma unroll
            for (int n = 0; n < NW; ++n)
                #pragma unroll
                for (int m = 0; m < MW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (u < nunits && !(FASTSVC_DBG_ON(p, DBG_NO_MFMA)))
            wx_unit<MW, NW>(acc, tiles + (u & 1) * bufsz, aoff, lo_off, wbuf + (u & 1) * WUNIT + wave_m * (NSLOT * HX_FRAG), lane, [](int) {});
        stamp(7);
        __syncthreads();
        stamp(6);
        if (++ch == nch) { ch = 0; ++tl; }
    };
    // the loop always runs both halves (an odd count ends with a phantom unit: no products, but the last tile's epilogue)
    const int nun2 = (nunits + 1) & ~1;
    if (stager) { for (int u = 0; u < nun2; u += 2) { unit(WxC<1>{}, Set1{}, u); unit(WxC<1>{}, Set0{}, u + 1); } }
    else { for (int u = 0; u < nun2; u += 2) { unit(WxC<0>{}, Set1{}, u); unit(WxC<0>{}, Set0{}, u + 1); } }
    if (!(nunits & 1)) epilogue(ntiles - 1);
    if (flags & F_STATS) {                                     // one f64 global atomic per channel per workgroup
        __syncthreads();
        for (int i = tid; i < 2 * 16 * MW * WM; i += WX_NT) {
            const int co = blockIdx.y * (WM * MW * 16) + (i >> 1);
            if (co < p.COUT) atomicAdd(&p.st_out[((long)b * p.COUT + co) * 2 + (i & 1)], sstat[i]);
        }
    }
}
#endif

constexpr size_t WX_STATIC_LDS = 256;
static size_t wx_smem(const ConvParams& p, int np = HX_NP) {     // np: operand pieces (1: bfloat16 storage)
    const int halo_al = (p.dil + 7) & ~7;
    const int W = WX_TILE + 2 * halo_al;
    return sizeof(double) * 2 * 16 * WX_MW * WX_WM + sizeof(float) * 2 * ((size_t)p.nch32 * HX_KC + 8) +
           (size_t)2 * np * (W + 8) * HX_ROW + (size_t)2 * WX_WM * 3 * WX_MW * np * HX_FRAG + (size_t)WX_NWAVES * WX_PATCH;
}

template <auto KERNEL>
static hipError_t wx_launch_instance(dim3 grid, size_t smem, hipStream_t stream, const ConvParams& p) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - WX_STATIC_LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(KERNEL, grid, dim3(WX_NT), smem, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_wx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream) {
#ifndef FASTSVC_ACT_BF16
    return hipErrorInvalidValue;                               // (float32 storage: not built)
#else
    if (p.mode != MODE_DIRECT || p.ntaps != 3 || (p.T & 3) || !p.whx || p.nch32 * HX_KC > WX_NT || p.dil < 1 || p.dil > 28 ||
        p.x2 || p.r1x || p.last_w || p.xsplit || (p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0)) || (p.CIN & 7) ||
        (p.flags & F_PRE_AFFINE) || cfg.MW != WX_MW || cfg.NW != WX_NW || cfg.WM != WX_WM || cfg.WN != WX_WN || p.ngroups % WX_WM != 0)
        return hipErrorInvalidValue;
    const size_t smem = wx_smem(p);
    if (smem + WX_STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
    const int ntx = (p.T + WX_TILE - 1) / WX_TILE;
    const int tpw = p.tpw > 0 ? p.tpw : 1;
    dim3 grid((ntx + tpw - 1) / tpw, p.ngroups / WX_WM, cfg.nsig * p.B);
    const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
    const int kind = aff ? EPI_AFF : p.res ? EPI_RES : EPI_PLAIN;
    const int pro = (p.flags & F_PRE_NORM) ? 2 : (p.flags & F_PRE_LRELU) ? 1 : 0;
#define FASTSVC_WX(pr, k) if (pro == pr && kind == k) return wx_launch_instance<&conv_wx_kernel<pr, k>>(grid, smem, stream, p);
    FASTSVC_WX(0, EPI_PLAIN) FASTSVC_WX(0, EPI_RES)
    FASTSVC_WX(1, EPI_PLAIN) FASTSVC_WX(1, EPI_RES)
    FASTSVC_WX(2, EPI_PLAIN) FASTSVC_WX(2, EPI_RES) FASTSVC_WX(2, EPI_AFF)
#undef FASTSVC_WX
    return hipErrorInvalidValue;
#endif
}

#ifndef FASTSVC_ACT_BF16      // storage-independent host queries: defined once
bool conv_wx_shape(int mode, int MW, int NW, int WM, int WN) {
    return mode == MODE_DIRECT && MW == WX_MW && NW == WX_NW && WM == WX_WM && WN == WX_WN;
}
// whether the window buffers (dilation dil), the two units of weights and the patches fit the CU's LDS
bool conv_wx_fits(int nch32, int dil) {
    ConvParams q{};
    q.nch32 = nch32; q.dil = dil;
    return wx_smem(q, 1) + WX_STATIC_LDS <= 160 * 1024;       // (the kernel exists in bfloat16 storage only)
}
#endif

#ifdef FASTSVC_ACT_BF16
}  // namespace bf16
#endif
}  // namespace fastsvc
