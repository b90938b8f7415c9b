// fastsvc_plan.cpp - host side of the C ABI (include/fastsvc_hip.h): layer table, weight-norm fold,
// MFMA-fragment weight packing, workspace layout and the launch sequence of one forward pass.
//
// Dataflow = SURVEY.md Appendix B (the de-duplicated restatement of
// harana/models/fastsvc.py:305-340): each conditioning chain is evaluated once, the 1x1 residual
// conv is commuted past the decimation, the two FiLM heads of the two signals are one
// (2C -> 2C) convolution whose K dimension concatenates [u_lft ; u_sine] so that
// scale = scale_lft + scale_sine falls out of the accumulation (fastsvc.py:129-130), and the
// InstanceNorm statistics are produced by the epilogue of the kernel that writes the tensor.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <atomic>
#include <string>
#include <unordered_map>
#include <vector>

#include "fastsvc_hip.h"
#include "fastsvc_kernels.h"

using namespace fastsvc;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A convolution whose weights live in the blob in MFMA fragment order.
struct PackedConv {
    int cin = 0, cout = 0, ntaps = 3, dil = 1;
    int MW = 1, KC = 4, nchunks = 1, Q = 0, ngroups = 1;
    size_t w_off = 0, b_off = 0;      // float offsets into the blob
    size_t w_floats = 0, b_floats = 0;
    // stretch + conv layers also carry the polyphase taps W0 | W0+W1+W2 | W2 (MODE_POLY), same
    // fragment layout and size as the plain weights
    bool poly = false;
    size_t wp_off = 0;
    // k=3, d in {1,2,4}, 48 | C_out layers also carry the Winograd F(2,3) weights
    // g = w0 | (w0+w1+w2)/2 | (w0-w1+w2)/2 | w2 (MODE_WINO): 4 "taps", Qw = 4 * nchunks * 6 k-steps
    bool wino = false;
    size_t ww_off = 0;
    long ww_pair = 0;                 // lft -> sine stride of the Winograd weights (paired convs)
    int Qw = 0;
    // the same with 32-channel groups (MW = 2; 32 | C_out): four accumulator sets + the weight
    // ring then fit 128 VGPRs, i.e. two workgroups per CU
    bool wino2 = false;
    size_t ww2_off = 0;
    long ww2_pair = 0;
    // MODE_DEC2 (first k=3 conv of a down stage + the stage's 1x1 residual conv in one launch):
    // w_off holds 4 "taps" per k-group (w0 w1 w2 | w1x1), Q = 24 * nchunks; second bias at b2_off
    bool dec2 = false;
    size_t b2_off = 0;
    long b2_pair = 0;
    // half-precision-MFMA kernels (fastsvc_hx.hip): pre-split fragments [group][32-channel chunk][tap]
    // [16-channel tile][piece][lane][8 halves]; index 0 = binary16 hi + lo pieces (float32 storage),
    // 1 = bfloat16 (bfloat16 storage).  Offsets in floats, pair strides (lft -> sine) in bytes.
    bool hx = false;
    int nch32 = 0;
    size_t hx_off[2] = {0, 0};
    long hx_pair[2] = {0, 0};
    size_t hxp_off[2] = {0, 0};       // the polyphase taps W0 | W0+W1+W2 | W2 in the same format (stretch convs)
    // MODE_CHAIN (this conv fused behind its predecessor, fastsvc_hx.hip): per group [pred's nch32 units |
    // this conv's nch32 units] of 3 slots x MW fragments each; bias of the predecessor at bmid_off
    size_t hxc_off[2] = {0, 0};
    long hxc_pair[2] = {0, 0};
    size_t bmid_off = 0;
    long bmid_pair = 0;
    // inverse per-output-channel weight scales of the split-binary16 fragments (ConvParams::whx_inv; float offsets,
    // pair strides in floats): hx -> [16*MW*ngroups] (MODE_DEC2: two tables), polyphase taps, and the fused pair
    // [first | second | l1_first, bmax_first, l1_in1, bmax_in1]
    size_t hx_inv_off = 0, hxp_inv_off = 0, hxc_inv_off = 0;
    long hx_inv_pair = 0, hxc_inv_pair = 0;
    // (l1, bmax) of the layer - largest absolute row sum of its weights, largest |bias| - i.e. |conv(x)| <= l1 * max|x|
    // + bmax: lets a kernel derive the bound of a tensor nobody measured from the measured maximum of an earlier one
    // (ConvParams::bnd_path)
    size_t bnd_off = 0;
    long bnd_pair = 0;
};

// Source description used by the packer: virtual weight W[co][ci][tap] assembled from up to four
// state-dict layers (FiLM heads) or one.
struct PackSource {
    // pieces: (layer name, co offset, ci offset); all pieces share ntaps
    struct Piece { std::string layer; int co_off; int ci_off; };
    std::vector<Piece> pieces;
    bool dec2 = false;          // pieces = {k=3 conv, 1x1 conv}: packed as MODE_DEC2
};

struct RawParam {            // weights kept in plain row-major layout (VALU kernels)
    std::string layer;
    size_t w_off = 0, b_off = 0;
    size_t w_floats = 0, b_floats = 0;
    size_t bnd_off = 0;      // (l1, bmax) of the layer, see PackedConv
};

struct DownStage {
    int C = 0, Cin = 0, scale = 1;
    // k == 0: c1 and the 1x1 are raw (C_in == 1); k >= 1: packed
    RawParam c1_raw[2], r_raw[2];
    PackedConv r[2], c1[2], c2[2], c3[2];
    PackedConv rc1[2];           // c1 and r fused (MODE_DEC2) when the stage qualifies (24-channel K chunks)
    PackedConv film[2];
    PackedConv heads;
    PackedConv filmc;            // film conv (both signals, block-diagonal 2C -> 2C) -> heads as ONE launch (MODE_CHAIN)
    // whole-stage launches (fastsvc_cond.hip), float32 storage: per-channel bounds of the LDS-resident tensors c1, c2, h,
    // u as affine functions of the stage input's measured maximum, |t[c]| <= alpha[c] * amax + beta[c]:
    // [tensor 0..3][alpha | beta][C] floats per signal (0 = the plan has none)
    size_t cbnd_off[2] = {0, 0};
};

struct UpStage {
    int C = 0, Cin = 0, scale = 1;
    PackedConv first, res, up, d3, d9, d27;
    PackedConv head;             // conv_first -> {res, up} as ONE launch (MODE_UPHEAD): hxc_off[0] / hxc_inv_off only
    RawParam emb;
};

struct BufferSpec { std::string name; size_t off_bytes; int64_t numel; int64_t shape[3]; };

// IEEE binary16 / bfloat16 conversions (round to nearest even), host side of the split-half weights;
// pinned against numpy in tests/test_boundary.py
uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);                // >= 65536: inf
    if (x < 0x38800000u) {                                                  // below 2^-14: subnormal, n * 2^-24
        float a;
        std::memcpy(&a, &x, 4);
        return (uint16_t)(sign | (uint32_t)std::lrintf(a * 16777216.0f));   // lrintf: nearest even
    }
    const uint32_t mant = x & 0x7fffffu;
    uint32_t h = (((x >> 23) - 112u) << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                 // a carry walks into the exponent (65520 -> inf)
    return (uint16_t)(sign | h);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float out;
    if (e == 0) {
        out = std::ldexp((float)m, -24);
    } else if (e == 31) {
        const uint32_t x = 0x7f800000u | (m << 13);
        std::memcpy(&out, &x, 4);
    } else {
        const uint32_t x = ((e + 112u) << 23) | (m << 13);
        std::memcpy(&out, &x, 4);
    }
    uint32_t x;
    std::memcpy(&x, &out, 4);
    x |= sign;
    std::memcpy(&out, &x, 4);
    return out;
}

uint16_t f32_to_bf16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    return (uint16_t)((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);
}

int choose_mw(int cout) {
    if (cout % 48 == 0) return 3;
    if (cout <= 16) return 1;
    if (cout <= 32) return 2;
    return 3;
}

}  // namespace

struct fastsvc_plan {
    fastsvc_config cfg;
    int n = 0;
    std::vector<int> down_scales;       // [1, s_{n-1}, ..., s_1]   (fastsvc.py:270-272)
    std::vector<DownStage> down;
    std::vector<UpStage> up;
    RawParam last;
    size_t blob_floats = 0;
    std::vector<std::pair<PackedConv*, PackSource>> pack_jobs;
    struct ChainJob { PackedConv* c; std::string first, second, in1; };     // c = the SECOND conv; layer names (in1: stage 0's 1 -> C conv)
    std::vector<ChainJob> chain_jobs;
    struct FilmChainJob { PackedConv* c; std::string conv[2]; PackSource heads; int C; };
    std::vector<FilmChainJob> film_chain_jobs;
    struct UpHeadJob { PackedConv* c; std::string first, res, up; int cin, C; };
    std::vector<UpHeadJob> up_head_jobs;
    struct XrJob { PackedConv* c; std::string conv, res; };                 // c = the block's d = 3 conv
    std::vector<XrJob> xr_jobs;
    std::vector<RawParam*> raw_jobs;
    std::vector<int> cond_bound_jobs;   // conditioning stages with cbnd tables
    double flops_per_sample = 0.0;
    int storage = 0;                    // activation storage in the workspace: 0 float32, 1 bfloat16
    bool compact = false;               // workspace layout: intermediates of different stages share buffers (fastsvc_plan_set_workspace_mode)
    // autotuned launch choices (fastsvc_autotune), keyed by "layer|B|T"; guarded by tune_mu
    struct Choice { int NW, WM, WN, tpw, algo; };    // algo: 0 direct / as launched, 1 Winograd F(2,3)
    mutable std::mutex tune_mu;
    mutable std::map<std::string, Choice> tuned;
    // Off-table shapes: the (NW, WM, WN, algorithm) of the table entry of the SAME layer whose problem size is
    // nearest (cached per key; algo < 0 = the table has nothing for the layer); cleared whenever `tuned` changes.
    mutable std::map<std::string, Choice> priors;
    mutable std::vector<double> pack_cost;      // per pack job: what the last fastsvc_pack_weights measured (longest first next time)

    // |ln work ratio| + a quarter of |ln row-length ratio|: what a launch shape trades off (workgroups against tiles
    // per workgroup against columns per tile) depends on B * T first and on the row length second
    bool prior_for(const char* layer, int B, int T, bool bf16, Choice& out) const {     // tune_mu held
        const size_t ln = std::strlen(layer);
        double best = 1e30;
        bool found = false;
        for (const auto& kv : tuned) {
            const std::string& k = kv.first;
            if (k.size() <= ln + 1 || k.compare(0, ln, layer) != 0 || k[ln] != '|') continue;
            int b = 0, t = 0;
            char tail = 0;
            const int n = std::sscanf(k.c_str() + ln + 1, "%d|%d|%c", &b, &t, &tail);
            if (n < 2 || b <= 0 || t <= 0 || (n == 3) != bf16) continue;
            const double d = std::fabs(std::log((double)b * t / ((double)B * T))) + 0.25 * std::fabs(std::log((double)t / T));
            if (d < best) { best = d; out = kv.second; found = true; }
        }
        return found;
    }

    size_t alloc(size_t nfloats) {
        size_t off = blob_floats;
        blob_floats += align_up(nfloats, 64);     // 256-byte granules
        return off;
    }

    void plan_conv(PackedConv& c, int cin, int cout, int ntaps, int dil) {
        c.cin = cin; c.cout = cout; c.ntaps = ntaps; c.dil = dil;
        c.MW = choose_mw(cout);
        const int cinp = (cin + 3) / 4 * 4;
        c.KC = cinp < 24 ? cinp : 24;
        c.nchunks = (cinp + c.KC - 1) / c.KC;
        c.Q = ntaps * c.nchunks * (c.KC / 4);
        c.ngroups = (cout + 16 * c.MW - 1) / (16 * c.MW);
        c.w_floats = (size_t)c.ngroups * c.Q * 64 * c.MW;
        c.b_floats = (size_t)c.ngroups * 16 * c.MW;
    }

    // allocate one or two (paired lft/sine) packed convs at a constant stride
    void add_conv(PackedConv* c, int npair, int cin, int cout, int ntaps, int dil,
                  const std::vector<PackSource>& src) {
        for (int i = 0; i < npair; ++i) plan_conv(c[i], cin, cout, ntaps, dil);
        for (int i = 0; i < npair; ++i) c[i].w_off = alloc(c[i].w_floats);
        for (int i = 0; i < npair; ++i) c[i].b_off = alloc(c[i].b_floats);
        for (int i = 0; i < npair; ++i) c[i].bnd_off = alloc(2);
        if (npair == 2) c[0].bnd_pair = (long)(c[1].bnd_off - c[0].bnd_off);
        for (int i = 0; i < npair; ++i) {
            if (ntaps == 3 && (dil == 1 || dil == 2 || dil == 4) && c[i].KC == 24 && c[i].MW >= 2) c[i].wino = true;
        }
        for (int i = 0; i < npair; ++i)
            if (c[i].wino) {
                c[i].Qw = 4 * c[i].nchunks * 6;
                c[i].ww_off = alloc((size_t)c[i].ngroups * c[i].Qw * 64 * c[i].MW);
            }
        if (npair == 2 && c[0].wino) c[0].ww_pair = (long)(c[1].ww_off - c[0].ww_off);
        for (int i = 0; i < npair; ++i)
            if (c[i].wino && c[i].MW == 3 && cout % 32 == 0) {
                c[i].wino2 = true;
                c[i].ww2_off = alloc((size_t)(cout / 32) * c[i].Qw * 64 * 2);
            }
        if (npair == 2 && c[0].wino2) c[0].ww2_pair = (long)(c[1].ww2_off - c[0].ww2_off);
        if (ntaps == 3 && c[0].MW >= 2) {
            for (int i = 0; i < npair; ++i) { c[i].hx = true; c[i].nch32 = (cin + 31) / 32; }
            for (int prec = 0; prec < 2; ++prec) {
                const size_t fl = (size_t)c[0].ngroups * c[0].nch32 * 3 * c[0].MW * (prec == 0 ? 2 : 1) * 256;   // 1 KB fragments
                for (int i = 0; i < npair; ++i) c[i].hx_off[prec] = alloc(fl);
                if (npair == 2) c[0].hx_pair[prec] = (long)(c[1].hx_off[prec] - c[0].hx_off[prec]) * 4;
            }
            for (int i = 0; i < npair; ++i) c[i].hx_inv_off = alloc(c[i].b_floats);
            if (npair == 2) c[0].hx_inv_pair = (long)(c[1].hx_inv_off - c[0].hx_inv_off);
        }
        for (int i = 0; i < npair; ++i) pack_jobs.emplace_back(&c[i], src[i]);
    }

    // the fused k=3 + 1x1 decimating pair of a down stage (lft / sine twins)
    void add_dec2(PackedConv* c, int cin, int cout, const std::string (&c1_layer)[2], const std::string (&r_layer)[2]) {
        for (int i = 0; i < 2; ++i) {
            plan_conv(c[i], cin, cout, 3, 1);
            if (c[i].KC != 24) return;                      // tiny generators keep the two generic launches
        }
        for (int i = 0; i < 2; ++i) {
            c[i].dec2 = true;
            c[i].Q = 24 * c[i].nchunks;
            c[i].w_floats = (size_t)c[i].ngroups * c[i].Q * 64 * c[i].MW;
        }
        for (int i = 0; i < 2; ++i) c[i].w_off = alloc(c[i].w_floats);
        for (int i = 0; i < 2; ++i) c[i].b_off = alloc(c[i].b_floats);
        for (int i = 0; i < 2; ++i) c[i].b2_off = alloc(c[i].b_floats);
        c[0].b2_pair = (long)(c[1].b2_off - c[0].b2_off);
        if (c[0].MW == 3) {                                  // four slots per chunk: w0 w1 w2 | w1x1
            for (int i = 0; i < 2; ++i) { c[i].hx = true; c[i].nch32 = (cin + 31) / 32; }
            for (int prec = 0; prec < 2; ++prec) {
                const size_t fl = (size_t)c[0].ngroups * c[0].nch32 * 4 * c[0].MW * (prec == 0 ? 2 : 1) * 256;
                for (int i = 0; i < 2; ++i) c[i].hx_off[prec] = alloc(fl);
                c[0].hx_pair[prec] = (long)(c[1].hx_off[prec] - c[0].hx_off[prec]) * 4;
            }
            for (int i = 0; i < 2; ++i) c[i].hx_inv_off = alloc(2 * c[i].b_floats);       // k=3 conv | 1x1 conv
            c[0].hx_inv_pair = (long)(c[1].hx_inv_off - c[0].hx_inv_off);
        }
        for (int i = 0; i < 2; ++i) {
            PackSource src;
            src.dec2 = true;
            src.pieces = {{c1_layer[i], 0, 0}, {r_layer[i], 0, 0}};
            pack_jobs.emplace_back(&c[i], src);
        }
    }

    // c2 -> c3 of a down stage as one launch: every workgroup keeps the whole C-channel intermediate tile in LDS,
    // so the stage must fit one workgroup row of channel groups (C <= 96) and the LDS (see hx_launch_shape)
    void add_chain(PackedConv* a, PackedConv* c, const std::string (&first)[2], const std::string (&second)[2],
                   const std::string* in1 = nullptr) {
        if (!a[0].hx || !c[0].hx || a[0].cout != c[0].cin || a[0].cin != c[0].cout || a[0].MW != c[0].MW ||
            c[0].ngroups > 2 || a[0].dil > 4 || c[0].dil > 4) return;
        for (int prec = 0; prec < 2; ++prec) {
            const size_t fl = (size_t)c[0].ngroups * (a[0].nch32 + c[0].nch32) * 3 * c[0].MW * (prec == 0 ? 2 : 1) * 256;
            for (int i = 0; i < 2; ++i) c[i].hxc_off[prec] = alloc(fl);
            c[0].hxc_pair[prec] = (long)(c[1].hxc_off[prec] - c[0].hxc_off[prec]) * 4;
        }
        for (int i = 0; i < 2; ++i) c[i].bmid_off = a[i].b_off;
        c[0].bmid_pair = (long)(a[1].b_off - a[0].b_off);
        for (int i = 0; i < 2; ++i) c[i].hxc_inv_off = alloc(2 * c[i].b_floats + 4);
        c[0].hxc_inv_pair = (long)(c[1].hxc_inv_off - c[0].hxc_inv_off);
        for (int i = 0; i < 2; ++i) chain_jobs.push_back(ChainJob{&c[i], first[i], second[i], in1 ? in1[i] : std::string()});
    }

    // the FiLM net of a stage as one launch: [lrelu(conv_lft(h_lft)) ; lrelu(conv_sine(h_sine))] is a block-diagonal
    // 2C -> 2C conv of the channel-concatenated input, the K-concatenated heads conv follows (fastsvc.py:120-131)
    void add_film_chain(PackedConv& c, const PackedConv& heads, int C, const std::string (&conv)[2], const PackSource& hsrc) {
        plan_conv(c, 2 * C, 2 * C, 3, 1);
        if (c.MW < 2 || c.ngroups > 2) return;
        c.hx = true; c.nch32 = (2 * C + 31) / 32;
        for (int prec = 0; prec < 2; ++prec)
            c.hxc_off[prec] = alloc((size_t)c.ngroups * 2 * c.nch32 * 3 * c.MW * (prec == 0 ? 2 : 1) * 256);
        c.b_off = heads.b_off;
        c.bmid_off = alloc(c.b_floats);
        c.hxc_inv_off = alloc(2 * c.b_floats + 4);
        film_chain_jobs.push_back(FilmChainJob{&c, {conv[0], conv[1]}, hsrc, C});
    }

    // the head of an up block as one launch (fastsvc_kernels.h, MODE_UPHEAD): per channel group
    // [conv_first's units | residual conv's polyphase units | up conv's], float32 storage (split-binary16) only
    void add_up_head(UpStage& u, const std::string& prefix) {
        PackedConv& c = u.head;
        if (!u.first.hx || !u.res.hx || !u.up.hx || !u.res.poly || !u.up.poly || u.first.MW != u.res.MW) return;
        c = u.res;                                     // geometry of the C -> C convs (MW, ngroups, biases of the residual conv)
        c.hx_off[0] = c.hx_off[1] = 0; c.hxp_off[0] = c.hxp_off[1] = 0; c.hx_inv_off = c.hxp_inv_off = 0;
        if (c.ngroups > 4) return;
        c.nch32 = u.first.nch32;                       // units of the first conv; the polyphase convs have ceil(C / 32) each
        const int nchb = (u.C + 31) / 32;
        c.hxc_off[0] = alloc((size_t)c.ngroups * (c.nch32 + 2 * nchb) * 3 * c.MW * 2 * 256);
        c.hxc_inv_off = alloc(3 * c.b_floats + 4);
        c.bmid_off = u.first.b_off;
        c.b2_off = u.up.b_off;
        up_head_jobs.push_back(UpHeadJob{&c, prefix + ".conv_first", prefix + ".residual_block.1", prefix + ".upsample_block0.2",
                                         u.Cin, u.C});
    }

    // the middle conv of an up block with the block's stretched residual conv folded in (ConvParams::x2): per channel
    // group the units [d3 chunk 0 | residual chunk 0 | d3 chunk 1 | ...], inverse scale tables [d3 | residual]
    void add_xr_fuse(UpStage& u, const std::string& prefix) {
        PackedConv& c = u.d3;
        if (!c.hx || !u.res.hx || c.cin != c.cout || u.res.cin != c.cin || u.res.cout != c.cout || u.res.MW != c.MW) return;
        for (int prec = 0; prec < 2; ++prec)
            c.hxc_off[prec] = alloc((size_t)c.ngroups * 2 * c.nch32 * 3 * c.MW * (prec == 0 ? 2 : 1) * 256);
        c.hxc_inv_off = alloc(2 * c.b_floats);
        c.b2_off = u.res.b_off;
        xr_jobs.push_back(XrJob{&c, prefix + ".conv_block1.1", prefix + ".residual_block.1"});
    }

    void add_raw(RawParam* r, int npair, const std::vector<std::string>& layers, size_t wf, size_t bf) {
        for (int i = 0; i < npair; ++i) { r[i].layer = layers[i]; r[i].w_floats = wf; r[i].b_floats = bf; }
        for (int i = 0; i < npair; ++i) r[i].w_off = alloc(wf);
        for (int i = 0; i < npair; ++i) r[i].b_off = alloc(bf);
        for (int i = 0; i < npair; ++i) r[i].bnd_off = alloc(2);
        for (int i = 0; i < npair; ++i) raw_jobs.push_back(&r[i]);
    }
};

namespace {

const char* SIG_NAME[2] = {"lft", "sine"};

PackSource single(const std::string& layer) {
    PackSource s;
    s.pieces.push_back({layer, 0, 0});
    return s;
}

int build_plan(fastsvc_plan& P) {
    const fastsvc_config& c = P.cfg;
    const int n = c.n_stages;
    P.n = n;
    P.down_scales.assign(n, 1);
    for (int k = 1; k < n; ++k) P.down_scales[k] = c.upsampling_scales[n - k];
    P.down.resize(n);
    P.up.resize(n);
    double flops = 0.0;          // per output sample
    double rate = 1.0;           // columns per output sample at the current stage

    // ---- conditioning chains (both signals), stage k runs at T / prod(down_scales[0..k]) ----
    int cin = 1;
    for (int k = 0; k < n; ++k) {
        DownStage& d = P.down[k];
        d.C = c.mid_channels[n - 1 - k];
        d.Cin = cin;
        d.scale = P.down_scales[k];
        rate /= d.scale;
        const std::string pl = "downsampling_lft." + std::to_string(k);
        const std::string ps = "downsampling_sine." + std::to_string(k);
        if (k == 0) {
            P.add_raw(d.r_raw, 2, {pl + ".residual_block.0", ps + ".residual_block.0"}, (size_t)d.C * cin, d.C);
            P.add_raw(d.c1_raw, 2, {pl + ".downsample_block.2", ps + ".downsample_block.2"}, (size_t)d.C * cin * 3, d.C);
        } else {
            P.add_conv(d.r, 2, cin, d.C, 1, 1, {single(pl + ".residual_block.0"), single(ps + ".residual_block.0")});
            P.add_conv(d.c1, 2, cin, d.C, 3, 1, {single(pl + ".downsample_block.2"), single(ps + ".downsample_block.2")});
            const std::string c1_layers[2] = {pl + ".downsample_block.2", ps + ".downsample_block.2"};
            const std::string r_layers[2] = {pl + ".residual_block.0", ps + ".residual_block.0"};
            P.add_dec2(d.rc1, cin, d.C, c1_layers, r_layers);
        }
        P.add_conv(d.c2, 2, d.C, d.C, 3, 2, {single(pl + ".downsample_block.4"), single(ps + ".downsample_block.4")});
        P.add_conv(d.c3, 2, d.C, d.C, 3, 4, {single(pl + ".downsample_block.6"), single(ps + ".downsample_block.6")});
        {
            const std::string first[2] = {pl + ".downsample_block.4", ps + ".downsample_block.4"};
            const std::string second[2] = {pl + ".downsample_block.6", ps + ".downsample_block.6"};
            const std::string in1[2] = {pl + ".downsample_block.2", ps + ".downsample_block.2"};
            P.add_chain(d.c2, d.c3, first, second, k == 0 ? in1 : nullptr);
        }
        const std::string fl = "film_lft." + std::to_string(k);
        const std::string fs = "film_sine." + std::to_string(k);
        P.add_conv(d.film, 2, d.C, d.C, 3, 1, {single(fl + ".conv"), single(fs + ".conv")});
        PackSource heads;
        heads.pieces = {{fl + ".conv_scale", 0, 0}, {fs + ".conv_scale", 0, d.C},
                        {fl + ".conv_shift", d.C, 0}, {fs + ".conv_shift", d.C, d.C}};
        P.add_conv(&d.heads, 1, 2 * d.C, 2 * d.C, 3, 1, {heads});
        {
            const std::string convs[2] = {fl + ".conv", fs + ".conv"};
            P.add_film_chain(d.filmc, d.heads, d.C, convs, heads);
        }
        if (k < 2) {
            for (int i = 0; i < 2; ++i) d.cbnd_off[i] = P.alloc((size_t)4 * 2 * d.C);
            P.cond_bound_jobs.push_back(k);
        }
        // 2*MAC per column: 1x1 + k3 first + two k3 (C->C) + three FiLM k3 convs, two signals
        const double per_col = 2.0 * ((double)cin * d.C + 3.0 * cin * d.C + 2.0 * 3.0 * d.C * d.C + 3.0 * 3.0 * d.C * d.C);
        flops += 2.0 * per_col * rate;
        cin = d.C;
    }

    // ---- up blocks ----
    cin = c.in_channels;
    double in_rate = 1.0;
    for (int i = 0; i < n; ++i) in_rate /= c.upsampling_scales[i];   // ppg frames per sample
    for (int i = 0; i < n; ++i) {
        UpStage& u = P.up[i];
        u.C = c.mid_channels[i];
        u.Cin = cin;
        u.scale = c.upsampling_scales[i];
        const std::string p = "upsampling_nets." + std::to_string(i);
        P.add_conv(&u.first, 1, cin, u.C, 3, 1, {single(p + ".conv_first")});
        P.add_conv(&u.res, 1, u.C, u.C, 3, 1, {single(p + ".residual_block.1")});
        P.add_conv(&u.up, 1, u.C, u.C, 3, 1, {single(p + ".upsample_block0.2")});
        for (PackedConv* pc : {&u.res, &u.up}) {               // the two convs behind the stretch
            if (pc->KC == 24 && (u.scale == 2 || u.scale == 4 || u.scale == 5)) {
                pc->poly = true;
                pc->wp_off = P.alloc(pc->w_floats);
                if (pc->hx) {
                    for (int prec = 0; prec < 2; ++prec)
                        pc->hxp_off[prec] = P.alloc((size_t)pc->ngroups * pc->nch32 * 3 * pc->MW * (prec == 0 ? 2 : 1) * 256);
                    pc->hxp_inv_off = P.alloc(pc->b_floats);
                }
            }
        }
        P.add_up_head(u, p);
        P.add_conv(&u.d3, 1, u.C, u.C, 3, 3, {single(p + ".conv_block1.1")});
        P.add_xr_fuse(u, p);
        P.add_conv(&u.d9, 1, u.C, u.C, 3, 9, {single(p + ".conv_block2.1")});
        P.add_conv(&u.d27, 1, u.C, u.C, 3, 27, {single(p + ".conv_block3.1")});
        if (c.use_spk_emb) P.add_raw(&u.emb, 1, {p + ".emb_projector"}, (size_t)u.C * c.spk_emb_size, u.C);
        const double out_rate = in_rate * u.scale;
        flops += 2.0 * 3.0 * cin * u.C * in_rate + 5.0 * 2.0 * 3.0 * u.C * u.C * out_rate;
        in_rate = out_rate;
        cin = u.C;
    }
    P.add_raw(&P.last, 1, {"conv_last"}, (size_t)c.out_channels * cin, c.out_channels);
    flops += 2.0 * cin * c.out_channels;
    P.flops_per_sample = flops;
    return FASTSVC_OK;
}

// ------------------------------------------------------------------------------------------
// weight lookup / fold
// ------------------------------------------------------------------------------------------
struct HostLayer { std::vector<float> w; std::vector<float> b; };

int fetch_layer(const std::unordered_map<std::string, const fastsvc_tensor*>& sd, const std::string& layer,
                int cout, size_t per_out, HostLayer& out) {
    auto bi = sd.find(layer + ".bias");
    if (bi == sd.end() || bi->second->numel != cout)
        return fail(FASTSVC_E_MISSING, "missing or mis-sized tensor: " + layer + ".bias");
    out.b.assign(bi->second->data, bi->second->data + cout);
    const int64_t wn = (int64_t)cout * (int64_t)per_out;
    auto wi = sd.find(layer + ".weight");
    if (wi != sd.end()) {
        if (wi->second->numel != wn) return fail(FASTSVC_E_MISSING, "mis-sized tensor: " + layer + ".weight");
        out.w.assign(wi->second->data, wi->second->data + wn);
        return FASTSVC_OK;
    }
    auto gi = sd.find(layer + ".weight_g");
    auto vi = sd.find(layer + ".weight_v");
    if (gi == sd.end() || vi == sd.end())
        return fail(FASTSVC_E_MISSING, "missing tensor: " + layer + ".weight (or .weight_g/.weight_v)");
    if (gi->second->numel != cout || vi->second->numel != wn)
        return fail(FASTSVC_E_MISSING, "mis-sized tensor: " + layer + ".weight_g/.weight_v");
    // legacy torch.nn.utils.weight_norm, dim=0: w = v * (g / ||v||), norm over everything but dim 0
    out.w.resize(wn);
    const float* v = vi->second->data;
    const float* g = gi->second->data;
    for (int co = 0; co < cout; ++co) {
        float ss = 0.f;
        for (size_t i = 0; i < per_out; ++i) ss += v[co * per_out + i] * v[co * per_out + i];
        const float sc = g[co] / std::sqrt(ss);
        for (size_t i = 0; i < per_out; ++i) out.w[co * per_out + i] = v[co * per_out + i] * sc;
    }
    return FASTSVC_OK;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int fastsvc_abi_version(void) { return FASTSVC_ABI_VERSION; }
const char* fastsvc_last_error(void) { return g_err.c_str(); }

int fastsvc_plan_create(const fastsvc_config* cfg, fastsvc_plan** out_plan) {
    if (!cfg || !out_plan) return fail(FASTSVC_E_INVALID, "null argument");
    if (cfg->n_stages < 1 || cfg->n_stages > FASTSVC_MAX_STAGES)
        return fail(FASTSVC_E_INVALID, "n_stages out of range");
    if (cfg->in_channels < 1 || cfg->out_channels < 1)
        return fail(FASTSVC_E_INVALID, "channel counts must be positive");
    for (int i = 0; i < cfg->n_stages; ++i)
        if (cfg->mid_channels[i] < 1 || cfg->upsampling_scales[i] < 1)
            return fail(FASTSVC_E_INVALID, "mid_channels / upsampling_scales must be positive");
    if (cfg->use_spk_emb && cfg->spk_emb_size < 1) return fail(FASTSVC_E_INVALID, "spk_emb_size must be positive");
    fastsvc_plan* P = new fastsvc_plan();
    P->cfg = *cfg;
    const int rc = build_plan(*P);
    if (rc != FASTSVC_OK) { delete P; return rc; }
    *out_plan = P;
    return FASTSVC_OK;
}

void fastsvc_plan_destroy(fastsvc_plan* plan) { delete plan; }

size_t fastsvc_weight_blob_bytes(const fastsvc_plan* plan) { return plan ? plan->blob_floats * sizeof(float) : 0; }

double fastsvc_flops_per_sample(const fastsvc_plan* plan) { return plan ? plan->flops_per_sample : 0.0; }

int fastsvc_pack_weights(const fastsvc_plan* plan, const fastsvc_tensor* tensors, int32_t n_tensors,
                         void* host_blob) {
    if (!plan || !tensors || !host_blob) return fail(FASTSVC_E_INVALID, "null argument");
    std::unordered_map<std::string, const fastsvc_tensor*> sd;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) return fail(FASTSVC_E_INVALID, "tensor entry with null name/data");
        sd[tensors[i].name] = &tensors[i];
    }
    float* blob = static_cast<float*>(host_blob);
    {
        // zero the blob (padding granules, absent formats) - in parallel: one thread takes 3-4 ms for the 54 MB, a
        // third of what the whole pack takes on 16 threads
        const size_t bytes = plan->blob_floats * sizeof(float);
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nz = (unsigned)std::min<size_t>(std::min(16u, hw), std::max<size_t>(1, bytes >> 22));
        std::vector<std::thread> zs;
        auto zero = [&](unsigned k) {
            const size_t a = (bytes / nz * k) & ~(size_t)63, b = k + 1 == nz ? bytes : (bytes / nz * (k + 1)) & ~(size_t)63;
            std::memset(reinterpret_cast<char*>(blob) + a, 0, b - a);
        };
        for (unsigned k = 1; k < nz; ++k) zs.emplace_back(zero, k);
        zero(0);
        for (auto& th : zs) th.join();
    }

    // split-half / bf16 fragments (fastsvc_hx.hip): lane l of fragment (group, chunk, slot, tile m) holds
    // Wt[co = (group*MW + m)*16 + (l & 15)][ci = chunk*32 + 8*(l >> 4) + e][slot], e = 0..7
    // binary16 pieces (prec 0): every output channel's weights are first multiplied by the power of two that puts
    // the channel's largest magnitude into [2^14, 2^15) - exact, and it keeps the hi + lo split at 22 bits for every
    // weight within 2^-17 of that maximum whatever the channel's absolute scale (weight_g) is; the inverse factors go
    // to inv[table * n16 + co] (ConvParams::whx_inv; the epilogue's bias FMA applies them).  `table_of(ci, slot)`
    // says which output a fragment feeds (MODE_DEC2: slot 3 = the 1x1 conv; fused pair: the second conv's units).
    // (generic over the accessors: through std::function the per-element calls made this the longest part of a pack)
    struct OneTable { int operator()(int, int) const { return 0; } };
    auto pack_hx = [&](const PackedConv& c, const size_t (&off)[2], int nslots, auto&& wt, size_t inv_off, int ntables,
                       auto&& table_of_, int precs = 3 /* bit 0: binary16 pieces (+ the scale tables), bit 1: bfloat16 */,
                       int gpart = 0, int gparts = 1 /* this call's share of the channel groups */) {
        const int g0 = c.ngroups * gpart / gparts, g1 = c.ngroups * (gpart + 1) / gparts;
        constexpr bool has_tables = !std::is_same<std::decay_t<decltype(table_of_)>, OneTable>::value;
        auto table_of = [&](int ci, int slot) -> int { return table_of_(ci, slot); };
        const int n16 = c.ngroups * 16 * c.MW;
        std::vector<int> ex((size_t)ntables * n16, 0);
        if (inv_off && (precs & 1)) {
            float* inv = blob + inv_off;
            for (int t = 0; t < ntables; ++t)
                for (int co = g0 * 16 * c.MW; co < g1 * 16 * c.MW; ++co) {
                    float m = 0.f;
                    if (co < c.cout)
                        for (int ci = 0; ci < c.cin; ++ci)
                            for (int slot = 0; slot < nslots; ++slot)
                                if (!has_tables || table_of(ci, slot) == t) m = std::max(m, std::fabs((float)wt(co, ci, slot)));
                    int e = 0;
                    if (m > 0.f && std::isfinite(m)) e = std::min(60, std::max(-60, 14 - std::ilogb(m)));
                    ex[(size_t)t * n16 + co] = e;
                    inv[(size_t)t * n16 + co] = std::ldexp(1.0f, -e);
                }
        }
        for (int prec = 0; prec < 2; ++prec) {
            if (!off[prec] || !(precs & (1 << prec))) continue;   // (a fragment set that exists for one storage type only)
            const int np = prec == 0 ? 2 : 1;
            uint16_t* hp = reinterpret_cast<uint16_t*>(blob + off[prec]);
            for (int grp = g0; grp < g1; ++grp)
                for (int ch = 0; ch < c.nch32; ++ch)
                    for (int slot = 0; slot < nslots; ++slot)
                        for (int m = 0; m < c.MW; ++m) {
                            uint16_t* frag = hp + ((((size_t)grp * c.nch32 + ch) * nslots + slot) * c.MW + m) * np * 512;
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 8; ++e) {
                                    const int co = (grp * c.MW + m) * 16 + (lane & 15);
                                    const int ci = ch * 32 + 8 * (lane >> 4) + e;
                                    float v = (co < c.cout && ci < c.cin) ? (float)wt(co, ci, slot) : 0.f;
                                    if (prec == 0) {
                                        if (inv_off) v = std::ldexp(v, ex[(size_t)(has_tables ? table_of(ci, slot) : 0) * n16 + co]);
                                        const uint16_t hi = f32_to_f16(v);
                                        frag[lane * 8 + e] = hi;
                                        frag[512 + lane * 8 + e] = f32_to_f16(v - f16_to_f32(hi));
                                    } else {
                                        frag[lane * 8 + e] = f32_to_bf16(v);
                                    }
                                }
                        }
        }
    };
    // Every job writes its own region of the blob: they run on a small thread pool (the fold + pack of the whole
    // generator is ~55 ms of single-thread work - paid by every training step, whose optimizer update invalidates the
    // packed copy - and ~7 ms on the pool: 11-13 ms before the big layers' jobs were cut up, ordered longest first
    // and given inlined accessors).  The first failing job's code and message are reported.
    std::vector<std::function<int()>> tasks;
    std::vector<std::string> task_names;                 // (FASTSVC_PACK_DEBUG)
    // A conv's formats (f32 fragments, split-binary16 / bf16 fragments, Winograd, polyphase) are independent regions
    // of the blob: the big layers' jobs are cut into parts that each build the dense matrix and write a subset of the
    // formats - the 2C -> 2C FiLM heads of the C = 192 stage alone took 7.9 ms of a 11 ms pack as one job
    for (const auto& job_ : plan->pack_jobs) {
      const size_t wcount = (size_t)job_.first->cout * job_.first->cin * job_.first->ntaps;
      const int nparts = job_.second.dec2 ? 1 : wcount > 300000 ? 6 : wcount > 60000 ? 2 : 1;
      for (int part = 0; part < nparts; ++part) {
      task_names.push_back(job_.second.pieces.empty() ? std::string("conv") : job_.second.pieces[0].layer + (nparts > 1 ? "#" + std::to_string(part) : ""));
      tasks.emplace_back([&, pj = &job_, part, nparts]() -> int {
        const auto& job = *pj;
        const PackedConv& c = *job.first;
        const PackSource& src = job.second;
        // sections: 0 / 1 binary16 pieces of the first / second half of the channel groups, 2 bfloat16 pieces,
        // 3 f32 fragments + bias + bounds, 4 Winograd, 5 Winograd (32-channel grouping), 6 / 7 polyphase pieces / fragments
        static const int to2[8] = {0, 0, 1, 1, 0, 1, 0, 1}, to6[8] = {0, 1, 2, 3, 4, 5, 4, 5};
        auto mine = [&](int section) { return (nparts == 1 ? 0 : nparts == 2 ? to2[section] : to6[section]) == part; };
        if (src.dec2) {
            // MODE_DEC2: components w0 w1 w2 (k=3 conv) | w1x1, step order inside a chunk
            // (half * 4 + component) * 3 + k-group-in-half (mfma_unit_dec2)
            HostLayer L3, L1;
            int rc = fetch_layer(sd, src.pieces[0].layer, c.cout, (size_t)c.cin * 3, L3);
            if (rc != FASTSVC_OK) return rc;
            rc = fetch_layer(sd, src.pieces[1].layer, c.cout, (size_t)c.cin, L1);
            if (rc != FASTSVC_OK) return rc;
            float* wp = blob + c.w_off;
            for (int grp = 0; grp < c.ngroups; ++grp)
                for (int ch = 0; ch < c.nchunks; ++ch)
                    for (int h = 0; h < 2; ++h)
                        for (int comp = 0; comp < 4; ++comp)
                            for (int jj = 0; jj < 3; ++jj) {
                                const int q = ch * 24 + (h * 4 + comp) * 3 + jj;
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int ci = ch * c.KC + 4 * (3 * h + jj) + (lane >> 4);
                                    for (int m = 0; m < c.MW; ++m) {
                                        const int co = (grp * c.MW + m) * 16 + (lane & 15);
                                        float v = 0.f;
                                        if (co < c.cout && ci < c.cin)
                                            v = comp < 3 ? L3.w[((size_t)co * c.cin + ci) * 3 + comp] : L1.w[(size_t)co * c.cin + ci];
                                        wp[(((size_t)grp * c.Q + q) * 64 + lane) * c.MW + m] = v;
                                    }
                                }
                            }
            for (int co = 0; co < c.cout; ++co) { blob[c.b_off + co] = L3.b[co]; blob[c.b2_off + co] = L1.b[co]; }
            if (c.hx)
                pack_hx(c, c.hx_off, 4, [&](int co, int ci, int slot) {
                    return slot < 3 ? L3.w[((size_t)co * c.cin + ci) * 3 + slot] : L1.w[(size_t)co * c.cin + ci];
                }, c.hx_inv_off, 2, [](int, int slot) { return slot == 3 ? 1 : 0; });
            return FASTSVC_OK;
        }
        // virtual dense weight W[co][ci][tap] and bias
        std::vector<float> W((size_t)c.cout * c.cin * c.ntaps, 0.f);
        std::vector<float> bias(c.cout, 0.f);
        const int npieces = (int)src.pieces.size();
        // piece geometry: heads use 4 pieces of (C x C); single uses the whole matrix
        const int pc_out = npieces == 1 ? c.cout : c.cout / 2;
        const int pc_in = npieces == 1 ? c.cin : c.cin / 2;
        for (const auto& pc : src.pieces) {
            HostLayer L;
            const int rc = fetch_layer(sd, pc.layer, pc_out, (size_t)pc_in * c.ntaps, L);
            if (rc != FASTSVC_OK) return rc;
            for (int co = 0; co < pc_out; ++co) {
                for (int ci = 0; ci < pc_in; ++ci)
                    for (int t = 0; t < c.ntaps; ++t)
                        W[((size_t)(co + pc.co_off) * c.cin + ci + pc.ci_off) * c.ntaps + t] =
                            L.w[((size_t)co * pc_in + ci) * c.ntaps + t];
                bias[co + pc.co_off] += L.b[co];     // lft + sine biases add (fastsvc.py:129-130)
            }
        }
        // fragment order: [group][q][lane][m],  q = (chunk * ntaps + tap) * (KC/4) + g
        //   value = W[co = (group*MW + m)*16 + (lane & 15)][ci = chunk*KC + 4g + (lane >> 4)][tap]
        auto pack_fragments = [&](const std::vector<float>& Wt, float* wp, int ntaps) {
            const int kg = c.KC / 4;
            const int Q = ntaps * c.nchunks * kg;
            for (int grp = 0; grp < c.ngroups; ++grp)
                for (int ch = 0; ch < c.nchunks; ++ch)
                    for (int tap = 0; tap < ntaps; ++tap)
                        for (int g = 0; g < kg; ++g) {
                            const int q = (ch * ntaps + tap) * kg + g;
                            for (int lane = 0; lane < 64; ++lane) {
                                const int ci = ch * c.KC + 4 * g + (lane >> 4);
                                for (int m = 0; m < c.MW; ++m) {
                                    const int co = (grp * c.MW + m) * 16 + (lane & 15);
                                    float v = 0.f;
                                    if (co < c.cout && ci < c.cin) v = Wt[((size_t)co * c.cin + ci) * ntaps + tap];
                                    wp[(((size_t)grp * Q + q) * 64 + lane) * c.MW + m] = v;
                                }
                            }
                        }
        };
        if (mine(3)) pack_fragments(W, blob + c.w_off, c.ntaps);
        auto wt3 = [&](int co, int ci, int tap) { return W[((size_t)co * c.cin + ci) * 3 + tap]; };
        if (c.hx && mine(0) && mine(1)) pack_hx(c, c.hx_off, 3, wt3, c.hx_inv_off, 1, OneTable{}, 1);
        else if (c.hx && mine(0)) pack_hx(c, c.hx_off, 3, wt3, c.hx_inv_off, 1, OneTable{}, 1, 0, 2);
        else if (c.hx && mine(1)) pack_hx(c, c.hx_off, 3, wt3, c.hx_inv_off, 1, OneTable{}, 1, 1, 2);
        if (c.hx && mine(2)) pack_hx(c, c.hx_off, 3, wt3, c.hx_inv_off, 1, OneTable{}, 2);
        if (c.wino && (mine(4) || mine(5))) {
            // Winograd F(2,3) weight transform G w (fastsvc_kernels.h, MODE_WINO), in f64
            std::vector<float> Ww((size_t)c.cout * c.cin * 4);
            for (size_t i = 0; i < (size_t)c.cout * c.cin; ++i) {
                const double w0 = W[3 * i], w1 = W[3 * i + 1], w2 = W[3 * i + 2];
                Ww[4 * i] = (float)w0;
                Ww[4 * i + 1] = (float)(0.5 * (w0 + w1 + w2));
                Ww[4 * i + 2] = (float)(0.5 * (w0 - w1 + w2));
                Ww[4 * i + 3] = (float)w2;
            }
            // step order inside a chunk: (half * 4 + component) * 3 + k-group-in-half (mfma_unit_wino)
            auto pack_wino = [&](float* wp, int MWp) {
                const int ngrp = (c.cout + 16 * MWp - 1) / (16 * MWp);
                for (int grp = 0; grp < ngrp; ++grp)
                    for (int ch = 0; ch < c.nchunks; ++ch)
                        for (int h = 0; h < 2; ++h)
                            for (int comp = 0; comp < 4; ++comp)
                                for (int jj = 0; jj < 3; ++jj) {
                                    const int q = ch * 24 + (h * 4 + comp) * 3 + jj;
                                    for (int lane = 0; lane < 64; ++lane) {
                                        const int ci = ch * c.KC + 4 * (3 * h + jj) + (lane >> 4);
                                        for (int m = 0; m < MWp; ++m) {
                                            const int co = (grp * MWp + m) * 16 + (lane & 15);
                                            float v = 0.f;
                                            if (co < c.cout && ci < c.cin) v = Ww[((size_t)co * c.cin + ci) * 4 + comp];
                                            wp[(((size_t)grp * c.Qw + q) * 64 + lane) * MWp + m] = v;
                                        }
                                    }
                                }
            };
            if (mine(4)) pack_wino(blob + c.ww_off, c.MW);
            if (c.wino2 && mine(5)) pack_wino(blob + c.ww2_off, 2);
        }
        if (c.poly && (mine(6) || mine(7))) {
            // polyphase taps (fastsvc_kernels.h, MODE_POLY): slot 0 = W0, slot 1 = W0+W1+W2, slot 2 = W2
            std::vector<float> Wp(W.size());
            for (size_t i = 0; i + 2 < W.size(); i += 3) {
                Wp[i] = W[i];
                Wp[i + 1] = (float)((double)W[i] + (double)W[i + 1] + (double)W[i + 2]);
                Wp[i + 2] = W[i + 2];
            }
            if (mine(7)) pack_fragments(Wp, blob + c.wp_off, c.ntaps);
            if (c.hx && mine(6)) pack_hx(c, c.hxp_off, 3, [&](int co, int ci, int tap) { return Wp[((size_t)co * c.cin + ci) * 3 + tap]; }, c.hxp_inv_off, 1, OneTable{});
        }
        float* bp = blob + c.b_off;
        if (mine(3)) for (int co = 0; co < c.cout; ++co) bp[co] = bias[co];
        if (mine(3)) {
            float l1 = 0.f, bmax = 0.f;
            const size_t per = (size_t)c.cin * c.ntaps;
            for (int co = 0; co < c.cout; ++co) {
                double sum = 0.0;
                for (size_t i = 0; i < per; ++i) sum += std::fabs((double)W[co * per + i]);
                l1 = std::max(l1, (float)sum);
                bmax = std::max(bmax, std::fabs(bias[co]));
            }
            blob[c.bnd_off] = l1; blob[c.bnd_off + 1] = bmax;
        }
        return FASTSVC_OK;
      });
      }
    }
    task_names.insert(task_names.end(), plan->chain_jobs.size(), "chain");
    for (const auto& job_ : plan->chain_jobs) tasks.emplace_back([&, pj = &job_]() -> int {
        const auto& job = *pj;
        const PackedConv& c = *job.c;                       // second conv; the first maps cout -> cin of it (square)
        HostLayer LA, LB;
        int rc = fetch_layer(sd, job.first, c.cin, (size_t)c.cout * 3, LA);
        if (rc != FASTSVC_OK) return rc;
        rc = fetch_layer(sd, job.second, c.cout, (size_t)c.cin * 3, LB);
        if (rc != FASTSVC_OK) return rc;
        // same fragment format as pack_hx with 2 * nch32 units per group: the first conv's, then the second's
        PackedConv v = c;
        v.nch32 = 2 * c.nch32;
        v.cin = 2 * c.nch32 * 32;                           // the accessor below bounds the real channels
        const int cin = c.cin, nch = c.nch32;
        pack_hx(v, c.hxc_off, 3, [&](int co, int ci, int tap) {
            const bool second = ci >= nch * 32;
            const int cj = second ? ci - nch * 32 : ci;
            if (cj >= cin) return 0.f;
            return (second ? LB.w : LA.w)[((size_t)co * cin + cj) * 3 + tap];
        }, c.hxc_inv_off, 2, [nch](int ci, int) { return ci >= nch * 32 ? 1 : 0; });
        // bounds of the tensors that never leave LDS: |first conv's output| <= amax_in * l1 + bmax
        float* cst = blob + c.hxc_inv_off + 2 * (size_t)c.ngroups * 16 * c.MW;
        auto l1_of = [](const HostLayer& L, int rows, size_t per_row, float& l1, float& bmax) {
            l1 = 0.f; bmax = 0.f;
            for (int r = 0; r < rows; ++r) {
                double sum = 0.0;
                for (size_t i = 0; i < per_row; ++i) sum += std::fabs((double)L.w[r * per_row + i]);
                l1 = std::max(l1, (float)sum);
                bmax = std::max(bmax, std::fabs(L.b[r]));
            }
        };
        l1_of(LA, c.cin, (size_t)c.cout * 3, cst[0], cst[1]);
        cst[2] = 0.f; cst[3] = 0.f;
        if (!job.in1.empty()) {
            HostLayer L1;
            rc = fetch_layer(sd, job.in1, c.cout, 3, L1);        // stage 0's first conv: 1 -> C, k = 3
            if (rc != FASTSVC_OK) return rc;
            l1_of(L1, c.cout, 3, cst[2], cst[3]);
        }
        return FASTSVC_OK;
    });
    task_names.insert(task_names.end(), plan->xr_jobs.size(), "xr_fuse");
    for (const auto& job_ : plan->xr_jobs) tasks.emplace_back([&, pj = &job_]() -> int {
        const auto& job = *pj;
        const PackedConv& c = *job.c;
        HostLayer LA, LB;
        int rc = fetch_layer(sd, job.conv, c.cout, (size_t)c.cin * 3, LA);
        if (rc != FASTSVC_OK) return rc;
        rc = fetch_layer(sd, job.res, c.cout, (size_t)c.cin * 3, LB);
        if (rc != FASTSVC_OK) return rc;
        // pack_hx's fragment format with 2 * nch32 units per group, the two convs' 32-channel chunks interleaved
        PackedConv v = c;
        v.nch32 = 2 * c.nch32;
        v.cin = 2 * c.nch32 * 32;                           // the accessor below bounds the real channels
        const int cin = c.cin;
        pack_hx(v, c.hxc_off, 3, [&](int co, int ci, int tap) {
            const int chv = ci >> 5;
            const int cj = (chv >> 1) * 32 + (ci & 31);
            if (cj >= cin) return 0.f;
            return ((chv & 1) ? LB.w : LA.w)[((size_t)co * cin + cj) * 3 + tap];
        }, c.hxc_inv_off, 2, [](int ci, int) { return (ci >> 5) & 1; });
        return FASTSVC_OK;
    });
    task_names.insert(task_names.end(), plan->film_chain_jobs.size(), "film_chain");
    for (const auto& job_ : plan->film_chain_jobs) tasks.emplace_back([&, pj = &job_]() -> int {
        const auto& job = *pj;
        const PackedConv& c = *job.c;
        const int C = job.C, C2 = 2 * C;
        std::vector<float> WA((size_t)C2 * C2 * 3, 0.f), WB((size_t)C2 * C2 * 3, 0.f);
        for (int sgn = 0; sgn < 2; ++sgn) {                 // block-diagonal first conv, biases side by side
            HostLayer L;
            const int rc = fetch_layer(sd, job.conv[sgn], C, (size_t)C * 3, L);
            if (rc != FASTSVC_OK) return rc;
            for (int co = 0; co < C; ++co) {
                for (int ci = 0; ci < C; ++ci)
                    for (int t = 0; t < 3; ++t)
                        WA[((size_t)(co + sgn * C) * C2 + ci + sgn * C) * 3 + t] = L.w[((size_t)co * C + ci) * 3 + t];
                blob[c.bmid_off + co + sgn * C] = L.b[co];
            }
        }
        for (const auto& pc : job.heads.pieces) {            // the heads' virtual weight, as packed for the separate launch
            HostLayer L;
            const int rc = fetch_layer(sd, pc.layer, C, (size_t)C * 3, L);
            if (rc != FASTSVC_OK) return rc;
            for (int co = 0; co < C; ++co)
                for (int ci = 0; ci < C; ++ci)
                    for (int t = 0; t < 3; ++t)
                        WB[((size_t)(co + pc.co_off) * C2 + ci + pc.ci_off) * 3 + t] = L.w[((size_t)co * C + ci) * 3 + t];
        }
        PackedConv v = c;
        v.nch32 = 2 * c.nch32;
        v.cin = 2 * c.nch32 * 32;
        const int nch = c.nch32;
        pack_hx(v, c.hxc_off, 3, [&](int co, int ci, int tap) {
            const bool second = ci >= nch * 32;
            const int cj = second ? ci - nch * 32 : ci;
            if (cj >= C2) return 0.f;
            return (second ? WB : WA)[((size_t)co * C2 + cj) * 3 + tap];
        }, c.hxc_inv_off, 2, [nch](int ci, int) { return ci >= nch * 32 ? 1 : 0; });
        float* cst = blob + c.hxc_inv_off + 2 * (size_t)c.ngroups * 16 * c.MW;
        float l1 = 0.f, bmax = 0.f;
        for (int co = 0; co < C2; ++co) {
            double sum = 0.0;
            for (size_t i = 0; i < (size_t)C2 * 3; ++i) sum += std::fabs((double)WA[(size_t)co * C2 * 3 + i]);
            l1 = std::max(l1, (float)sum);
            bmax = std::max(bmax, std::fabs(blob[c.bmid_off + co]));
        }
        cst[0] = l1; cst[1] = bmax; cst[2] = 0.f; cst[3] = 0.f;
        return FASTSVC_OK;
    });
    task_names.insert(task_names.end(), plan->up_head_jobs.size(), "up_head");
    for (const auto& job_ : plan->up_head_jobs) tasks.emplace_back([&, pj = &job_]() -> int {
        const auto& job = *pj;
        const PackedConv& c = *job.c;
        const int cin = job.cin, C = job.C;
        HostLayer LF, LR, LU;
        int rc = fetch_layer(sd, job.first, C, (size_t)cin * 3, LF);
        if (rc != FASTSVC_OK) return rc;
        rc = fetch_layer(sd, job.res, C, (size_t)C * 3, LR);
        if (rc != FASTSVC_OK) return rc;
        rc = fetch_layer(sd, job.up, C, (size_t)C * 3, LU);
        if (rc != FASTSVC_OK) return rc;
        const int nchA = c.nch32, nchB = (C + 31) / 32;
        PackedConv v = c;
        v.nch32 = nchA + 2 * nchB;
        v.cin = v.nch32 * 32;                               // virtual channel axis: [first | residual | up] units
        auto poly_tap = [](const HostLayer& L, size_t base, int slot) -> float {    // W0 | W0+W1+W2 | W2 (MODE_POLY)
            if (slot == 1) return (float)((double)L.w[base] + (double)L.w[base + 1] + (double)L.w[base + 2]);
            return L.w[base + slot];
        };
        pack_hx(v, c.hxc_off, 3, [&](int co, int ci, int slot) -> float {
            if (ci < nchA * 32) return ci < cin ? LF.w[((size_t)co * cin + ci) * 3 + slot] : 0.f;
            const int cj = ci - nchA * 32;
            const bool up = cj >= nchB * 32;
            const int ck = up ? cj - nchB * 32 : cj;
            if (ck >= C) return 0.f;
            return poly_tap(up ? LU : LR, ((size_t)co * C + ck) * 3, slot);
        }, c.hxc_inv_off, 3, [nchA, nchB](int ci, int) { return ci < nchA * 32 ? 0 : (ci - nchA * 32 < nchB * 32 ? 1 : 2); });
        float* cst = blob + c.hxc_inv_off + 3 * (size_t)c.ngroups * 16 * c.MW;
        float l1 = 0.f, bmax = 0.f;
        for (int co = 0; co < C; ++co) {
            double sum = 0.0;
            for (size_t i = 0; i < (size_t)cin * 3; ++i) sum += std::fabs((double)LF.w[(size_t)co * cin * 3 + i]);
            l1 = std::max(l1, (float)sum);
            bmax = std::max(bmax, std::fabs(LF.b[co]));
        }
        cst[0] = l1; cst[1] = bmax; cst[2] = 0.f; cst[3] = 0.f;
        return FASTSVC_OK;
    });
    task_names.insert(task_names.end(), plan->cond_bound_jobs.size(), "cond_bounds");
    for (int k_ : plan->cond_bound_jobs) tasks.emplace_back([&, k = k_]() -> int {
        // |c1| <= A1 x, |c2| <= M2 |c1| + |b2|, |h| <= M3 |c2| + |b3| + |r|, |u| <= M4 |h| + |b4| (LeakyReLU only shrinks),
        // M[co][ci] = sum over taps of |w|: carried as per-channel (alpha, beta) pairs, bound = alpha * amax(x) + beta.
        // The max over channels of these is what the launch scales a tile by: orders of magnitude tighter than chaining
        // (largest row sum) x (largest input) layer by layer when weight_g differs across channels by decades.
        const DownStage& d = plan->down[k];
        const int C = d.C, Cin = d.Cin;
        for (int sgn = 0; sgn < 2; ++sgn) {
            const std::string pre = std::string("downsampling_") + (sgn ? "sine." : "lft.") + std::to_string(k);
            const std::string fpre = std::string("film_") + (sgn ? "sine." : "lft.") + std::to_string(k);
            HostLayer L1, LR, L2, L3, L4;
            int rc = fetch_layer(sd, pre + ".downsample_block.2", C, (size_t)Cin * 3, L1); if (rc != FASTSVC_OK) return rc;
            rc = fetch_layer(sd, pre + ".residual_block.0", C, (size_t)Cin, LR); if (rc != FASTSVC_OK) return rc;
            rc = fetch_layer(sd, pre + ".downsample_block.4", C, (size_t)C * 3, L2); if (rc != FASTSVC_OK) return rc;
            rc = fetch_layer(sd, pre + ".downsample_block.6", C, (size_t)C * 3, L3); if (rc != FASTSVC_OK) return rc;
            rc = fetch_layer(sd, fpre + ".conv", C, (size_t)C * 3, L4); if (rc != FASTSVC_OK) return rc;
            std::vector<double> a(C), b(C), a2(C), b2(C);
            float* out = blob + d.cbnd_off[sgn];
            auto emit = [&](int t) { for (int c = 0; c < C; ++c) { out[(t * 2 + 0) * C + c] = (float)(a[c] * 1.0000005); out[(t * 2 + 1) * C + c] = (float)(b[c] * 1.0000005); } };
            for (int c = 0; c < C; ++c) {
                double sum = 0.0;
                for (size_t i = 0; i < (size_t)Cin * 3; ++i) sum += std::fabs((double)L1.w[c * Cin * 3 + i]);
                a[c] = sum; b[c] = std::fabs((double)L1.b[c]);
            }
            emit(0);
            auto through = [&](const HostLayer& L) {
                for (int co = 0; co < C; ++co) {
                    double sa = 0.0, sb = 0.0;
                    for (int ci = 0; ci < C; ++ci) {
                        const float* w = &L.w[((size_t)co * C + ci) * 3];
                        const double m = std::fabs((double)w[0]) + std::fabs((double)w[1]) + std::fabs((double)w[2]);
                        sa += m * a[ci]; sb += m * b[ci];
                    }
                    a2[co] = sa; b2[co] = sb + std::fabs((double)L.b[co]);
                }
                a.swap(a2); b.swap(b2);
            };
            through(L2); emit(1);
            through(L3);
            for (int c = 0; c < C; ++c) {                     // + the 1x1 residual conv of the stage's input
                double sum = 0.0;
                for (int i = 0; i < Cin; ++i) sum += std::fabs((double)LR.w[c * Cin + i]);
                a[c] += sum; b[c] += std::fabs((double)LR.b[c]);
            }
            emit(2);
            through(L4); emit(3);
        }
        return FASTSVC_OK;
    });
    task_names.insert(task_names.end(), plan->raw_jobs.size(), "raw");
    for (const RawParam* r_ : plan->raw_jobs) tasks.emplace_back([&, r = r_]() -> int {
        HostLayer L;
        const int cout = (int)r->b_floats;
        const int rc = fetch_layer(sd, r->layer, cout, r->w_floats / cout, L);
        if (rc != FASTSVC_OK) return rc;
        std::memcpy(blob + r->w_off, L.w.data(), r->w_floats * sizeof(float));
        std::memcpy(blob + r->b_off, L.b.data(), r->b_floats * sizeof(float));
        {
            float l1 = 0.f, bmax = 0.f;
            const size_t per = r->w_floats / cout;
            for (int co = 0; co < cout; ++co) {
                double sum = 0.0;
                for (size_t i = 0; i < per; ++i) sum += std::fabs((double)L.w[co * per + i]);
                l1 = std::max(l1, (float)sum);
                bmax = std::max(bmax, std::fabs(L.b[co]));
            }
            blob[r->bnd_off] = l1; blob[r->bnd_off + 1] = bmax;
        }
        return FASTSVC_OK;
    });
    static const int env_threads = std::getenv("FASTSVC_PACK_THREADS") ? std::atoi(std::getenv("FASTSVC_PACK_THREADS")) : 0;
    unsigned nthreads = env_threads > 0 ? (unsigned)env_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = (unsigned)std::min<size_t>(nthreads, tasks.size());
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{FASTSVC_OK};
    std::mutex err_mu;
    std::string err_msg;
    static const bool pack_debug = std::getenv("FASTSVC_PACK_DEBUG") != nullptr;
    // Longest job first: the jobs differ 20:1 (the 2C -> 2C FiLM heads and block-diagonal chains of the C = 192 stage
    // against a 1 x 1 conv), the queue is dynamic, and a long job that starts last IS the makespan.  The order comes
    // from the durations the previous pack of this plan measured (a training step packs after every optimizer
    // update); the first pack runs in declaration order.
    std::vector<double> task_ms(tasks.size(), 0.0);
    std::vector<size_t> order(tasks.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    {
        std::lock_guard<std::mutex> lock(plan->tune_mu);
        if (plan->pack_cost.size() == tasks.size())
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return plan->pack_cost[a] > plan->pack_cost[b]; });
    }
    auto worker = [&]() {
        for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
            const size_t i = order[k];
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = tasks[i]();
            task_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rc != FASTSVC_OK) {
                std::lock_guard<std::mutex> lock(err_mu);
                if (first_rc.load() == FASTSVC_OK) { first_rc = rc; err_msg = g_err; }     // g_err: this thread's message
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    {
        std::lock_guard<std::mutex> lock(plan->tune_mu);
        plan->pack_cost = task_ms;
    }
    if (pack_debug) {
        double sum = 0.0, mx = 0.0;
        for (double v : task_ms) { sum += v; mx = std::max(mx, v); }
        std::vector<size_t> top(task_ms.size());
        for (size_t i = 0; i < top.size(); ++i) top[i] = i;
        std::sort(top.begin(), top.end(), [&](size_t a, size_t b) { return task_ms[a] > task_ms[b]; });
        std::fprintf(stderr, "[fastsvc] pack: %zu tasks on %u threads, sum %.1f ms, longest %.1f ms, top:", tasks.size(), nthreads, sum, mx);
        for (size_t i = 0; i < std::min<size_t>(8, top.size()); ++i)
            std::fprintf(stderr, " %s %.1f", top[i] < task_names.size() ? task_names[top[i]].c_str() : "?", task_ms[top[i]]);
        std::fprintf(stderr, "\n");
    }
    if (first_rc.load() != FASTSVC_OK) return fail(first_rc.load(), err_msg);
    return FASTSVC_OK;
}

}  // extern "C"

// ==========================================================================================
// workspace layout
// ==========================================================================================
namespace {

struct Workspace {
    std::vector<BufferSpec> bufs;
    size_t bytes = 0;
    std::map<std::string, int> index;

    size_t add(const std::string& name, int64_t d0, int64_t d1, int64_t d2, size_t elem = sizeof(float)) {
        BufferSpec b;
        b.name = name;
        b.off_bytes = bytes;
        b.numel = d0 * d1 * d2;
        b.shape[0] = d0; b.shape[1] = d1; b.shape[2] = d2;
        bytes += align_up((size_t)b.numel * elem, 256);
        index[name] = (int)bufs.size();
        bufs.push_back(b);
        return b.off_bytes;
    }
    const BufferSpec* find(const std::string& name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : &bufs[it->second];
    }
};

// Which conditioning stages run as ONE launch each (run_cond_stage0 / run_cond_stage1) - a function of the plan and
// the frame count alone, so that the compact layout can leave out what those launches never write (h_k, r_k, c1 / c2,
// film_u of the stage) and the forward can insist on the launch instead of falling back into missing buffers.
int cond_env_mode() {
    static const int v = std::getenv("FASTSVC_COND") ? std::atoi(std::getenv("FASTSVC_COND")) : 1;     // 0 none, 1 all, 4 stage 0 only
    return v;
}
bool amax_scan_disabled() {
    static const bool v = std::getenv("FASTSVC_NO_AMAX_SCAN") != nullptr;    // A/B timing only: inputs then count as unit-scale
    return v;
}
bool cond_stage_whole(const fastsvc_plan& P, int k, int F) {
    const int mode = cond_env_mode();
    if (!P.compact || !mode || P.n < k + 2) return false;
    if (P.storage == 0 && (F <= 4 || amax_scan_disabled())) return false;    // exact-f32 mode / no measured maxima
    int64_t hop = 1;
    for (int i = 0; i < P.n; ++i) hop *= P.cfg.upsampling_scales[i];
    const int64_t T = hop * F;
    const DownStage& d0 = P.down[0];
    const DownStage& d1 = P.down[1];
    if (d0.scale != 1 || d0.C != 24 || !d0.c2[0].hx || !d0.c3[0].hx || !d0.film[0].hx || !d0.heads.hx || d0.c2[0].MW != 2 ||
        d0.heads.MW != 3 || d0.heads.nch32 != 2 || !d1.rc1[0].dec2 || (T % 8) != 0 || T % d1.scale != 0 || (T / d1.scale) % 4 != 0)
        return false;
    if (k == 0) return true;
    if (k != 1 || (!(mode & 2) && mode != 1)) return false;
    const DownStage& d2 = P.down[2];
    const int64_t T1 = T / d1.scale;
    return d1.C == 48 && d1.Cin == 24 && d1.rc1[0].hx && d1.rc1[0].MW == 3 && d1.c2[0].hx && d1.c3[0].hx && d1.film[0].hx &&
           d1.c2[0].MW == 3 && d1.c2[0].nch32 == 2 && d1.heads.hx && d1.heads.MW == 3 && d1.heads.nch32 == 3 &&
           d1.heads.ngroups == 2 && d2.rc1[0].dec2 && (T1 % 8) == 0 && T1 % d2.scale == 0 && (T1 / d2.scale) % 4 == 0;
}

// Default layout: every intermediate has its own buffer, all taps stay readable after a forward.
// Compact layout (fastsvc_plan_set_workspace_mode): buffers whose lifetimes cannot overlap share one slot sized for
// the largest user -
//   down_c1.k / down_c2.k of all stages (written and read on the caller's stream inside stage k only);
//   a / xr / u1 / xmid / u2 / u3 of all up blocks (dead when the block's last conv has run; the helper-stream
//   producer of xr.i+1 is ordered behind conv_first.i+1, i.e. behind every reader of block i's buffers) - in the
//   SAME bytes as c1 / c2, which are dead when the first up block starts;
//   film_u.k of the stages whose FiLM net runs on the helper stream (written and read there, stage after stage).
// Tensors that a helper stream still reads while the caller's stream moves on (down_h, ss) and the block outputs
// keep their own buffers.  Taps of shared slots hold the LAST user's tensor.
Workspace layout_workspace(const fastsvc_plan& P, int B, int F) {
    Workspace ws;
    std::map<std::string, size_t> slot_off;           // compact layout: slot name -> offset
    auto add_shared = [&](const std::string& slot, const std::string& name, int64_t d0, int64_t d1, int64_t d2, size_t elem) {
        if (!P.compact) { ws.add(name, d0, d1, d2, elem); return; }
        BufferSpec b;
        b.name = name;
        b.off_bytes = slot_off.at(slot);
        b.numel = d0 * d1 * d2;
        b.shape[0] = d0; b.shape[1] = d1; b.shape[2] = d2;
        ws.index[name] = (int)ws.bufs.size();
        ws.bufs.push_back(b);
    };
    const int n = P.n;
    int64_t hop = 1;
    for (int i = 0; i < n; ++i) hop *= P.cfg.upsampling_scales[i];
    const int64_t T = hop * F;
    const size_t ae = P.storage == 1 ? 2 : sizeof(float);      // activation element size
    if (P.storage == 1) ws.add("ppg_act", B, P.cfg.in_channels, F, ae);
    // stages that run as whole-stage launches keep their intermediates in LDS: zero-sized placeholders (the amax rows
    // are indexed by buffer, taps of those names are empty)
    const bool whole[2] = {cond_stage_whole(P, 0, F), cond_stage_whole(P, 0, F) && cond_stage_whole(P, 1, F)};
    if (P.compact) {
        int64_t down_max = 0, up_max = 0, a_max = 0, u_max = 0, Td = T, Ti = F;
        for (int k = 0; k < n; ++k) {
            Td /= P.down[k].scale;
            if (k < 2 && whole[k]) continue;
            down_max = std::max<int64_t>(down_max, 2LL * B * P.down[k].C * Td);
            if (k + 1 < n) u_max = std::max<int64_t>(u_max, 2LL * B * P.down[k].C * Td);
        }
        for (int i = 0; i < n; ++i) {
            a_max = std::max<int64_t>(a_max, (int64_t)B * P.up[i].C * Ti);
            Ti *= P.up[i].scale;
            up_max = std::max<int64_t>(up_max, (int64_t)B * P.up[i].C * Ti);
        }
        // one arena for both phases: the conditioning chains' c1 / c2 are dead before the first up block starts
        // (same stream), whose a / xr / u1 / xmid / u2 / u3 then take the same bytes
        const size_t dn = align_up((size_t)down_max * ae, 256), up = align_up((size_t)up_max * ae, 256), aa = align_up((size_t)a_max * ae, 256);
        const size_t arena = std::max(2 * dn, aa + 5 * up);
        const size_t base = ws.add("shared.arena", 1, 1, (int64_t)(arena / ae), ae);
        slot_off["c1"] = base; slot_off["c2"] = base + dn;
        slot_off["a"] = base;
        const char* ups[5] = {"xr", "u1", "xmid", "u2", "u3"};
        for (int j = 0; j < 5; ++j) slot_off[ups[j]] = base + aa + j * up;
        // film_u of the stages whose FiLM net runs on the helper stream: written and read there, one after the other
        slot_off["film_u"] = ws.add("shared.film_u", 1, 1, u_max, ae);
    }
    int64_t Tk = T;
    for (int k = 0; k < n; ++k) {
        const DownStage& d = P.down[k];
        Tk = Tk / d.scale;
        const std::string s = std::to_string(k);
        const int64_t Bk = (k < 2 && whole[k]) ? 0 : B;
        if (k > 0) ws.add("down_r." + s, 2 * Bk, d.C, Tk, ae);
        add_shared("c1", "down_c1." + s, 2 * Bk, d.C, Tk, ae);
        add_shared("c2", "down_c2." + s, 2 * Bk, d.C, Tk, ae);
        ws.add("down_h." + s, 2 * Bk, d.C, Tk, ae);              // [lft batch ; sine batch]
        // h_0[..., ::s_1] compact: what the whole-stage launch of stage 0 hands to stage 1 instead of h_0 (run_cond_stage0)
        if (k == 0 && n > 1) ws.add("down_hd.1", 2 * B, d.C, Tk / P.down[1].scale, ae);
        if (k == 1 && n > 2) ws.add("down_hd.2", 2 * B, d.C, Tk / P.down[2].scale, ae);     // ... and of stage 1 (run_cond_stage1)
        if (k + 1 < n) add_shared("film_u", "film_u." + s, Bk, 2 * d.C, Tk, ae);    // channels [lft ; sine]
        else ws.add("film_u." + s, B, 2 * d.C, Tk, ae);          // (last stage: on the caller's stream)
        ws.add("ss." + s, B, 2 * d.C, Tk, ae);                   // channels [scale ; shift]
    }
    int64_t Tin = F;
    for (int i = 0; i < n; ++i) {
        const UpStage& u = P.up[i];
        const int64_t Tout = Tin * u.scale;
        const std::string s = std::to_string(i);
        add_shared("a", "up." + s + ".a", B, u.C, Tin, ae);
        add_shared("xr", "up." + s + ".xr", B, u.C, Tout, ae);
        add_shared("u1", "up." + s + ".u1", B, u.C, Tout, ae);      // scale * t0 + shift      (fastsvc.py:131-132)
        add_shared("xmid", "up." + s + ".xmid", B, u.C, Tout, ae);
        add_shared("u2", "up." + s + ".u2", B, u.C, Tout, ae);      // scale * xmid + shift
        add_shared("u3", "up." + s + ".u3", B, u.C, Tout, ae);      // scale * t2 + shift
        ws.add("up." + s + ".out", B, u.C, Tout, ae);
        ws.add("up." + s + ".spk", B, u.C, 1);
        Tin = Tout;
    }
    // InstanceNorm accumulators of all blocks side by side: ONE memset zeroes them
    for (int i = 0; i < n; ++i)
        ws.add("up." + std::to_string(i) + ".stats", 3 * B, P.up[i].C, 2, sizeof(double));
    // float32 storage: largest magnitude per (tensor, utterance) - the scale of the split-binary16 staging
    // (ConvParams::amax_in / amax_out); an entry is 8 slots in 8 different 128-byte lines = 256 floats (writers spread
    // over the slots, readers take the max).
    // amax_in: the caller's inputs [lft B | sine B | ppg B]; amax: one row of 2B entries per workspace tensor,
    // indexed like `bufs` (zeroed at the start of every forward).
    ws.add("amax_in", 3, B, 256);
    const int64_t nrows = (int64_t)ws.bufs.size() + 1;
    ws.add("amax", nrows, 2 * B, 256);
    return ws;
}

// Optional per-launch instrumentation (fastsvc_forward_profile): hipEvents on the launch stream.
struct Profiler {
    hipStream_t stream = nullptr;          // the caller's stream (synchronised in finish)
    hipStream_t cur = nullptr;             // stream of the launch being bracketed
    std::vector<fastsvc_launch_record> recs;
    std::vector<hipEvent_t> ev;
    hipError_t begin(hipStream_t on, const std::string& layer, const std::string& kernel, double flops, double bytes) {
        cur = on;
        fastsvc_launch_record r;
        std::memset(&r, 0, sizeof(r));
        std::snprintf(r.layer, sizeof(r.layer), "%s", layer.c_str());
        std::snprintf(r.kernel, sizeof(r.kernel), "%s", kernel.c_str());
        r.flops = flops; r.bytes = bytes;
        recs.push_back(r);
        hipEvent_t e0, e1;
        hipError_t e = hipEventCreate(&e0); if (e != hipSuccess) return e;
        e = hipEventCreate(&e1); if (e != hipSuccess) return e;
        ev.push_back(e0); ev.push_back(e1);
        return hipEventRecord(e0, cur);
    }
    hipError_t end() { return hipEventRecord(ev.back(), cur); }
    hipError_t finish() {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        for (size_t i = 0; i < recs.size(); ++i) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            if (e != hipSuccess) return e;
            recs[i].ms = ms;
        }
        for (hipEvent_t x : ev) hipEventDestroy(x);
        ev.clear();
        return hipSuccess;
    }
};

// Helper streams for the parts of the forward that are off the critical path (FiLM nets of the
// stages whose scale/shift are only needed by later up blocks, the 1x1 / stretch residual convs).
// They are forked from and joined back into the caller's stream with events, so the call stays
// asynchronous with respect to the host and ordered with respect to the caller's stream.
struct ExecCtx {
    hipStream_t aux[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;
    std::mutex busy;                   // one forward at a time enqueues through a context
    bool pays = false;                 // a fork + join through these streams is cheap enough to use them (measured once)
    bool measured = false;             // measure_fork_join has run (not under stream capture: tried again on the next forward)
    float fork_join_us = 0.f;
};

// What a fork + join between the caller's stream and a helper stream costs HERE, measured once per context with
// empty kernels: R rounds of [record, wait, launch on the helper, record, wait, launch] against 2R launches on the
// caller's stream alone.  Normally a few microseconds.  Helper streams that come to life after another library's
// streams (an RCCL communicator initialised between the process's first allocation and its first forward) take far
// longer per dependency - the two-stream schedule of cfg2 then ran 2.54 ms instead of 1.43 (tools/stream_order_check.py)
// - and the forward then stays on one stream.  Synchronises `stream` (first forward on a stream, or
// fastsvc_stream_prepare); skipped, and the helper streams left unused, while the stream is being captured.
bool measure_fork_join(hipStream_t stream, ExecCtx* c) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    hipEvent_t t[3] = {nullptr, nullptr, nullptr};
    for (hipEvent_t& e : t)
        if (hipEventCreate(&e) != hipSuccess) return false;
    const int R = 12;
    bool ok = true;
    auto alone = [&]() { for (int r = 0; r < 2 * R && ok; ++r) ok = launch_noop(stream) == hipSuccess; };
    auto forked = [&]() {
        for (int r = 0; r < R && ok; ++r) {
            hipEvent_t a = c->ev[(2 * r) % c->ev.size()], b = c->ev[(2 * r + 1) % c->ev.size()];
            ok = hipEventRecord(a, stream) == hipSuccess && hipStreamWaitEvent(c->aux[0], a, 0) == hipSuccess &&
                 launch_noop(c->aux[0]) == hipSuccess && hipEventRecord(b, c->aux[0]) == hipSuccess &&
                 hipStreamWaitEvent(stream, b, 0) == hipSuccess && launch_noop(stream) == hipSuccess;
        }
    };
    alone(); forked();                                    // (code objects loaded, queues awake)
    float best_alone = 1e30f, best_forked = 1e30f;
    for (int rep = 0; rep < 3 && ok; ++rep) {
        ok = ok && hipEventRecord(t[0], stream) == hipSuccess;
        alone();
        ok = ok && hipEventRecord(t[1], stream) == hipSuccess;
        forked();
        ok = ok && hipEventRecord(t[2], stream) == hipSuccess && hipEventSynchronize(t[2]) == hipSuccess;
        float ms_a = 0.f, ms_f = 0.f;
        ok = ok && hipEventElapsedTime(&ms_a, t[0], t[1]) == hipSuccess && hipEventElapsedTime(&ms_f, t[1], t[2]) == hipSuccess;
        if (ms_a < best_alone) best_alone = ms_a;
        if (ms_f < best_forked) best_forked = ms_f;
    }
    for (hipEvent_t e : t) (void)hipEventDestroy(e);
    if (!ok) return false;
    c->fork_join_us = (best_forked - best_alone) * 1e3f / R;
    static const double limit_us = std::getenv("FASTSVC_FORK_JOIN_LIMIT_US") ? std::atof(std::getenv("FASTSVC_FORK_JOIN_LIMIT_US")) : 40.0;   // measured: 24-27 us where the streams are independent, 52 us where they are not
    c->pays = c->fork_join_us < limit_us;
    if (std::getenv("FASTSVC_STREAM_DEBUG"))
        std::fprintf(stderr, "[fastsvc] fork + join through a helper stream: %.1f us (limit %.0f): helper streams %s\n",
                     c->fork_join_us, limit_us, c->pays ? "on" : "off");
    return true;
}

// One context per (device, caller stream): two host threads driving different streams of one device
// get their own helper streams and event rings, so a wait can never bind to the other thread's
// record; two threads sharing ONE stream serialise on `busy` for the duration of the enqueue.
// `nev` events cover every fork/join of a forward with FASTSVC_MAX_STAGES stages and all helper
// streams enabled (2 per down stage + 2 per up block + 1 ss_ready per stage, with slack), so the
// ring never wraps inside one call.  Created on first use: the call that creates it allocates
// streams/events and is therefore not graph-capturable - fastsvc_forward_prepare() does it ahead.
std::mutex g_ctx_mu;
std::map<std::pair<int, hipStream_t>, ExecCtx*> g_ctxs;

ExecCtx* exec_ctx_for(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    const auto key = std::make_pair(dev, stream);
    auto it = g_ctxs.find(key);
    if (it != g_ctxs.end()) return it->second;
    ExecCtx* c = new ExecCtx();
    // the FiLM helper stream runs at the LOWEST priority: its kernels only fill the CUs the
    // critical-path kernels leave idle; the residual-conv stream keeps the default priority
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);     // lo = numerically largest
    if (hipStreamCreateWithPriority(&c->aux[0], hipStreamNonBlocking, prio_lo) != hipSuccess) { delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->aux[1], hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    const int nev = 8 * FASTSVC_MAX_STAGES + 16;
    c->ev.resize(nev);
    for (int i = 0; i < nev; ++i)
        if (hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
    // (the fork + join calibration synchronises the stream: it runs OUTSIDE this global lock, under the context's own
    // `busy` lock - ensure_measured - so that first forwards on other streams / devices are not held up behind it)
    g_ctxs[key] = c;
    return c;
}

// under c->busy: calibrate once; while the stream is being captured the context stays unmeasured (helper streams
// unused) and the next uncaptured forward measures it
void ensure_measured(hipStream_t stream, ExecCtx* c) {
    if (c->measured) return;
    if (measure_fork_join(stream, c)) c->measured = true;
}

// fastsvc_stream_release: the context of (current device, stream) is drained and destroyed.  The helper
// streams are synchronised first - they may still hold work forked from a forward on `stream` - so a later
// stream that happens to get the same handle value starts from a fresh context with nothing pending.
// Returns 1 when a context existed, 0 when there was none.
int exec_ctx_release(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    ExecCtx* c = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        auto it = g_ctxs.find(std::make_pair(dev, stream));
        if (it == g_ctxs.end()) return 0;
        c = it->second;
        g_ctxs.erase(it);
    }
    {
        std::lock_guard<std::mutex> busy(c->busy);          // a forward still enqueueing through it finishes first
        for (hipStream_t a : c->aux)
            if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
        for (hipEvent_t e : c->ev)
            if (e) (void)hipEventDestroy(e);
    }
    delete c;
    return 1;
}

// Autotuning state of one forward call (fastsvc_autotune): when `tuning` every pipelined conv
// times all its candidate launch shapes on the device before the real launch.
struct TuneCtx {
    const fastsvc_plan* plan = nullptr;
    bool tuning = false;
    int trials = 0;
};
thread_local TuneCtx g_tune;
// set by run_conv when the launch it made also computed conv_last (ConvParams::last_w accepted)
thread_local bool g_last_fused = false;
// Batches of at most 4 frames (40 ms of audio) run on the f32-input-MFMA kernels throughout: an InstanceNorm over the
// 2 - 8 samples such an utterance has at the first blocks is ill-conditioned (1 / sqrt(var + 1e-5) up to 316), and the
// split-binary16 products' error - 2^-22 of the TENSOR's maximum rather than of each value - then shows: a 1-frame
// utterance sat at 3.4e-3 of the output's range where the float32 reference is 4e-5 from float64
// (tools/stress_parity.py, seed 75).  Set by forward_impl for the duration of the call; float32 storage only.
thread_local bool g_exact_f32 = false;


#ifdef FASTSVC_TIMELINE
// diagnostic build (python -m svcc23_fastsvc_amd.build --timeline; tools/timeline.py): per-wave cycle stamps
// of the launch named by FASTSVC_TIMELINE_LAYER, dumped to FASTSVC_TIMELINE_OUT.  True when this launch was it.
template <class LaunchFn>
bool timeline_launch(const char* layer, ConvParams p, long wgs, int nchunks, const ConvLaunch& L, hipStream_t stream,
                     LaunchFn&& fn, hipError_t& err) {
    const char* want = std::getenv("FASTSVC_TIMELINE_LAYER");
    const char* out = std::getenv("FASTSVC_TIMELINE_OUT");
    if (!want || !out || std::strcmp(want, layer) != 0) return false;
    const long cap = wgs < 8192 ? wgs : 8192;
    unsigned long long* dbuf = nullptr;
    const size_t bytes = (size_t)cap * 8 * 64 * sizeof(unsigned long long);
    if (hipMalloc(&dbuf, bytes) != hipSuccess) { err = hipErrorOutOfMemory; return true; }
    hipMemsetAsync(dbuf, 0, bytes, stream);
    p.tl = dbuf; p.tl_wgs = (int)cap;
    err = fn(p);
    if (err != hipSuccess) return true;
    hipStreamSynchronize(stream);
    std::vector<unsigned long long> host((size_t)cap * 8 * 64);
    hipMemcpy(host.data(), dbuf, bytes, hipMemcpyDeviceToHost);
    hipFree(dbuf);
    if (FILE* f = std::fopen(out, "wb")) {
        const long hdr[8] = {cap, wgs, p.tpw, nchunks, L.MW, L.NW, L.WM, L.WN};
        std::fwrite(hdr, sizeof(long), 8, f);
        std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
        std::fclose(f);
    }
    return true;
}
#endif

// The c2 -> c3 pair of a down stage as ONE launch (fastsvc_hx.hip, MODE_CHAIN): p describes the SECOND conv's
// launch (x = the first conv's input, y / res / r1x as for c3).  `done` = false when the pair has no fused
// variant for this call (the caller then runs the two launches); with a tuning pass the fused launch is
// also timed against the two separate ones and the table remembers which won (algo 3 fused, 0 separate).
hipError_t run_conv(const PackedConv& c, const float* blob, ConvParams p, int nsig, long pair_w_stride,
                    long pair_b_stride, hipStream_t stream, Profiler* prof, const char* layer);

hipError_t run_chain(const PackedConv& a, const PackedConv& c, const float* blob, ConvParams p, int nsig,
                     long pair_b_stride, hipStream_t stream, Profiler* prof, const char* layer, double sep_ms, bool& done,
                     const RawParam* in1 = nullptr) {
    // in1: the stage's 1 -> C first conv (lft / sine twins) computed by the launch itself from the raw signal p.x
    // (MODE_CHAIN1)
    done = false;
    static const int hx_env = std::getenv("FASTSVC_HX") ? std::atoi(std::getenv("FASTSVC_HX")) : 1;
    static const int chain_env = std::getenv("FASTSVC_CHAIN") ? std::atoi(std::getenv("FASTSVC_CHAIN")) : 1;
    const bool act_bf16 = g_tune.plan && g_tune.plan->storage == 1;
    const int prec = act_bf16 ? 1 : 0;
    p.ldx = p.x_T; p.ldy = p.T;
    if (p.lens) { p.len_mul = p.T / p.frames_ld; p.xlen_mul = p.x_T / p.frames_ld; }
    if (!hx_env || !chain_env || g_exact_f32 || !c.hxc_off[prec] || (p.T & 3) || p.x_T != p.T ||
        (p.lens && ((p.len_mul & 3) || (p.xlen_mul & 3))))
        return hipSuccess;
    p.mode = in1 ? MODE_CHAIN1 : MODE_CHAIN;
    if (in1) {
        p.in1_w = blob + in1[0].w_off; p.in1_b = blob + in1[0].b_off;
        p.in1_w_sig = (long)(in1[1].w_off - in1[0].w_off); p.in1_b_sig = (long)(in1[1].b_off - in1[0].b_off);
    }
    p.CIN = a.cin; p.CMID = a.cout; p.COUT = c.cout;
    p.nch32 = a.nch32; p.nch32b = c.nch32; p.dil = a.dil; p.dil2 = c.dil; p.ntaps = 3; p.ngroups = c.ngroups;
    p.whx = blob + c.hxc_off[prec]; p.whx_sig = c.hxc_pair[prec];
    p.whx_inv = (prec == 0 && c.hxc_inv_off) ? blob + c.hxc_inv_off : nullptr; p.whx_inv_sig = c.hxc_inv_pair;
    p.bias = blob + c.b_off; p.bias_sig = pair_b_stride;
    p.bias_mid = blob + c.bmid_off; p.bias_mid_sig = c.bmid_pair;
    p.vec = 1; p.tpw = 1;
    {
        static const int dbg = std::getenv("FASTSVC_DBG") ? std::atoi(std::getenv("FASTSVC_DBG")) : 0;
        p.dbg = dbg;
    }
    auto launch = [&](const ConvParams& q, const ConvLaunch& Lq) {
        return act_bf16 ? bf16::launch_conv_hx(q, Lq, stream) : launch_conv_hx(q, Lq, stream);
    };
    struct Cand { int NW, WM, WN; };
    std::vector<Cand> cands;
    static const int shapes[][3] = {{6, 2, 2}, {4, 2, 2}, {3, 1, 4}, {2, 1, 4}};
    const int np = act_bf16 ? 1 : 2;
    for (const auto& sh : shapes) {
        if (!conv_hx_shape(p.mode, c.MW, sh[0], sh[1], sh[2]) || c.ngroups != sh[1]) continue;
        const int NT = 16 * sh[0] * sh[2];
        const size_t lds = (size_t)2 * np * (NT + 16 + 8 + 4) * 64 + (size_t)c.nch32 * np * (NT + 16) * 64 + 4096;
        if (lds <= 160 * 1024) cands.push_back(Cand{sh[0], sh[1], sh[2]});
    }
    if (cands.empty()) return hipSuccess;
    char key[96];
    std::snprintf(key, sizeof(key), act_bf16 ? "%s|%d|%d|b" : "%s|%d|%d", layer, p.B, p.T);
    bool have = false, fused = true;
    Cand best = cands[0];
    if (g_tune.plan) {
        std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
        auto it = g_tune.plan->tuned.find(key);
        if (it != g_tune.plan->tuned.end()) {
            if (it->second.algo != 3 && chain_env != 2) { have = true; fused = false; }      // FASTSVC_CHAIN=2: fused regardless (A/B)
            for (const Cand& cd : cands)
                if (it->second.algo == 3 && cd.NW == it->second.NW && cd.WM == it->second.WM && cd.WN == it->second.WN &&
                    it->second.tpw >= 1 && it->second.tpw <= 64) { best = cd; p.tpw = it->second.tpw; have = true; }
        }
    }
    if (!have && g_tune.tuning && g_tune.plan) {
        static const int tpws[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24};
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return hipErrorUnknown;
        float best_ms = 1e30f;
        ConvParams q = p;
        for (const Cand& cd : cands) {
            const long ntx = (p.T + 16 * cd.NW * cd.WN - 1) / (16 * cd.NW * cd.WN);
            for (int tpw : tpws) {
                q.tpw = tpw;
                ConvLaunch Lq{c.MW, cd.NW, cd.WM, cd.WN, nsig, 2};
                hipError_t e = launch(q, Lq);
                if (e != hipSuccess) return e;
                hipEventRecord(e0, stream);
                for (int r = 0; r < 3; ++r) { e = launch(q, Lq); if (e != hipSuccess) return e; }
                hipEventRecord(e1, stream);
                if (hipEventSynchronize(e1) != hipSuccess) return hipErrorUnknown;
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                ++g_tune.trials;
                if (ms < best_ms) { best_ms = ms; best = cd; p.tpw = tpw; }
                if (tpw >= ntx) break;
            }
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
        fused = sep_ms <= 0.0 || best_ms / 3.0 < sep_ms;
        std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
        g_tune.plan->tuned[key] = fastsvc_plan::Choice{best.NW, best.WM, best.WN, p.tpw, fused ? 3 : 0};
        g_tune.plan->priors.clear();
        have = true;
    }
    if (!have) {
        // no table entry: the widest tile that exists, workgroups sized so that the grid fills the CUs about evenly
        const int NT = 16 * best.NW * best.WN;
        const long ntx = (p.T + NT - 1) / NT;
        const long zb = (long)nsig * p.B;
        double best_t = 1e30;
        for (int tpw = 1; tpw <= 16; ++tpw) {
            const long wgs = ((ntx + tpw - 1) / tpw) * zb;
            const double t = (double)((wgs + 255) / 256) * (4.0 + tpw * c.nch32 * 2.0);
            if (t < best_t * 0.999) { best_t = t; p.tpw = tpw; }
        }
    }
    if (!fused) return hipSuccess;
    ConvLaunch L{c.MW, best.NW, best.WM, best.WN, nsig, 2};
    done = true;
    if (prof) {
        const double cols = (double)p.T * p.B * nsig;
        // (split input = the FiLM net: its first conv is block-diagonal over the two signals, half the dense count)
        const double flops = 2.0 * 3.0 * ((p.xsplit ? 0.5 : 1.0) * a.cin * a.cout + (double)c.cin * c.cout + (in1 ? (double)a.cin : 0.0)) * cols;
        double el = (in1 ? 1.0 : (double)a.cin) * p.T + (double)c.cout * p.T;
        if (p.res) el += (double)c.cout * p.T;
        if (p.r1x) el += (double)p.T;
        const double bytes = (act_bf16 ? 2.0 : 4.0) * el * p.B * nsig +
                             4.0 * (double)(a.w_floats + a.b_floats + c.w_floats + c.b_floats) * nsig;
        char kname[48];
        std::snprintf(kname, sizeof(kname), "conv_hx<%d,%d,%d,%d,%d,%d,1,%s>", L.MW, L.NW, L.WM, L.WN, p.mode,
                      p.r1x ? 3 : p.res ? 2 : 1, act_bf16 ? "x1" : "x3");
        hipError_t e = prof->begin(stream, layer, kname, flops, bytes);
        if (e != hipSuccess) return e;
        e = launch(p, L);
        if (e != hipSuccess) return e;
        return prof->end();
    }
#ifdef FASTSVC_TIMELINE
    {
        const int NT = 16 * L.NW * L.WN;
        const long ntx = (p.T + NT - 1) / NT;
        const long wgs = ((ntx + p.tpw - 1) / p.tpw) * (long)nsig * p.B;
        hipError_t e = hipSuccess;
        if (timeline_launch(layer, p, wgs, p.nch32, L, stream, [&](const ConvParams& q) { return launch(q, L); }, e)) return e;
    }
#endif
    return launch(p, L);
}

// The head of an up block as ONE launch (fastsvc_hx.hip, MODE_UPHEAD): p describes the input x (T = x_T = input
// columns), y = xr, y2 = u1 with ss_out / st_out; `done` = false when this call has no fused variant (float32 storage,
// rows a multiple of 4 long, table entry "<layer>|B|T" not 0).
hipError_t run_uphead(const UpStage& u, const float* blob, ConvParams p, hipStream_t stream, Profiler* prof, const char* layer,
                      bool& done) {
    done = false;
    static const int hx_env = std::getenv("FASTSVC_HX") ? std::atoi(std::getenv("FASTSVC_HX")) : 1;
    static const int head_env = std::getenv("FASTSVC_UPHEAD") ? std::atoi(std::getenv("FASTSVC_UPHEAD")) : 1;
    const PackedConv& c = u.head;
    const bool act_bf16 = g_tune.plan && g_tune.plan->storage == 1;
    if (!hx_env || !head_env || g_exact_f32 || act_bf16 || !c.hxc_off[0] || (p.x_T & 3) || g_tune.tuning) return hipSuccess;
    p.T = p.x_T;
    p.ldx = p.x_T; p.ldy = p.x_T * u.scale;
    if (p.lens) { p.len_mul = p.T / p.frames_ld; p.xlen_mul = p.x_T / p.frames_ld; if ((p.len_mul & 3) != 0) return hipSuccess; }
    p.mode = MODE_UPHEAD;
    p.CIN = u.Cin; p.CMID = u.C; p.COUT = u.C;
    p.nch32 = c.nch32; p.nch32b = (u.C + 31) / 32; p.dil = 1; p.dil2 = 1; p.ntaps = 3; p.ngroups = c.ngroups; p.s = u.scale;
    p.whx = blob + c.hxc_off[0]; p.whx_sig = 0;
    p.whx_inv = blob + c.hxc_inv_off; p.whx_inv_sig = 0;
    p.bias = blob + c.b_off; p.bias2 = blob + c.b2_off; p.bias_mid = blob + c.bmid_off;
    p.vec = 1; p.tpw = 1;
    // one shape per channel count: the workgroup holds every channel of its tile (twice)
    ConvLaunch L{c.MW, 2, 1, 4, 1, 2};
    if (c.MW == 3 && c.ngroups == 4) { L.NW = 4; L.WM = 4; L.WN = 1; }
    else if (c.MW == 3 && c.ngroups == 2) { L.NW = 2; L.WM = 2; L.WN = 2; }
    else if (c.ngroups != 1) return hipSuccess;
    if (!conv_hx_shape(MODE_UPHEAD, L.MW, L.NW, L.WM, L.WN)) return hipSuccess;
    const int NT = 16 * L.NW * L.WN;
    const size_t lds = (size_t)2 * 2 * (NT + 16 + 8 + 4) * 64 + (size_t)2 * p.nch32b * 2 * (NT + 16) * 64 + 8192;
    if (lds > 160 * 1024) return hipSuccess;
    char key[96];
    std::snprintf(key, sizeof(key), "%s|%d|%d", layer, p.B, p.T);
    // Measured (profiles/r3_layers_*): the fused head costs about what the three launches cost (its consumer waves
    // run conv_first, two polyphase convs and two epilogues back to back while the separate launches spread over more
    // workgroups), so it only runs where the launch table says it won for this (B, T): algorithm 3; FASTSVC_UPHEAD=2
    // forces it (A/B).
    bool fused = head_env == 2;
    if (g_tune.plan) {
        std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
        auto it = g_tune.plan->tuned.find(key);
        if (it != g_tune.plan->tuned.end()) {
            fused = it->second.algo == 3 || (head_env == 2 && it->second.algo != 0);
            if (it->second.algo == 0) fused = false;                     // table says: separate launches
            if (it->second.tpw >= 1 && it->second.tpw <= 64) p.tpw = it->second.tpw;
        }
    }
    if (!fused) return hipSuccess;
    done = true;
    if (prof) {
        const double cin = u.Cin, C = u.C, Ti = (double)p.x_T, To = Ti * u.scale;
        const double flops = (2.0 * 3.0 * cin * C * Ti + 2.0 * 2.0 * 3.0 * C * C * To) * p.B;
        const double bytes = 4.0 * (cin * Ti + 4.0 * C * To) * p.B +
                             4.0 * (double)(u.first.w_floats + u.res.w_floats + u.up.w_floats + 3 * u.res.b_floats);
        char kname[48];
        std::snprintf(kname, sizeof(kname), "conv_hx<%d,%d,%d,%d,8,4,%d,x3>", L.MW, L.NW, L.WM, L.WN, u.scale);
        hipError_t e = prof->begin(stream, layer, kname, flops, bytes);
        if (e != hipSuccess) return e;
        e = launch_conv_hx(p, L, stream);
        if (e != hipSuccess) return e;
        return prof->end();
    }
    return launch_conv_hx(p, L, stream);
}

// The middle conv of an up block (d = 3) with the block's stretched residual conv folded into its accumulator
// (fastsvc_hx.hip, ConvParams::x2; fastsvc.py:94-100: xmid = conv_d3(lrelu(norm(u1))) + conv_res(stretch(a))): the residual
// tensor xr - one launch, one write and one read of a (B, C, T) tensor per block - is never materialised.  p describes the
// d = 3 conv's launch WITHOUT its residual operand, plus x2 / x2_b / x2_T / s2 / amax_x2 / bnd_x2.
// what: 0 = only answer whether this call would run fused (`done`), 1 = launch, 2 = tuning pass: time the fused shapes
// against `sep_ms` (the two separate launches) and record the winner under "<layer>|B|T" (algorithm 3 fused, 0 separate).
hipError_t run_d3x(const UpStage& u, const float* blob, ConvParams p, hipStream_t stream, Profiler* prof, const char* layer,
                   int what, double sep_ms, bool& done) {
    done = false;
    static const int hx_env = std::getenv("FASTSVC_HX") ? std::atoi(std::getenv("FASTSVC_HX")) : 1;
    static const int x_env = std::getenv("FASTSVC_D3X") ? std::atoi(std::getenv("FASTSVC_D3X")) : 1;     // 0: never, 2: wherever it exists (A/B)
    const PackedConv& c = u.d3;
    const bool act_bf16 = g_tune.plan && g_tune.plan->storage == 1;
    const int prec = act_bf16 ? 1 : 0;
    p.ldx = p.x_T; p.ldy = p.T; p.ldx2 = p.x2_T;
    if (p.lens) { p.len_mul = p.T / p.frames_ld; p.xlen_mul = p.x_T / p.frames_ld; p.x2len_mul = p.x2_T / p.frames_ld; }
    if (!hx_env || !x_env || g_exact_f32 || p.no_hx || !c.hxc_off[prec] || (p.T & 3) || p.x_T != p.T || (long)p.x2_T * p.s2 != p.T ||
        (p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0)) || !conv_hx_x2_ok(c.MW, c.nch32, p.s2) || c.dil > 28)
        return hipSuccess;
    p.mode = MODE_DIRECT; p.s = 1;
    p.CIN = c.cin; p.KC = c.KC; p.nchunks = c.nchunks; p.w = blob + c.w_off; p.Q = c.Q;
    p.COUT = c.cout; p.ngroups = c.ngroups; p.ntaps = 3; p.dil = c.dil; p.nch32 = c.nch32;
    p.whx = blob + c.hxc_off[prec]; p.whx_sig = 0;
    p.whx_inv = prec == 0 ? blob + c.hxc_inv_off : nullptr; p.whx_inv_sig = 0;
    p.bias = blob + c.b_off; p.bias_sig = 0; p.bias2 = blob + c.b2_off; p.bias2_sig = 0;
    p.res = nullptr;
    p.vec = 1; p.tpw = 1; p.xs = 0; p.ps = 0;
    {
        static const int stagger = std::getenv("FASTSVC_STAGGER") ? std::atoi(std::getenv("FASTSVC_STAGGER")) : 2;
        p.stagger = stagger;
        static const int dbg = std::getenv("FASTSVC_DBG") ? std::atoi(std::getenv("FASTSVC_DBG")) : 0;
        p.dbg = dbg;
    }
    auto launch = [&](const ConvParams& q, const ConvLaunch& Lq) {
        return act_bf16 ? bf16::launch_conv_hx(q, Lq, stream) : launch_conv_hx(q, Lq, stream);
    };
    struct Cand { int NW, WM, WN; };
    std::vector<Cand> cands;
    static const int shapes[][3] = {{4, 2, 2}, {2, 1, 4}, {3, 1, 4}, {2, 2, 2}};      // (FiLM-affine epilogue: MW * NW <= 12)
    for (const auto& sh : shapes)
        if (conv_hx_shape(MODE_DIRECT, c.MW, sh[0], sh[1], sh[2]) && c.ngroups % sh[1] == 0 && sh[0] * c.MW <= 12)
            cands.push_back(Cand{sh[0], sh[1], sh[2]});
    if (cands.empty()) return hipSuccess;
    char key[96];
    std::snprintf(key, sizeof(key), act_bf16 ? "%s|%d|%d|b" : "%s|%d|%d", layer, p.B, p.T);
    bool have = false, fused = true;
    Cand best = cands[0];
    if (g_tune.plan) {
        std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
        auto it = g_tune.plan->tuned.find(key);
        if (it != g_tune.plan->tuned.end()) {
            if (it->second.algo != 3 && x_env != 2) { have = true; fused = false; }
            for (const Cand& cd : cands)
                if (cd.NW == it->second.NW && cd.WM == it->second.WM && cd.WN == it->second.WN &&
                    it->second.tpw >= 1 && it->second.tpw <= 64) { best = cd; p.tpw = it->second.tpw; have = true; }
        }
    }
    if (what == 2) {
        if (have || !g_tune.tuning || !g_tune.plan) return hipSuccess;
        static const int tpws[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24};
        struct EventPair {                                 // (destroyed on every way out of the timing loop: ADVICE r5)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            ~EventPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
        } ev;
        if (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess) return hipErrorUnknown;
        float best_ms = 1e30f;
        ConvParams q = p;
        q.flags &= ~F_STATS;                               // (the trial launches do not accumulate InstanceNorm sums)
        for (const Cand& cd : cands) {
            const long ntx = (p.T + 16 * cd.NW * cd.WN - 1) / (16 * cd.NW * cd.WN);
            for (int tpw : tpws) {
                q.tpw = tpw;
                ConvLaunch Lq{c.MW, cd.NW, cd.WM, cd.WN, 1, 2};
                hipError_t e = launch(q, Lq);
                if (e != hipSuccess) return e;
                hipEventRecord(ev.e0, stream);
                for (int r = 0; r < 3; ++r) { e = launch(q, Lq); if (e != hipSuccess) return e; }
                hipEventRecord(ev.e1, stream);
                if (hipEventSynchronize(ev.e1) != hipSuccess) return hipErrorUnknown;
                float ms = 0.f;
                hipEventElapsedTime(&ms, ev.e0, ev.e1);
                ++g_tune.trials;
                if (ms < best_ms) { best_ms = ms; best = cd; p.tpw = tpw; }
                if (tpw >= ntx) break;
            }
        }
        // (sep_ms times the two separate launches back to back on ONE stream; in production the residual conv runs on the side
        // stream under up_stretch, so this comparison favours the fused launch by at most the residual conv's own time - the
        // fused launch also saves that tensor's write and read, which the whole-forward timing of tools/tune_shapes.py sees)
        fused = sep_ms <= 0.0 || best_ms / 3.0 < sep_ms;
        std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
        g_tune.plan->tuned[key] = fastsvc_plan::Choice{best.NW, best.WM, best.WN, p.tpw, fused ? 3 : 0};
        g_tune.plan->priors.clear();
        return hipSuccess;
    }
    if (!have) {
        if (g_tune.tuning) return hipSuccess;              // (a tuning pass runs the separate launches, then times this one)
        // no table entry: the nearest entry of this layer, else the first shape; workgroups sized so that the grid
        // fills the CUs about evenly
        if (g_tune.plan) {
            fastsvc_plan::Choice pr{0, 0, 0, 0, -1};
            std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
            if (g_tune.plan->prior_for(layer, p.B, p.T, act_bf16, pr)) {
                if (pr.algo != 3 && x_env != 2) fused = false;
                for (const Cand& cd : cands) if (cd.NW == pr.NW && cd.WM == pr.WM && cd.WN == pr.WN) best = cd;
            }
        }
        const int NT = 16 * best.NW * best.WN;
        const long ntx = (p.T + NT - 1) / NT;
        const long gy = c.ngroups / best.WM;
        double best_t = 1e30;
        for (int tpw = 1; tpw <= 24; ++tpw) {
            const long wgs = ((ntx + tpw - 1) / tpw) * gy * p.B;
            const double t = (double)((wgs + 255) / 256) * (4.0 + tpw * c.nch32 * 2.0);
            if (t < best_t * 0.999) { best_t = t; p.tpw = tpw; }
        }
    }
    if (!fused) return hipSuccess;
    done = true;
    if (what == 0) return hipSuccess;
    ConvLaunch L{c.MW, best.NW, best.WM, best.WN, 1, 2};
    if (prof) {
        const double cols = (double)p.T * p.B;
        const double flops = 2.0 * 2.0 * 3.0 * c.cin * c.cout * cols;      // both convs at the OUTPUT rate, as the reference runs them
        const double el = (double)c.cin * p.T + (double)c.cin * p.x2_T + 4.0 * c.cout * p.T;   // u1, a | xmid, u2, scale, shift
        const double bytes = (act_bf16 ? 2.0 : 4.0) * el * p.B + 4.0 * 2.0 * (double)(c.w_floats + c.b_floats);
        char kname[48];
        std::snprintf(kname, sizeof(kname), "conv_hx<%d,%d,%d,%d,0,4,%d,%s>", L.MW, L.NW, L.WM, L.WN, p.s2, act_bf16 ? "x1" : "x3");
        hipError_t e = prof->begin(stream, layer, kname, flops, bytes);
        if (e != hipSuccess) return e;
        e = launch(p, L);
        if (e != hipSuccess) return e;
        return prof->end();
    }
#ifdef FASTSVC_TIMELINE
    {
        const int NT = 16 * L.NW * L.WN;
        const long ntx = (p.T + NT - 1) / NT;
        const long wgs = ((ntx + p.tpw - 1) / p.tpw) * (long)(c.ngroups / L.WM) * p.B;
        hipError_t e = hipSuccess;
        if (timeline_launch(layer, p, wgs, 2 * p.nch32, L, stream, [&](const ConvParams& q) { return launch(q, L); }, e)) return e;
    }
#endif
    return launch(p, L);
}

// Stage 0 of the conditioning nets as ONE launch (fastsvc_cond.hip): raw signals -> ss.0 and the compact h_0[..., ::s_1].
// `done` = false when this call has no such variant (the caller runs the separate launches).
hipError_t run_cond_stage0(const fastsvc_plan& P, const float* blob, const float* sig, long sig_stride, int B, long T, int F,
                           const int* lengths, float* ss, float* hd, const float* amax_in, float* amax_hd, hipStream_t stream,
                           Profiler* prof, bool& done) {
    done = false;
    static const int tpw_env = std::getenv("FASTSVC_COND_TPW") ? std::atoi(std::getenv("FASTSVC_COND_TPW")) : 0;
    // (float32 storage needs the measured maxima of the raw signals: not with FASTSVC_NO_AMAX_SCAN)
    if (!cond_stage_whole(P, 0, F) || g_exact_f32 || (P.storage == 0 && (!amax_in || !amax_hd))) return hipSuccess;
    const DownStage& d = P.down[0];
    const DownStage& d1 = P.down[1];
    const int prec = P.storage == 1 ? 1 : 0;
    CondStage0Params q;
    std::memset(&q, 0, sizeof(q));
    q.x = sig; q.x_sig = sig_stride; q.x_b = T;
    q.B = B; q.C = d.C; q.T = (int)T; q.ld = (int)T;
    q.lens = lengths; q.len_mul = (int)(T / F);
    const PackedConv* layers[3] = {d.c2, d.c3, d.film};
    for (int s = 0; s < 2; ++s) {
        q.in1_w[s] = blob + d.c1_raw[s].w_off; q.in1_b[s] = blob + d.c1_raw[s].b_off;
        q.r1w[s] = blob + d.r_raw[s].w_off; q.r1b[s] = blob + d.r_raw[s].b_off;
        for (int l = 0; l < 3; ++l) {
            q.w[l][s] = blob + layers[l][s].hx_off[prec];
            q.bias[l][s] = blob + layers[l][s].b_off;
            q.winv[l][s] = blob + layers[l][s].hx_inv_off;
            q.bnd[l + 1][s] = blob + layers[l][s].bnd_off;
        }
        q.bnd[0][s] = blob + d.c1_raw[s].bnd_off;
        q.bnd_r[s] = blob + d.r_raw[s].bnd_off;
        q.cbnd[s] = d.cbnd_off[s] ? blob + d.cbnd_off[s] : nullptr;
    }
    q.w5 = blob + d.heads.hx_off[prec]; q.b5 = blob + d.heads.b_off; q.winv5 = blob + d.heads.hx_inv_off;
    q.amax_in = P.storage == 0 ? amax_in : nullptr; q.amax_hd = P.storage == 0 ? amax_hd : nullptr;
    q.ss = ss; q.ss_b = 2L * d.C * T;
    q.hd = hd; q.hd_ld = (int)(T / d1.scale); q.hd_s = d1.scale;
    q.hd_b = (long)d.C * q.hd_ld; q.hd_sig = (long)B * q.hd_b;
    // tiles per workgroup: the grid runs in whole rounds of the 2 x 256 resident workgroups, at most ~32 tiles each
    // (a workgroup's start - 24 KB of weight fragments per wave, zeroed tiles - costs about two tiles)
    // short tiles where the long ones would leave CUs without a workgroup (a batch below one round of them)
    const long slots_all = P.storage == 1 ? 512 : 256;
    static const int small_env = std::getenv("FASTSVC_COND_SMALL") ? std::atoi(std::getenv("FASTSVC_COND_SMALL")) : -1;
    {
        const int NTb = cond_stage0_tile_columns(0);
        q.small = small_env >= 0 ? small_env : (((T + NTb - 1) / NTb) * B < slots_all ? 1 : 0);
    }
    // The layer pipeline (cond_stage0_pipe_kernel; q.small = 2, tpw = chunks per workgroup) where a workgroup gets a long
    // run of chunks - its fill costs 5 + 8 / N steps -: one or two rounds of the 256 resident workgroups (one per CU).
    // FASTSVC_COND_PIPE = 0: never, 2: always (tests, A/B); a launch-table entry "cond.0|B|T[|b]" with algorithm 4
    // (phase kernel) / 5 (pipeline) decides for one batch shape.
    static const int pipe_env = std::getenv("FASTSVC_COND_PIPE") ? std::atoi(std::getenv("FASTSVC_COND_PIPE")) : 1;
    int pipe_mode = pipe_env;
    {
        char key[96];
        std::snprintf(key, sizeof(key), P.storage == 1 ? "cond.0|%d|%ld|b" : "cond.0|%d|%ld", B, (long)T);
        std::lock_guard<std::mutex> lock(P.tune_mu);
        auto it = P.tuned.find(key);
        if (it != P.tuned.end() && it->second.algo == 4) pipe_mode = 0;
        if (it != P.tuned.end() && it->second.algo == 5) pipe_mode = 2;
    }
    if (pipe_mode && small_env < 0) {
        const int NTp = cond_stage0_tile_columns(P.storage == 1 ? 3 : 2);
        const long nchunks = (T + NTp - 1) / NTp;
        long kc = 0;
        {   // ONE round of the 256 resident workgroups (two rounds of half the run measured 1-3 % slower: tools/cond_pipe_tpw.sh)
            const long per_utt = std::max<long>(1, 256L / B);
            const long k = (nchunks + per_utt - 1) / per_utt;
            if (k >= 24 || pipe_mode == 2) kc = k;
        }
        // (the pipeline addresses both signals' hd rows of an utterance through ONE 32-bit-offset descriptor)
        const bool hd_fits = ((long)B * q.hd_b + (long)d.C * q.hd_ld) * (P.storage == 1 ? 2L : 4L) < (1L << 31);
        if (kc && hd_fits) { q.small = 2; q.tpw = (int)std::min<long>(tpw_env > 0 ? tpw_env : kc, 0xffff); }
    }
    const int NT = q.small == 2 ? 0 : cond_stage0_tile_columns(q.small);
    const long ntx = q.small == 2 ? 0 : (T + NT - 1) / NT;
    if (q.small == 2) {}
    else if (tpw_env > 0) q.tpw = tpw_env;
    else {
        const long slots = P.storage == 1 ? 512 : 256, total = ntx * B;     // workgroups resident at a time (LDS: two per CU in bfloat16 storage, one in float32)
        const long rounds = std::max<long>(1, (total + slots * 32 - 1) / (slots * 32));
        long tpw = std::max<long>(1, (total + slots * rounds - 1) / (slots * rounds));
        // (per-utterance rounding: ceil(ntx / tpw) * B workgroups must not spill into one more round)
        while (tpw < ntx && ((ntx + tpw - 1) / tpw) * B > slots * rounds) ++tpw;
        q.tpw = (int)std::min<long>(tpw, ntx);
    }
    static const int dbg_env = std::getenv("FASTSVC_COND_DBG") ? std::atoi(std::getenv("FASTSVC_COND_DBG")) : 0;
    q.tpw |= dbg_env << 16;
    static unsigned long long* trace_buf = nullptr;
#ifdef FASTSVC_COND_TRACE
    static const bool trace_env = std::getenv("FASTSVC_COND_TRACE") != nullptr;
#else
    // (only a library whose kernels were built with -DFASTSVC_COND_TRACE writes stamps through amax_hd: in the product build the
    // kernel's atomicMax into that 4 KB buffer would run out of bounds for B >= 3)
    static const bool trace_env = false;
#endif
    if (trace_env && prof) {
        if (!trace_buf) (void)hipMalloc(&trace_buf, 8 * 64 * 8);
        (void)hipMemsetAsync(trace_buf, 0, 8 * 64 * 8, stream);
        q.amax_hd = reinterpret_cast<float*>(trace_buf);
    }
    done = true;
    auto launch = [&]() { return P.storage == 1 ? bf16::launch_cond_stage0(q, stream) : launch_cond_stage0(q, stream); };
    if (prof) {
        const double C = d.C, cols = (double)T * B;
        const double flops = (2.0 * (2.0 * (3.0 * C + C) + 2.0 * 3.0 * 3.0 * C * C) + 2.0 * 3.0 * 4.0 * C * C) * cols;
        const double ae = P.storage == 1 ? 2.0 : 4.0;
        const double bytes = (2.0 * 4.0 + 2.0 * C * ae + 2.0 * C * ae / d1.scale) * cols +
                             4.0 * (double)(2 * (d.c2[0].w_floats + d.c3[0].w_floats + d.film[0].w_floats) + d.heads.w_floats);
        hipError_t e = prof->begin(stream, "cond.0", q.small == 2 ? (P.storage == 1 ? "cond_stage0_pipe<x1>" : "cond_stage0_pipe<x3>")
                                                                 : (P.storage == 1 ? "cond_stage0<x1>" : "cond_stage0<x3>"), flops, bytes);
        if (e != hipSuccess) return e;
        e = launch();
        if (e != hipSuccess) return e;
        e = prof->end();
        if (trace_env && trace_buf) {
            unsigned long long h[4 * 64];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h, trace_buf, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 4; ++w) {
                std::fprintf(stderr, "[cond trace] wave %d:", w);
                for (int i = 1; i < 64 && h[w * 64 + i]; ++i) std::fprintf(stderr, " %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
                std::fprintf(stderr, "\n");
            }
        }
        return e;
    }
    return launch();
}

// Stage 1 of the conditioning nets (C = 48) as ONE launch (fastsvc_cond.hip): the compact decimated output of stage 0 ->
// ss.1 and the compact h_1[..., ::s_2].  `done` = false when this call has no such variant.
hipError_t run_cond_stage1(const fastsvc_plan& P, const float* blob, const float* x, int B, long T1, int F, const int* lengths,
                           float* ss, float* hd, const float* amax_in, float* amax_hd, hipStream_t stream, Profiler* prof,
                           bool& done) {
    done = false;
    static const int tpw_env = std::getenv("FASTSVC_COND1_TPW") ? std::atoi(std::getenv("FASTSVC_COND1_TPW")) : 0;
    if (!cond_stage_whole(P, 1, F) || g_exact_f32 || (P.storage == 0 && (!amax_in || !amax_hd))) return hipSuccess;
    const DownStage& d = P.down[1];
    const DownStage& d2 = P.down[2];
    const int prec = P.storage == 1 ? 1 : 0;
    CondStage1Params q;
    std::memset(&q, 0, sizeof(q));
    q.x = x; q.ldx = (int)T1; q.x_b = (long)d.Cin * T1; q.x_sig = (long)B * q.x_b;
    q.B = B; q.C = d.C; q.Cin = d.Cin; q.T = (int)T1; q.ld = (int)T1;
    q.lens = lengths; q.len_mul = (int)(T1 / F);
    const PackedConv* layers[3] = {d.c2, d.c3, d.film};
    for (int s = 0; s < 2; ++s) {
        q.w1[s] = blob + d.rc1[s].hx_off[prec]; q.b1[s] = blob + d.rc1[s].b_off; q.br[s] = blob + d.rc1[s].b2_off;
        q.winv1[s] = blob + d.rc1[s].hx_inv_off;
        for (int l = 0; l < 3; ++l) {
            q.w[l][s] = blob + layers[l][s].hx_off[prec];
            q.bias[l][s] = blob + layers[l][s].b_off;
            q.winv[l][s] = blob + layers[l][s].hx_inv_off;
            q.bnd[l + 1][s] = blob + layers[l][s].bnd_off;
        }
        q.bnd[0][s] = blob + d.c1[s].bnd_off;
        q.bnd_r[s] = blob + d.r[s].bnd_off;
        q.cbnd[s] = d.cbnd_off[s] ? blob + d.cbnd_off[s] : nullptr;
    }
    q.w5 = blob + d.heads.hx_off[prec]; q.b5 = blob + d.heads.b_off; q.winv5 = blob + d.heads.hx_inv_off;
    q.amax_in = P.storage == 0 ? amax_in : nullptr; q.amax_hd = P.storage == 0 ? amax_hd : nullptr;
    q.ss = ss; q.ss_b = 2L * d.C * T1;
    q.hd = hd; q.hd_ld = (int)(T1 / d2.scale); q.hd_s = d2.scale;
    q.hd_b = (long)d.C * q.hd_ld; q.hd_sig = (long)B * q.hd_b;
    // short tiles where the long ones would leave CUs without a workgroup (a batch below one round of them)
    const long slots_all = P.storage == 1 ? 512 : 256;
    static const int small_env = std::getenv("FASTSVC_COND_SMALL") ? std::atoi(std::getenv("FASTSVC_COND_SMALL")) : -1;
    {
        const int NTb = cond_stage1_tile_columns(0);
        q.small = small_env >= 0 ? small_env : (((T1 + NTb - 1) / NTb) * B < slots_all ? 1 : 0);
    }
    // the layer pipeline (cond_stage1_pipe_kernel, bfloat16 storage only; q.small = 2, tpw = 32-column chunks per workgroup):
    // its fill is 11 steps, so only for long runs; "cond.1|B|T1|b" in the launch table forces either (algorithm 4 / 5)
    static const int pipe_env = std::getenv("FASTSVC_COND_PIPE") ? std::atoi(std::getenv("FASTSVC_COND_PIPE")) : 1;
    int pipe_mode = P.storage == 1 ? pipe_env : 0;
    if (P.storage == 1) {
        char key[96];
        std::snprintf(key, sizeof(key), "cond.1|%d|%ld|b", B, (long)T1);
        std::lock_guard<std::mutex> lock(P.tune_mu);
        auto it = P.tuned.find(key);
        if (it != P.tuned.end() && it->second.algo == 4) pipe_mode = 0;
        if (it != P.tuned.end() && it->second.algo == 5) pipe_mode = 2;
    }
    if (pipe_mode && small_env < 0) {
        const int NTp = cond_stage1_tile_columns(2);
        const long nchunks = (T1 + NTp - 1) / NTp;
        long kc = 0;
        {
            const long per_utt = std::max<long>(1, 256L / B);
            const long k = (nchunks + per_utt - 1) / per_utt;
            if (k >= 16 || pipe_mode == 2) kc = k;          // (8 x 600 frames, 19 chunks each: 58 vs 94 us for the phase kernel)
        }
        const bool fits = ((long)B * q.hd_b + (long)d.C * q.hd_ld) * 2L < (1L << 31) && (long)d.Cin * T1 * 2L < (1L << 31);
        if (kc && fits) { q.small = 2; q.tpw = (int)std::min<long>(tpw_env > 0 ? tpw_env : kc, 0xffff); }
    }
    const int NT = q.small == 2 ? 0 : cond_stage1_tile_columns(q.small);
    const long ntx = q.small == 2 ? 0 : (T1 + NT - 1) / NT;
    if (q.small == 2) {}
    else if (tpw_env > 0) q.tpw = tpw_env;
    else {
        const long slots = P.storage == 1 ? 512 : 256, total = ntx * B;
        const long rounds = std::max<long>(1, (total + slots * 32 - 1) / (slots * 32));
        long tpw = std::max<long>(1, (total + slots * rounds - 1) / (slots * rounds));
        while (tpw < ntx && ((ntx + tpw - 1) / tpw) * B > slots * rounds) ++tpw;
        q.tpw = (int)std::min<long>(tpw, ntx);
    }
    done = true;
    auto launch = [&]() { return P.storage == 1 ? bf16::launch_cond_stage1(q, stream) : launch_cond_stage1(q, stream); };
    if (prof) {
        const double C = d.C, Ci = d.Cin, cols = (double)T1 * B;
        const double flops = (2.0 * (2.0 * (3.0 * Ci * C + Ci * C) + 2.0 * 3.0 * 3.0 * C * C) + 2.0 * 3.0 * 4.0 * C * C) * cols;
        const double ae = P.storage == 1 ? 2.0 : 4.0;
        const double bytes = (2.0 * Ci * ae + 2.0 * C * ae + 2.0 * C * ae / d2.scale) * cols +
                             4.0 * (double)(2 * (d.rc1[0].w_floats + d.c2[0].w_floats + d.c3[0].w_floats + d.film[0].w_floats) + d.heads.w_floats);
        hipError_t e = prof->begin(stream, "cond.1", q.small == 2 ? "cond_stage1_pipe<x1>" : P.storage == 1 ? "cond_stage1<x1>" : "cond_stage1<x3>", flops, bytes);
        if (e != hipSuccess) return e;
        e = launch();
        if (e != hipSuccess) return e;
        return prof->end();
    }
    return launch();
}

hipError_t run_conv(const PackedConv& c, const float* blob, ConvParams p, int nsig, long pair_w_stride,
                    long pair_b_stride, hipStream_t stream, Profiler* prof,
                    const char* layer) {
    p.CIN = c.cin; p.KC = c.KC; p.nchunks = c.nchunks;
    p.w = blob + c.w_off; p.w_sig = pair_w_stride;
    const bool act_bf16 = g_tune.plan && g_tune.plan->storage == 1;
    auto launch = [&](const ConvParams& q, const ConvLaunch& Lq, hipStream_t st) {
        if (Lq.pipe == 3) return act_bf16 ? bf16::launch_conv_wx(q, Lq, st) : launch_conv_wx(q, Lq, st);
        if (Lq.pipe == 2) return act_bf16 ? bf16::launch_conv_hx(q, Lq, st) : launch_conv_hx(q, Lq, st);
        return act_bf16 ? bf16::launch_conv(q, Lq, st) : launch_conv(q, Lq, st);
    };
    static const bool no_poly = std::getenv("FASTSVC_NO_POLY") != nullptr;   // A/B switch
    const long T_out = p.T;                                  // output columns (accounting below)
    if (p.mode == MODE_STRETCH && c.poly && !no_poly && c.dil == 1 && (long)p.x_T * p.s == p.T) {
        // Stretch2d + conv at the INPUT rate (polyphase): tiles walk the input columns
        ConvParams q = p;
        q.mode = MODE_POLY; q.T = p.x_T; q.w = blob + c.wp_off;
        q.Q = c.Q; q.ngroups = c.ngroups; q.COUT = c.cout; q.ntaps = c.ntaps; q.dil = c.dil; q.vec = 1;
        const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
        const bool tail_fine = (p.x_T % 4) == 0 || conv_ws_tail_ok(c.MW, 1, MODE_POLY, aff ? 4 : 1, p.s);
        if (conv_pipe_supported(q) && tail_fine) { p.mode = MODE_POLY; p.T = p.x_T; p.w = q.w; }
    }
    const bool poly = p.mode == MODE_POLY;
    // row pitches; with a ragged batch the kernels take each utterance's own lengths from `lens`
    p.ldx = p.x_T;
    p.ldy = (int)T_out;
    if (p.lens) { p.len_mul = p.T / p.frames_ld; p.xlen_mul = p.x_T / p.frames_ld; }
    p.bias = blob + c.b_off; p.bias_sig = pair_b_stride;
    if (c.dec2) { p.bias2 = blob + c.b2_off; p.bias2_sig = c.b2_pair; }
    p.Q = c.Q; p.ngroups = c.ngroups; p.COUT = c.cout;
    p.ntaps = c.ntaps; p.dil = c.dil;
    p.vec = (p.T % 4 == 0) ? 1 : 0;
    {
        static const int dbg = std::getenv("FASTSVC_DBG") ? std::atoi(std::getenv("FASTSVC_DBG")) : 0;
        p.dbg = dbg;
    }
    const int halo = c.ntaps == 3 ? c.dil : 0;
    const long zb = (long)nsig * p.B;
    ConvLaunch L{c.MW, 4, 1, 4, nsig, 0};
    p.tpw = 1;
    if (conv_pipe_supported(p)) {
        // Tile shape and tiles-per-workgroup from a small cost model: a workgroup costs a fixed
        // start-up (launch, slot set-up, first window in flight) plus tpw * nchunks units, a unit
        // being bounded by the consumer wave's MFMA stream or by the producers' window traffic;
        // the grid runs in ceil(workgroups / resident slots) rounds.
        // compile-time epilogue kind of the launch (same rule as launch_conv_pipe)
        int epi_kind = 0;
        if (c.ntaps == 3 && (p.mode == MODE_DIRECT || p.mode == MODE_STRETCH || poly)) {
            const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
            epi_kind = aff ? 4 : ((p.mode == MODE_STRETCH || poly) ? 1 : p.r1x ? 3 : p.res ? 2 : 1);
        }
        // algo 0: as launched, 1: Winograd (MW of the layer), 2: Winograd MW = 2, 3: half-precision MFMA (fastsvc_hx.hip)
        struct Cand { int NW, WM, WN, algo; };
        auto is_wino = [](int algo) { return algo == 1 || algo == 2; };
        std::vector<Cand> cands;
        if (poly) {
            if (c.MW == 3 && c.ngroups % 2 == 0) cands = {{1, 2, 2}, {1, 1, 4}};
            else if (c.MW == 3) cands = {{1, 1, 4}};
            else cands = {{2, 1, 4}, {1, 1, 4}};
        }
        else if (c.MW == 3 && c.ngroups % 4 == 0) cands = {{4, 4, 1}, {2, 4, 1}, {4, 2, 2}, {2, 2, 2}, {1, 2, 2}, {4, 1, 4}, {2, 1, 4}, {1, 1, 4}};
        else if (c.MW == 3 && c.ngroups % 2 == 0) cands = {{4, 2, 2}, {2, 2, 2}, {1, 2, 2}, {4, 1, 4}, {2, 1, 4}, {1, 1, 4}};
        else cands = {{4, 1, 4}, {2, 1, 4}, {1, 1, 4}};
        if (p.mode == MODE_DEC2) {                      // two accumulator sets: compiled for NW <= 2
            std::vector<Cand> keep;
            for (const Cand& cd : cands) if (cd.NW <= 2) keep.push_back(cd);
            cands.swap(keep);
        }
        // Winograd F(2,3) variants of the same launch (fastsvc_kernels.h, MODE_WINO)
        static const int wino_env = std::getenv("FASTSVC_WINO") ? std::atoi(std::getenv("FASTSVC_WINO")) : -1;
        const bool wino_ok = c.wino && wino_env != 0 && p.mode == MODE_DIRECT && (!p.r1x || c.MW == 2) &&
                             !(p.flags & (F_STATS | F_AFF_OUT | F_PRE_AFFINE));
        if (wino_ok) {
            const size_t nd = cands.size();
            for (size_t i = 0; i < nd; ++i)
                if (cands[i].NW <= (c.MW == 2 ? 1 : 2)) cands.push_back(Cand{cands[i].NW, cands[i].WM, cands[i].WN, 1});
            if (c.wino2) {
                const int ng2 = c.cout / 32;
                cands.push_back(Cand{1, 1, 4, 2});
                if (ng2 % 2 == 0) cands.push_back(Cand{1, 2, 2, 2});
                if (ng2 % 4 == 0) cands.push_back(Cand{1, 4, 1, 2});
            }
        }
        // Half-precision MFMA variants (split-half f16 products in float32 storage, bf16 products in bfloat16
        // storage): FASTSVC_HX=0 switches them off (A/B), they need rows that are a multiple of 4 long
        static const int hx_env = std::getenv("FASTSVC_HX") ? std::atoi(std::getenv("FASTSVC_HX")) : 1;
        const bool hx_mode = (p.mode == MODE_DIRECT && p.x_T == p.T) || (p.mode == MODE_POLY && c.hxp_off[0]) ||
                             (p.mode == MODE_DEC2 && (p.x_T & 3) == 0);
        // ragged batch whose rows run at the frame rate or twice it (len_mul 1 or 2): an utterance's own row length
        // need not be a multiple of 4.  The TAILK instances handle that (producers zero what lies past the row end
        // after the prologue, epilogues keep it out of the sums) as long as the row PITCHES are a multiple of 4
        const bool ragged_tail = p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0);
        const bool hx_tail_ok = !ragged_tail || ((p.ldx & 3) == 0 && (p.ldy & 3) == 0 &&
                                                 conv_hx_tail_ok(p.mode, c.MW, epi_kind, poly ? p.s : 1));
        const bool hx_ok = hx_env != 0 && !p.no_hx && !g_exact_f32 && c.hx && hx_mode && c.ntaps == 3 && (p.T & 3) == 0 &&
                           c.dil <= 28 && !(p.flags & F_PRE_AFFINE) && hx_tail_ok;
        if (hx_ok) {
            static const int shapes[][3] = {{8, 4, 1}, {4, 4, 1}, {6, 2, 2}, {4, 2, 2}, {2, 2, 2}, {3, 1, 4}, {2, 1, 4}};
            for (const auto& sh : shapes)
                if (conv_hx_shape(p.mode, c.MW, sh[0], sh[1], sh[2]) && c.ngroups % sh[1] == 0 &&
                    !(p.mode == MODE_DIRECT && epi_kind >= 3 && sh[0] * c.MW > 12))   // FiLM-affine / rank-1 epilogues: those tiles spill
                    cands.push_back(Cand{sh[0], sh[1], sh[2], 3});
        }
        // The wide-layer kernel (fastsvc_wx.hip, algorithm 6): every wave multiplies, weights through LDS.  Direct launches of
        // the 48-channel-group layers (C >= 96) in bfloat16 storage with a plain / residual / FiLM-affine epilogue.
        // FASTSVC_WX = 0: off, 2: wherever it exists (A/B, tests), else by the launch table / tuning
        static const int wx_env = std::getenv("FASTSVC_WX") ? std::atoi(std::getenv("FASTSVC_WX")) : 1;
        const bool wx_ok = hx_ok && wx_env != 0 && act_bf16 && p.mode == MODE_DIRECT && c.MW == 3 && !ragged_tail && !p.last_w &&
                           !p.r1x && !p.x2 && !p.xsplit && (epi_kind == 1 || epi_kind == 2 || epi_kind == 4);
        bool wx_have = false;                            // ... and the layer has a shape of it
        if (wx_ok) {
            static const int wshapes[][3] = {{6, 4, 2}};
            for (const auto& sh : wshapes)
                if (conv_wx_shape(p.mode, c.MW, sh[0], sh[1], sh[2]) && c.ngroups % sh[1] == 0 && conv_wx_fits(c.nch32, c.dil)) {
                    cands.push_back(Cand{sh[0], sh[1], sh[2], 6});
                    wx_have = true;
                }
        }
        const bool wx_force = wx_env == 2 && wx_have;
        if (p.last_w) {
            // conv_last rides on the half-precision instances with one workgroup row of channel groups; it saves a
            // launch and a write + read of the block's output, more than any shape of the other family wins back:
            // only those candidates, timed as they will run (no store of the C-channel tensor)
            std::vector<Cand> keep;
            if (hx_ok && c.ngroups == 1 && p.mode == MODE_DIRECT && p.res && !p.r1x && !(p.flags & (F_STATS | F_AFF_OUT)))
                for (const Cand& cd : cands) if (cd.algo == 3 && cd.WM == 1) keep.push_back(cd);
            if (!keep.empty()) { cands.swap(keep); p.y = nullptr; }
            else p.last_w = nullptr;
        }
        if ((p.T & 3) != 0 || (p.lens && (p.len_mul & 3) != 0)) {
            // rows that are not a multiple of 4 long (with a ragged batch: ANY utterance's own length,
            // whatever the padded maximum is): only the variants compiled with the row-end handling
            std::vector<Cand> keep;
            for (const Cand& cd : cands)
                if (cd.algo == 6 ? false : cd.algo == 3 ? (p.T & 3) == 0 :            // (hx candidates are only there when hx_tail_ok)
                    conv_ws_tail_ok(cd.algo == 2 ? 2 : c.MW, cd.NW, is_wino(cd.algo) ? (int)MODE_WINO : p.mode, epi_kind,
                                    poly ? p.s : 1)) keep.push_back(cd);
            if (!keep.empty()) cands.swap(keep);       // (never empty: NW <= 2 variants always have it)
        }
        auto round_to = [](int v, int rem, int mod) { int r = v / mod * mod + rem; return r < v ? r + mod : r; };
        // LDS geometry of a candidate: row stride (== 16 mod 32) and, for Winograd, the phase-plane
        // stride that keeps the component reads conflict-free (planes land 16/D banks apart)
        auto geometry = [&](const Cand& cd, ConvParams& q) {
            if (cd.algo == 3 || cd.algo == 6) {
                const int prec = act_bf16 ? 1 : 0;
                q.whx = blob + (p.mode == MODE_POLY ? c.hxp_off[prec] : c.hx_off[prec]); q.whx_sig = c.hx_pair[prec]; q.nch32 = c.nch32;
                const size_t inv_off = p.mode == MODE_POLY ? c.hxp_inv_off : c.hx_inv_off;
                q.whx_inv = (prec == 0 && inv_off) ? blob + inv_off : nullptr; q.whx_inv_sig = c.hx_inv_pair;
                static const int stagger = std::getenv("FASTSVC_STAGGER") ? std::atoi(std::getenv("FASTSVC_STAGGER")) : 2;
                // (conv_wx: start delay between the workgroups of an XCD, in 64-cycle steps per K chunk - an eighth of a unit)
                q.stagger = stagger;
                q.xs = 0; q.ps = 0;
            } else if (cd.algo >= 1) {
                const int NTo = 32 * cd.NW * cd.WN, D = c.dil;
                const int Q = (NTo + 16) / (2 * D);
                const int ps = D == 1 ? (Q + 1) / 2 * 2 : D == 2 ? round_to(Q, 8, 32) : round_to(Q, 12, 32);
                q.ps = ps;
                q.xs = round_to(2 * D * ps, 16, 32);
                q.mode = MODE_WINO; q.Q = c.Qw;
                if (cd.algo == 2) { q.w = blob + c.ww2_off; q.w_sig = c.ww2_pair; q.ngroups = c.cout / 32; }
                else { q.w = blob + c.ww_off; q.w_sig = c.ww_pair; }
            } else {
                const int W = 16 * cd.NW * cd.WN + 2 * ((halo + 3) & ~3);
                q.xs = (W + 15) / 32 * 32 + 16;
                q.ps = 0;
            }
        };
        char key[96];
        // bfloat16 storage compiles some variants under a different register budget: its own entries ("|b")
        std::snprintf(key, sizeof(key), act_bf16 ? "%s|%d|%d|b" : "%s|%d|%d", layer, p.B, p.T);
        bool have = false;
        Cand best = cands[0];
        if (g_tune.plan) {
            std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
            auto it = g_tune.plan->tuned.find(key);
            if (it != g_tune.plan->tuned.end()) {
                // a loaded table may be stale: only shapes this launch is compiled for are taken
                for (const Cand& cd : cands)
                    if (cd.NW == it->second.NW && cd.WM == it->second.WM && cd.WN == it->second.WN &&
                        cd.algo == it->second.algo && it->second.tpw >= 1 && it->second.tpw <= 64 &&
                        !(hx_env == 2 && hx_ok && cd.algo != 3 && cd.algo != 6) &&   // FASTSVC_HX=2: older tables do not hold it back
                        !(wx_force && cd.algo != 6)) {
                        best = cd; p.tpw = it->second.tpw;
                        have = true;
                    }
            }
        }
        if (!have && g_tune.tuning && g_tune.plan) {
            // time every (shape, tiles-per-workgroup) on the device; InstanceNorm sums are not
            // accumulated by the trial launches (the real launch below does that once)
            // (fine steps matter: the winner is often the count that fills the last round of workgroups best)
            static const int tpws[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24};
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return hipErrorUnknown;
            float best_ms = 1e30f;
            ConvParams q = p;
            q.flags &= ~F_STATS;
            for (const Cand& cd : cands) {
                const int NT = (is_wino(cd.algo) ? 32 : 16) * cd.NW * cd.WN;
                const long ntx = (p.T + NT - 1) / NT;
                if (wx_force && cd.algo != 6) continue;                   // FASTSVC_WX=2: only the wide-layer kernel's shapes
                q = p; q.flags &= ~F_STATS;
                geometry(cd, q);
                for (int tpw : tpws) {
                    q.tpw = tpw;
                    ConvLaunch Lq{cd.algo == 2 ? 2 : c.MW, cd.NW, cd.WM, cd.WN, nsig, cd.algo == 6 ? 3 : cd.algo == 3 ? 2 : 1};
                    hipError_t e = launch(q, Lq, stream);                 // warm
                    if (e != hipSuccess && cd.algo == 6) break;           // (this shape does not fit the layer's LDS budget)
                    if (e != hipSuccess) return e;
                    hipEventRecord(e0, stream);
                    for (int r = 0; r < 3; ++r) { e = launch(q, Lq, stream); if (e != hipSuccess) return e; }
                    hipEventRecord(e1, stream);
                    if (hipEventSynchronize(e1) != hipSuccess) return hipErrorUnknown;
                    float ms = 0.f;
                    hipEventElapsedTime(&ms, e0, e1);
                    ++g_tune.trials;
                    if (ms < best_ms) { best_ms = ms; best = cd; p.tpw = tpw; }
                    if (tpw >= ntx) break;                                // larger tpw changes nothing
                }
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
            std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
            g_tune.plan->tuned[key] = fastsvc_plan::Choice{best.NW, best.WM, best.WN, p.tpw, best.algo};
            g_tune.plan->priors.clear();
            have = true;
        }
        // No entry for this (B, T): the tile shape and algorithm of the table entry of this layer with the nearest
        // problem size, where one exists and this launch is compiled for it - a layer's best shape moves slowly with
        // the size (the tuned tables agree on it across batch shapes far more often than the model below finds it:
        // tools/costmodel_gap.py) - and only the tiles per workgroup from the model.
        bool prior_pick = false;
        static const int prior_env = std::getenv("FASTSVC_PRIOR") ? std::atoi(std::getenv("FASTSVC_PRIOR")) : 1;
        if (!have && g_tune.plan && prior_env && !wx_force) {
            fastsvc_plan::Choice pr{0, 0, 0, 0, -1};
            {
                std::lock_guard<std::mutex> lock(g_tune.plan->tune_mu);
                auto it = g_tune.plan->priors.find(key);
                if (it != g_tune.plan->priors.end()) pr = it->second;
                else {
                    if (!g_tune.plan->prior_for(layer, p.B, p.T, act_bf16, pr)) pr.algo = -1;
                    g_tune.plan->priors[key] = pr;
                }
            }
            if (pr.algo >= 0)
                for (const Cand& cd : cands)
                    if (cd.NW == pr.NW && cd.WM == pr.WM && cd.WN == pr.WN && cd.algo == pr.algo) {
                        const Cand only = cd;
                        cands.assign(1, only);
                        prior_pick = true;
                        break;
                    }
        }
        if (!have) {
        // Tile shape and tiles-per-workgroup from a small cost model: a workgroup costs a fixed
        // start-up (launch, slot set-up, first window in flight) plus tpw * nchunks units, a unit
        // being bounded by the consumer wave's MFMA stream or by the producers' window traffic;
        // the grid runs in ceil(workgroups / resident slots) rounds.
        static const double startup_us = std::getenv("FASTSVC_STARTUP_US") ? std::atof(std::getenv("FASTSVC_STARTUP_US")) : 4.0;
        double best_t = 1e30;
        for (const Cand& cd : cands) {
            // Without a table entry the model takes Winograd wherever the layer has it, the 32-channel
            // grouping (two workgroups per CU) when that exists - measured better than direct on cfg1
            // and cfg2 (0.56 vs 0.62 ms, 2.24 vs 2.32 ms).  C = 24 layers (MW == 2) stay direct: their
            // Winograd variant loses to the scalar phase-plane staging.  FASTSVC_WINO = 0 / 1 / 2
            // forces direct / layer grouping / 32-channel grouping.
            const int wino_pref = wino_env >= 0 ? wino_env : (c.MW == 2 ? 0 : 2);
            if (cd.algo == 6) {
                // no model of its own: taken by table / prior, or wherever it exists with FASTSVC_WX=2 (first shape that fits,
                // one round of workgroups where the batch allows)
                if (!(prior_pick || wx_force)) continue;
                const int NT = 16 * cd.NW * cd.WN;
                const long ntx = (p.T + NT - 1) / NT;
                const long gy = c.ngroups / cd.WM;
                int tpw = 1;
                while (tpw < 64 && ((ntx + tpw - 1) / tpw) * gy * zb > 256) ++tpw;
                if (best_t > 0.0) { best_t = 0.0; best = cd; p.tpw = tpw; }
                continue;
            }
            if (wx_force) continue;
            if (!prior_pick && hx_ok != (cd.algo == 3)) continue;          // the half-precision MFMA variant wherever the layer has one
            if (cd.algo == 3) {
                // data-movement model: one workgroup per CU; a unit moves its window in and (per tile) its
                // outputs; the matrix work (3 taps x NW x MW x 1|3 products of 16 cycles) hides under it
                const int NT = 16 * cd.NW * cd.WN;
                const long ntx = (p.T + NT - 1) / NT;
                const long gy = c.ngroups / cd.WM;
                const double win = NT + 2.0 * ((halo + 3) & ~3);
                const double esz = act_bf16 ? 2.0 : 4.0;
                const double out_streams = (p.y ? 1 : 0) + ((p.flags & F_AFF_OUT) ? 3 : 0) + (p.res ? 1 : 0);
                const double bytes_unit = 32.0 * win * esz + 16.0 * c.MW * cd.WM * NT * (poly ? p.s : 1) * esz *
                                          (out_streams + (p.mode == MODE_DEC2 ? 1 : 0)) / c.nch32;
                const double mem_us = bytes_unit / (4.5e6 / 256.0);
                const double mfma_us = (poly ? 5.0 : p.mode == MODE_DEC2 ? 4.0 : 3.0) * cd.NW * c.MW * (act_bf16 ? 1 : 3) * 17.0 / 2.0e3;
                const double unit_us = mem_us > mfma_us ? mem_us : mfma_us;
                for (int tpw = 1; tpw <= 24; ++tpw) {
                    const long wgs = ((ntx + tpw - 1) / tpw) * gy * zb;
                    const long rounds = (wgs + 255) / 256;
                    const double t = rounds * (startup_us + tpw * c.nch32 * unit_us);
                    if (t < best_t * 0.999) { best_t = t; best = cd; p.tpw = tpw; }
                }
                continue;
            }
            if (!prior_pick) {
                if ((cd.algo >= 1) != (wino_ok && wino_pref >= 1)) continue;
                if (wino_pref >= 1 && wino_ok && c.wino2 && (cd.algo == 2) != (wino_pref == 2)) continue;
            }
            const int NT = (cd.algo >= 1 ? 32 : 16) * cd.NW * cd.WN;
            const int cMW = cd.algo == 2 ? 2 : c.MW;
            const long ntx = (p.T + NT - 1) / NT;
            const long gy = ((cd.algo == 2 ? c.cout / 32 : c.ngroups) + cd.WM - 1) / cd.WM;
            const int resident = conv_ws_resident(cMW, cd.NW, cd.algo >= 1 ? (int)MODE_WINO : p.mode, epi_kind);   // VGPR budget, see the kernel
            const long slots = 256L * resident;
            // unit time: MFMA stream of one consumer wave vs bytes the 256 producer threads move
            const double mfma_us = 6.0 * (cd.algo >= 1 ? 4 : c.ntaps) * cd.NW * cMW * 32.0 / 2.2e3;
            const double win = NT + 2.0 * ((halo + 3) & ~3);
            const double bytes_unit = c.KC * win * 4.0 + (double)16 * cMW * cd.WM * NT * (poly ? p.s : 1) * 4.0 *
                                      ((p.y ? 1 : 0) + ((p.flags & F_AFF_OUT) ? 3 : 0) + (p.res ? 1 : 0)) / c.nchunks;
            const double mem_us = bytes_unit / (5.0e6 / 256.0 / resident);    // ~5 TB/s shared by all slots
            // packed weights streamed from L2 by the four consumer waves (per unit, per workgroup):
            // ~16 TB/s of L2 shared by all resident workgroups; this is what punishes small NW
            const double wbytes_unit = 4.0 * 6.0 * c.ntaps * 64.0 * cMW * 4.0;
            const double l2_us = wbytes_unit / (16.0e6 / 256.0 / resident);
            double unit_us = mfma_us > mem_us ? mfma_us : mem_us;
            if (l2_us > unit_us) unit_us = l2_us;
            unit_us /= (resident == 2 ? 1.6 : 1.0);
            for (int tpw = 1; tpw <= 16; ++tpw) {
                const long wgs = ((ntx + tpw - 1) / tpw) * gy * zb;
                const long rounds = (wgs + slots - 1) / slots;
                const double t = rounds * (startup_us + tpw * c.nchunks * unit_us);
                if (t < best_t * 0.999) { best_t = t; best = cd; p.tpw = tpw; }
            }
        }
        }
        L.pipe = best.algo == 6 ? 3 : best.algo == 3 ? 2 : 1;
        L.NW = best.NW; L.WM = best.WM; L.WN = best.WN;
        if (p.last_w) g_last_fused = true;                  // (the candidates were restricted to the instances that have it)
        if (best.algo == 2) L.MW = 2;
        geometry(best, p);
        {
            static const bool verbose = std::getenv("FASTSVC_VERBOSE_SHAPES") != nullptr;   // tuning aid
            if (verbose && prof)
                std::fprintf(stderr, "[fastsvc] %-20s mode %d  MW %d NW %d WM %d WN %d tpw %d\n", layer, p.mode,
                             L.MW, L.NW, L.WM, L.WN, p.tpw);
        }
    } else {
        p.last_w = nullptr;
        int NW = 4;
        while (NW > 1 && ((p.T + 64 * NW - 1) / (64 * NW)) * zb * c.ngroups < 768) NW >>= 1;
        L.NW = NW;
        const int W = 64 * NW + 2 * halo;
        p.xs = (W + 15) / 32 * 32 + 16;
    }
    if (prof) {
        // algorithmic work of the layer as the reference defines it (3 taps at the OUTPUT rate for
        // the stretched convs, whichever way they are computed)
        const double cols = (double)T_out * p.B * nsig;
        const double flops = 2.0 * (c.dec2 ? 4 : c.ntaps) * c.cin * c.cout * cols;    // DEC2: k=3 conv + 1x1
        double in_cols = (double)T_out;                     // source columns actually needed
        if (p.mode == MODE_STRETCH || poly) in_cols = (double)p.x_T;
        double el = (double)c.cin * in_cols;
        if (p.y) el += (double)c.cout * T_out;
        if (p.mode == MODE_DEC2 && p.y2) el += (double)c.cout * T_out;     // the decimating pair's second output (the 1x1 residual conv's)
        if (p.flags & F_AFF_OUT) el += (double)c.cout * T_out;
        if (p.flags & F_PRE_AFFINE) el += 2.0 * c.cin * T_out;
        if (p.res) el += (double)c.cout * T_out;
        if (p.r1x) el += (double)T_out;
        if (p.flags & (F_STATS | F_AFF_OUT)) el += 2.0 * c.cout * T_out;
        double bytes = (act_bf16 ? 2.0 : 4.0) * el * p.B * nsig + 4.0 * (double)(c.w_floats + c.b_floats) * nsig;
        double flops_all = flops;
        if (p.last_w) {                                   // + conv_last: C -> 1, float32 waveform out (the C-channel tensor is not written: p.y is null)
            flops_all += 2.0 * c.cout * cols;
            bytes += 4.0 * cols;
        }
        char kname[40];
        if (L.pipe == 3) {
            const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
            std::snprintf(kname, sizeof(kname), "conv_wx<%d,%d,%d,%d,%d,%s>", L.MW, L.NW, L.WM, L.WN, aff ? 4 : p.res ? 2 : 1, act_bf16 ? "x1" : "x3");
        } else if (L.pipe == 2) {
            const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
            const int kind = aff ? 4 : p.r1x ? 3 : p.res ? 2 : 1;
            const bool tail_inst = p.lens && (((p.len_mul | p.xlen_mul) & 3) != 0);      // the row-end (TAILK) instance
            std::snprintf(kname, sizeof(kname), "conv_hx<%d,%d,%d,%d,%d,%d,%d,%s%s>", L.MW, L.NW, L.WM, L.WN, p.mode,
                          p.mode == MODE_DEC2 ? 1 : poly ? (aff ? 4 : 1) : kind, poly ? p.s : 1, act_bf16 ? "x1" : "x3",
                          tail_inst ? ",tail" : "");
        } else if (L.pipe)
        {
            // same rule as launch_conv_pipe: compile-time epilogue kind of the 3-tap DIRECT / STRETCH launches
            int kind = 0;
            if (c.ntaps == 3 && (p.mode == MODE_DIRECT || p.mode == MODE_STRETCH || poly || p.mode == MODE_WINO)) {
                const bool aff = (p.flags & (F_STATS | F_AFF_OUT)) != 0;
                kind = aff ? 4 : ((p.mode == MODE_STRETCH || poly) ? 1 : p.r1x ? 3 : p.res ? 2 : 1);
            }
            // the template arguments of the instance that runs (last one: polyphase stretch / Winograd dilation)
            std::snprintf(kname, sizeof(kname), "conv_mfma_ws<%d,%d,%d,%d,%d,%d,%d,%d>", L.MW, L.NW, L.WM, L.WN,
                          p.mode, c.ntaps, kind, poly ? p.s : p.mode == MODE_WINO ? c.dil : 1);
        }
        else
            std::snprintf(kname, sizeof(kname), "conv_mfma<%d,%d,1,4>", L.MW, L.NW);
        hipError_t e = prof->begin(stream, layer, kname, flops_all, bytes);
        if (e != hipSuccess) return e;
        e = launch(p, L, stream);
        if (e != hipSuccess) return e;
        return prof->end();
    }
    if (act_bf16 && !L.pipe) return hipErrorNotSupported;   // the scalar kernel exists for float32 storage only
#ifdef FASTSVC_TIMELINE
    if (L.pipe) {
        const int NT = (p.mode == MODE_WINO ? 32 : 16) * L.NW * L.WN;
        const long ntx = (p.T + NT - 1) / NT;
        const long wgs = ((ntx + p.tpw - 1) / p.tpw) * ((p.ngroups + L.WM - 1) / L.WM) * zb;
        hipError_t e = hipSuccess;
        if (timeline_launch(layer, p, wgs, p.nchunks, L, stream, [&](const ConvParams& q) { return launch(q, L, stream); }, e)) return e;
    }
#endif
    return launch(p, L, stream);
}

}  // namespace

extern "C" {

int fastsvc_plan_set_workspace_mode(fastsvc_plan* plan, int32_t compact) {
    if (!plan) return fail(FASTSVC_E_INVALID, "null plan");
    plan->compact = compact != 0;
    return FASTSVC_OK;
}

size_t fastsvc_workspace_bytes(const fastsvc_plan* plan, int32_t B, int32_t F) {
    if (!plan || B < 1 || F < 1) return 0;
    return layout_workspace(*plan, B, F).bytes;
}

int fastsvc_workspace_tap(const fastsvc_plan* plan, int32_t B, int32_t F, const char* tap_name,
                          size_t* byte_offset, int64_t* numel, int64_t shape3[3]) {
    if (!plan || !tap_name || !byte_offset || !numel || !shape3) return fail(FASTSVC_E_INVALID, "null argument");
    const Workspace ws = layout_workspace(*plan, B, F);
    if (std::strncmp(tap_name, "amax:", 5) == 0) {
        // the amax row of a workspace tensor (float32 storage): 2B floats, [signal][utterance] for the conditioning
        // chains' tensors, the first B otherwise
        const BufferSpec* t = ws.find(tap_name + 5);
        const BufferSpec* a = ws.find("amax");
        if (!t || !a) return fail(FASTSVC_E_INVALID, std::string("unknown tap: ") + tap_name);
        *byte_offset = a->off_bytes + (size_t)(t - ws.bufs.data()) * 2 * B * 256 * sizeof(float);
        *numel = 2 * B * 256;
        shape3[0] = 2 * B; shape3[1] = 256; shape3[2] = 1;
        return FASTSVC_OK;
    }
    const BufferSpec* b = ws.find(tap_name);
    if (!b) return fail(FASTSVC_E_INVALID, std::string("unknown tap: ") + tap_name);
    *byte_offset = b->off_bytes;
    *numel = b->numel;
    for (int i = 0; i < 3; ++i) shape3[i] = b->shape[i];
    return FASTSVC_OK;
}

int fastsvc_forward_launch_count(const fastsvc_plan* plan, int32_t with_spk_emb) {
    if (!plan) return 0;
    const int n = plan->n;
    // kernels only: [speaker projection] + down stage 0 (3) + stages >= 1 (3 each with the fused
    // c1 + 1x1 launch, else 4) + FiLM (2 per stage) + up blocks (6 each) + conv_last.
    // An upper bound: where a call's shapes and the launch table allow it, a down stage's c2 -> c3 pair is one
    // launch (MODE_CHAIN) and stage 0 is one launch altogether (MODE_CHAIN1).
    static const bool no_dec2 = std::getenv("FASTSVC_NO_DEC2") != nullptr;
    int down = 3;
    for (int k = 1; k < n; ++k) down += (plan->down[k].rc1[0].dec2 && !no_dec2) ? 3 : 4;
    return (with_spk_emb ? 1 : 0) + (plan->storage == 0 ? 1 : 0) + down + 2 * n + 6 * n + 1;     // (+ the input scan of float32 storage)
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(FASTSVC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

static int forward_impl(const fastsvc_plan* plan, const void* dev_blob,
                    const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                    float* out, int32_t B, int32_t F, const int32_t* lengths,
                    void* workspace, size_t workspace_bytes, void* stream_, Profiler* prof) {
    if (!plan || !dev_blob || !ppg || !sine || !lft || !out || !workspace)
        return fail(FASTSVC_E_INVALID, "null argument");
    if (B < 1 || F < 1) return fail(FASTSVC_E_INVALID, "B and F must be >= 1");
    const fastsvc_plan& P = *plan;
    if (P.storage == 1) {
        if (F % 4 != 0) return fail(FASTSVC_E_UNSUPPORTED, "bfloat16 storage needs a frame count that is a multiple of 4");
        for (int k = 0; k < P.n; ++k)
            if (P.down[k].c2[0].KC != 24 || P.up[k].d3.KC != 24 || P.up[k].first.KC != 24)
                return fail(FASTSVC_E_UNSUPPORTED, "bfloat16 storage needs the pipelined kernels (24-channel K chunks)");
        if (P.n > 1 && !P.down[1].rc1[0].dec2) return fail(FASTSVC_E_UNSUPPORTED, "bfloat16 storage needs the fused decimating pair");
    }
    if (spk_emb && !P.cfg.use_spk_emb)
        return fail(FASTSVC_E_INVALID, "spk_emb given but the generator was built with use_spk_emb=False");
    const Workspace ws = layout_workspace(P, B, F);
    if (workspace_bytes < ws.bytes) return fail(FASTSVC_E_WORKSPACE, "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    g_tune.plan = plan;
    g_exact_f32 = plan->storage == 0 && F <= 4;
    // Scheduling: the FiLM nets of stages 0..n-2 + the speaker projections can run on a lowest-priority
    // helper stream (mask bit 1; bit 0: the 1x1 / stretch residual convs, an experiment that never paid).  With the
    // fused launches that is worth ~2 % at cfg2 and nothing at cfg3 / cfg1 - and only where a fork + join through the
    // helper stream is as cheap as it should be, which ExecCtx measures once (measure_fork_join): with helper streams
    // created after an RCCL communicator the same schedule ran cfg2 in 2.54 ms instead of 1.43.  Default: bit 1 from
    // ~1.5e5 samples per call where the context says it pays, one stream otherwise; FASTSVC_STREAMS=<mask> forces a
    // mask (0: one stream), FASTSVC_SERIAL one stream.
    static const bool serial = std::getenv("FASTSVC_SERIAL") != nullptr;
    static const int streams_env = std::getenv("FASTSVC_STREAMS") ? std::atoi(std::getenv("FASTSVC_STREAMS")) : -1;
    int64_t hop_all = 1;
    for (int i = 0; i < plan->n; ++i) hop_all *= plan->cfg.upsampling_scales[i];
    int streams_mask = serial ? 0 : streams_env >= 0 ? (streams_env & 3) : ((int64_t)B * F * hop_all >= 150000 ? 2 : 0);
    // per-launch profiling and autotuning run on ONE stream so that every kernel is timed alone
    ExecCtx* ctx = (streams_mask == 0 || g_tune.tuning || prof) ? nullptr : exec_ctx_for(stream);
    std::unique_lock<std::mutex> ctx_lock;
    if (ctx) {
        ctx_lock = std::unique_lock<std::mutex>(ctx->busy);
        ensure_measured(stream, ctx);
        if (streams_env < 0 && !ctx->pays) { ctx_lock.unlock(); ctx_lock = std::unique_lock<std::mutex>(); ctx = nullptr; streams_mask = 0; }
    }
    hipStream_t s_film = (ctx && (streams_mask & 2)) ? ctx->aux[0] : stream;      // FiLM nets of stages 0..n-2
    hipStream_t s_side = (ctx && (streams_mask & 1)) ? ctx->aux[1] : stream;      // 1x1 / stretch residual convs
    int evi = 0;
    // `to` waits for everything enqueued on `from` so far
    auto order_after = [&](hipStream_t from, hipStream_t to) -> hipError_t {
        if (from == to || !ctx) return hipSuccess;
        hipEvent_t e = ctx->ev[evi++ % ctx->ev.size()];
        hipError_t r = hipEventRecord(e, from);
        if (r != hipSuccess) return r;
        return hipStreamWaitEvent(to, e, 0);
    };
    std::vector<hipEvent_t> ss_ready(P.n, nullptr);       // FiLM output of stage k is complete
    const float* blob = static_cast<const float*>(dev_blob);
    unsigned char* wsb = static_cast<unsigned char*>(workspace);
    auto buf = [&](const std::string& name) -> float* {
        return reinterpret_cast<float*>(wsb + ws.find(name)->off_bytes);
    };
    const int n = P.n;
    int64_t hop = 1;
    for (int i = 0; i < n; ++i) hop *= P.cfg.upsampling_scales[i];
    const long T = (long)hop * F;
    {
        // The kernels address every tensor of ONE utterance through a 32-bit buffer descriptor and
        // 32-bit byte offsets ((row * pitch + t) * 4, 2*C rows for scale/shift): the largest of those
        // spans must stay below 2 GiB.  (C = 24 at the full rate: ~11 M samples = 7.7 min at 24 kHz.)
        long worst = 4L * T;
        long Tk = T;
        for (int k = 0; k < P.n; ++k) { Tk /= P.down[k].scale; worst = std::max(worst, 2L * P.down[k].C * Tk * 4); }   // film_u / ss: 2C rows
        long Ti = F;
        for (int i = 0; i < P.n; ++i) { Ti *= P.up[i].scale; worst = std::max(worst, 2L * P.up[i].C * Ti * 4); }
        worst = std::max(worst, (long)P.cfg.in_channels * F * 4);
        if (worst >= 0x7fffffffL)
            return fail(FASTSVC_E_UNSUPPORTED, "utterance too long for the 32-bit tensor descriptors of the kernels "
                                                "(largest per-utterance tensor must stay below 2 GiB): split it");
    }
    const bool spk = spk_emb != nullptr;
    // amax row of a workspace tensor (float32 storage; see layout_workspace)
    float* amax_inb = buf("amax_in");
    float* amaxb = buf("amax");
    auto am = [&](const std::string& name) -> float* {
        return amaxb + (size_t)(ws.find(name) - ws.bufs.data()) * 2 * B * 256;
    };

    // ---- raw signals: sig 0 = lft, sig 1 = sine, read in place (the dual-signal launches address signal
    // `sig` as base + sig * stride; the stride between the caller's two tensors is whatever it is) ----
    const float* sigbuf = lft;
    const std::ptrdiff_t sig_bytes = reinterpret_cast<const char*>(sine) - reinterpret_cast<const char*>(lft);
    if (sig_bytes % (std::ptrdiff_t)sizeof(float) != 0) return fail(FASTSVC_E_INVALID, "sine / lft must be 4-byte aligned");
    const long sig_stride = (long)(sig_bytes / (std::ptrdiff_t)sizeof(float));
    const bool no_scan = amax_scan_disabled();                     // A/B timing only: inputs then count as unit-scale
    if (P.storage == 0 && !no_scan) {
        // float32 storage: largest magnitudes of the inputs (scales of the split-binary16 staging) and zeroed amax
        // rows of the intermediates - ONE small launch, first thing on the caller's stream (everything else is
        // ordered behind it)
        const BufferSpec* ab = ws.find("amax");
        if (prof) HIP_TRY(prof->begin(stream, "amax_inputs", "amax_inputs", 0.0,
                                      4.0 * ((double)P.cfg.in_channels * F + 2.0 * (double)hop * F) * B));
        HIP_TRY(launch_amax_inputs(sigbuf, sig_stride, ppg, B, P.cfg.in_channels, F, (int)hop, lengths, amax_inb, amaxb,
                                   (int)ab->numel, stream));
        if (prof) HIP_TRY(prof->end());
    }

    // ---- speaker bias for all blocks + zeroed InstanceNorm accumulators: first needed by up block 0, so
    // off the critical path (the down chains) on the second helper stream, joined before the up blocks ----
    // (only where the helper streams pay at all: below ~1.5e5 samples per call the fork / join events cost more)
    hipStream_t s_pre = (ctx && (streams_mask & 2)) ? ctx->aux[1] : stream;
    if (spk) {
        HIP_TRY(order_after(stream, s_pre));               // after whatever the caller enqueued before us
        const BufferSpec* s0 = ws.find("up.0.stats");
        const BufferSpec* sl = ws.find("up." + std::to_string(n - 1) + ".stats");
        const size_t span = sl->off_bytes + align_up((size_t)sl->numel * sizeof(double), 256) - s0->off_bytes;
        HIP_TRY(hipMemsetAsync(wsb + s0->off_bytes, 0, span, s_pre));
        SpkBlock blocks[FASTSVC_MAX_STAGES];
        for (int i = 0; i < n; ++i) {
            blocks[i].w = blob + P.up[i].emb.w_off;
            blocks[i].bias = blob + P.up[i].emb.b_off;
            blocks[i].out = buf("up." + std::to_string(i) + ".spk");
            blocks[i].C = P.up[i].C;
            blocks[i].amax = P.storage == 0 ? am("up." + std::to_string(i) + ".spk") : nullptr;
        }
        double spk_c = 0; for (int i = 0; i < n; ++i) spk_c += P.up[i].C;
        if (prof) HIP_TRY(prof->begin(s_pre, "spk_proj", "spk_proj", 2.0 * spk_c * P.cfg.spk_emb_size * B,
                                      4.0 * (spk_c * P.cfg.spk_emb_size + (double)B * P.cfg.spk_emb_size + spk_c * B)));
        HIP_TRY(launch_spk_proj(spk_emb, blocks, n, B, P.cfg.spk_emb_size, s_pre));
        if (prof) HIP_TRY(prof->end());
    }

    // ---- conditioning chains: both signals per launch (gridDim.z = 2B) ----
    long Tk = T;
    const float* hprev = sigbuf;
    int Cprev = 1;
    long Tprev = T;
    bool hprev_compact = false;          // hprev already holds h_{k-1}[..., ::s_k] (run_cond_stage0)
    for (int k = 0; k < n; ++k) {
        const DownStage& d = P.down[k];
        Tk = Tk / d.scale;
        const std::string s = std::to_string(k);
        float* c1 = buf("down_c1." + s);
        float* c2 = buf("down_c2." + s);
        float* h = buf("down_h." + s);
        const long tsig = (long)B * d.C * Tk;     // signal stride of the (2B, C, Tk) buffers
        const long tb = (long)d.C * Tk;
        ConvParams base;
        std::memset(&base, 0, sizeof(base));
        base.B = B; base.T = (int)Tk; base.s = 1; base.mode = MODE_DIRECT;
        base.lens = lengths; base.frames_ld = F;
        base.amax_in_sig = B; base.amax_out_sig = B;               // (2B, C, T) tensors: one amax per (signal, utterance)
        // MEASURED: the stage's input (raw signal scan / the previous stage's output) and its output h_k; the tensors
        // in between are bounded from the input's maximum through the layers' (l1, bmax) - ConvParams::bnd_path
        float* am_prev = k == 0 ? amax_inb : am("down_h." + std::to_string(k - 1));
        float* am_h = am("down_h." + s);
        const float* bnd_c1 = blob + (k == 0 ? d.c1_raw[0].bnd_off : d.c1[0].bnd_off);
        const long bnd_c1_sig = k == 0 ? (long)(d.c1_raw[1].bnd_off - d.c1_raw[0].bnd_off) : d.c1[0].bnd_pair;
        // the stage up to h_k without the whole-stage fusion below: first conv (+ 1x1 residual), then c2 -> c3
        auto stage_unfused = [&](Profiler* pr) -> int {
            if (k == 0) {
                if (pr) HIP_TRY(pr->begin(stream, "down.0.c1", "in1_conv", 2.0 * 3 * d.C * (double)Tk * B * 2,
                                              4.0 * (1.0 + d.C) * (double)Tk * B * 2));
                HIP_TRY((P.storage == 1 ? bf16::launch_in1_conv : launch_in1_conv)(sigbuf, sig_stride, blob + d.c1_raw[0].w_off, blob + d.c1_raw[0].b_off,
                                        (long)(d.c1_raw[1].w_off - d.c1_raw[0].w_off),
                                        (long)(d.c1_raw[1].b_off - d.c1_raw[0].b_off), c1, 2, B, d.C, (int)Tk,
                                        lengths, (int)(Tk / F), stream, nullptr));     // (c1 is bounded, not measured)
                if (pr) HIP_TRY(pr->end());
            } else {
                float* r = buf("down_r." + s);
                ConvParams p = base;                                   // r = conv1x1(h_{k-1}[::s])
                p.x = hprev; p.x_sig = (long)B * Cprev * Tprev; p.x_b = (long)Cprev * Tprev; p.x_T = (int)Tprev;
                p.mode = MODE_DECIMATE; p.s = hprev_compact ? 1 : d.scale;
                p.amax_in = am_prev;
                static const bool no_dec2 = std::getenv("FASTSVC_NO_DEC2") != nullptr;      // A/B switch
                if (d.rc1[0].dec2 && !no_dec2) {
                    // both convs of the decimated input in one launch: c1 -> y, r -> y2
                    p.mode = MODE_DEC2;
                    p.y = c1; p.y_sig = tsig; p.y_b = tb;
                    p.y2 = r; p.y2_sig = tsig; p.y2_b = tb;
                    HIP_TRY(run_conv(d.rc1[0], blob, p, 2, (long)(d.rc1[1].w_off - d.rc1[0].w_off),
                                     (long)(d.rc1[1].b_off - d.rc1[0].b_off), stream, pr, ("down." + s + ".c1_res1x1").c_str()));
                } else {
                p.y = r; p.y_sig = tsig; p.y_b = tb;
                HIP_TRY(order_after(stream, s_side));                  // h_{k-1} is ready
                HIP_TRY(run_conv(d.r[0], blob, p, 2, (long)(d.r[1].w_off - d.r[0].w_off), (long)(d.r[1].b_off - d.r[0].b_off), s_side, pr, ("down." + s + ".res1x1").c_str()));
                p.flags = F_PRE_LRELU;                                 // c1 = conv3_d1(lrelu(h_{k-1}[::s]))
                p.y = c1;
                HIP_TRY(run_conv(d.c1[0], blob, p, 2, (long)(d.c1[1].w_off - d.c1[0].w_off), (long)(d.c1[1].b_off - d.c1[0].b_off), stream, pr, ("down." + s + ".c1").c_str()));
                }
            }
            {
                ConvParams p = base;                                   // c2 = conv3_d2(lrelu(c1))
                p.x = c1; p.x_sig = tsig; p.x_b = tb; p.x_T = (int)Tk;
                p.flags = F_PRE_LRELU;
                p.amax_in = am_prev; p.bnd_path[0] = bnd_c1; p.bnd_sig[0] = bnd_c1_sig;       // |c1| <= l1 |h_{k-1}| + bmax
                ConvParams p3 = p;                                     // h = conv3_d4(lrelu(c2)) + r
                p3.y = h; p3.y_sig = tsig; p3.y_b = tb; p3.amax_out = am_h;
                if (k == 0) {
                    p3.r1x = sigbuf; p3.r1x_sig = sig_stride; p3.r1x_b = T;
                    p3.r1w = blob + d.r_raw[0].w_off; p3.r1b = blob + d.r_raw[0].b_off;
                    p3.r1_sig = (long)(d.r_raw[1].w_off - d.r_raw[0].w_off);
                    if ((d.r_raw[1].b_off - d.r_raw[0].b_off) != (d.r_raw[1].w_off - d.r_raw[0].w_off))
                        return fail(FASTSVC_E_INVALID, "internal: rank-1 pair strides");
                } else {
                    p3.res = buf("down_r." + s); p3.res_sig = tsig; p3.res_b = tb;
                    HIP_TRY(order_after(s_side, stream));              // r is ready
                }
                const std::string n2 = "down." + s + ".c2_d2", n3 = "down." + s + ".c3_d4", n23 = "down." + s + ".c23";
                const long b3 = (long)(d.c3[1].b_off - d.c3[0].b_off);
                auto separate = [&](Profiler* pr) -> int {
                    p.y = c2; p.y_sig = tsig; p.y_b = tb;
                    HIP_TRY(run_conv(d.c2[0], blob, p, 2, (long)(d.c2[1].w_off - d.c2[0].w_off), (long)(d.c2[1].b_off - d.c2[0].b_off), stream, pr, n2.c_str()));
                    ConvParams q = p3;
                    q.x = c2; q.bnd_path[1] = blob + d.c2[0].bnd_off; q.bnd_sig[1] = d.c2[0].bnd_pair;   // ... and c2 behind it
                    HIP_TRY(run_conv(d.c3[0], blob, q, 2, (long)(d.c3[1].w_off - d.c3[0].w_off), b3, stream, pr, n3.c_str()));
                    return FASTSVC_OK;
                };
                // one launch for the pair where the stage has the fused variant (c2 then never reaches memory);
                // a tuning pass times it against the two launches and the table keeps the winner
                double sep_ms = 0.0;
                if (g_tune.tuning && !pr && d.c3[0].hxc_off[0]) {
                    int rc = separate(nullptr);                        // tunes the two launches' own shapes
                    if (rc != FASTSVC_OK) return rc;
                    hipEvent_t e0, e1;
                    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(FASTSVC_E_HIP, "hipEventCreate");
                    hipEventRecord(e0, stream);
                    for (int r = 0; r < 3 && rc == FASTSVC_OK; ++r) rc = separate(nullptr);
                    hipEventRecord(e1, stream);
                    float ms = 0.f;
                    if (rc == FASTSVC_OK && hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
                    hipEventDestroy(e0); hipEventDestroy(e1);
                    if (rc != FASTSVC_OK) return rc;
                    sep_ms = ms / 3.0;
                }
                bool fused = false;
                HIP_TRY(run_chain(d.c2[0], d.c3[0], blob, p3, 2, b3, stream, pr, n23.c_str(), sep_ms, fused));
                if (!fused) { const int rc = separate(pr); if (rc != FASTSVC_OK) return rc; }
            }
            return FASTSVC_OK;
        };
        // The WHOLE stage - chain and FiLM net of both signals - as one launch where it has the variant (compact
        // workspace: h_0 itself is then not materialised, stage 1 reads the compact decimated copy)
        if (k == 0 && P.compact) {
            bool whole = false;
            HIP_TRY(run_cond_stage0(P, blob, sigbuf, sig_stride, B, T, F, lengths, buf("ss.0"), buf("down_hd.1"),
                                    (P.storage == 0 && !no_scan) ? amax_inb : nullptr, am("down_h.0"), stream, prof, whole));
            if (whole) {
                hprev = buf("down_hd.1"); Cprev = d.C; Tprev = Tk / P.down[1].scale; hprev_compact = true;
                continue;
            }
            // (the layout left this stage's buffers out: there is nothing to fall back into)
            if (cond_stage_whole(P, 0, F)) return fail(FASTSVC_E_HIP, "whole-stage launch of stage 0 declined at run time");
        }
        if (k == 1 && P.compact && hprev_compact && n > 2) {
            bool whole = false;
            HIP_TRY(run_cond_stage1(P, blob, hprev, B, Tk, F, lengths, buf("ss.1"), buf("down_hd.2"),
                                    P.storage == 0 ? am("down_h.0") : nullptr, am("down_h.1"), stream, prof, whole));
            if (whole) {
                hprev = buf("down_hd.2"); Cprev = d.C; Tprev = Tk / P.down[2].scale; hprev_compact = true;
                continue;
            }
            if (cond_stage_whole(P, 1, F)) return fail(FASTSVC_E_HIP, "whole-stage launch of stage 1 declined at run time");
        }
        // Stage 0 as ONE launch where it has the variant (MODE_CHAIN1): the staging waves compute the 1 -> C conv
        // from the raw signal, so neither c1 nor c2 reaches memory; a tuning pass times it against the other path
        bool stage_fused = false;
        if (k == 0 && d.c3[0].hxc_off[0]) {
            ConvParams p3 = base;
            p3.x = sigbuf; p3.x_sig = sig_stride; p3.x_b = T; p3.x_T = (int)Tk;
            p3.flags = F_PRE_LRELU;
            p3.y = h; p3.y_sig = tsig; p3.y_b = tb;
            p3.amax_in = amax_inb; p3.amax_out = am_h;
            p3.r1x = sigbuf; p3.r1x_sig = sig_stride; p3.r1x_b = T;
            p3.r1w = blob + d.r_raw[0].w_off; p3.r1b = blob + d.r_raw[0].b_off;
            p3.r1_sig = (long)(d.r_raw[1].w_off - d.r_raw[0].w_off);
            if ((d.r_raw[1].b_off - d.r_raw[0].b_off) != (d.r_raw[1].w_off - d.r_raw[0].w_off))
                return fail(FASTSVC_E_INVALID, "internal: rank-1 pair strides");
            double sep_ms = 0.0;
            if (g_tune.tuning && !prof) {
                int rc = stage_unfused(nullptr);                   // tunes that path's own launches
                if (rc != FASTSVC_OK) return rc;
                hipEvent_t e0, e1;
                if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(FASTSVC_E_HIP, "hipEventCreate");
                hipEventRecord(e0, stream);
                for (int r = 0; r < 3 && rc == FASTSVC_OK; ++r) rc = stage_unfused(nullptr);
                hipEventRecord(e1, stream);
                float ms = 0.f;
                if (rc == FASTSVC_OK && hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
                hipEventDestroy(e0); hipEventDestroy(e1);
                if (rc != FASTSVC_OK) return rc;
                sep_ms = ms / 3.0;
            }
            HIP_TRY(run_chain(d.c2[0], d.c3[0], blob, p3, 2, (long)(d.c3[1].b_off - d.c3[0].b_off), stream, prof,
                              "down.0.c123", sep_ms, stage_fused, d.c1_raw));
        }
        if (!stage_fused) { const int rc = stage_unfused(prof); if (rc != FASTSVC_OK) return rc; }
        {
            // the FiLM net of the LAST stage feeds the first up block: critical path, caller's stream;
            // the others are only needed by later up blocks: helper stream
            hipStream_t sf = (k == n - 1) ? stream : s_film;
            HIP_TRY(order_after(stream, sf));                      // h_k is ready
            float* u = buf("film_u." + s);                         // (B, 2C, Tk): [lft ; sine] channels
            float* ssb = buf("ss." + s);
            const std::string nconv = "film." + s + ".conv", nheads = "film." + s + ".heads", nchain = "film." + s + ".chain";
            auto film_separate = [&](Profiler* pr) -> int {
                ConvParams p = base;
                p.x = h; p.x_sig = tsig; p.x_b = tb; p.x_T = (int)Tk;
                p.flags = F_POST_LRELU;
                p.y = u; p.y_sig = tb; p.y_b = 2 * tb;
                p.amax_in = am_h;
                HIP_TRY(run_conv(d.film[0], blob, p, 2, (long)(d.film[1].w_off - d.film[0].w_off), (long)(d.film[1].b_off - d.film[0].b_off), sf, pr, nconv.c_str()));
                ConvParams q = base;                               // [scale ; shift] summed over both signals
                q.x = u; q.x_sig = 0; q.x_b = 2 * tb; q.x_T = (int)Tk;
                q.y = ssb; q.y_sig = 0; q.y_b = 2 * tb;
                // u = [lrelu(conv_lft(h_lft)) ; lrelu(conv_sine(h_sine))]: the larger of the two halves' bounds
                q.amax_in = am_h; q.amax_in_sig = 0; q.amax_in2 = B;
                q.bnd_path[0] = blob + d.film[0].bnd_off; q.bnd_sig[0] = d.film[0].bnd_pair;
                HIP_TRY(run_conv(d.heads, blob, q, 1, 0, 0, sf, pr, nheads.c_str()));
                return FASTSVC_OK;
            };
            // the whole FiLM net of the stage in one launch where it has the variant: the 2C-channel intermediate
            // stays in LDS (block-diagonal first conv over both chains' outputs, then the heads)
            bool film_fused = false;
            if (d.filmc.hxc_off[0] && ((long)tsig + (long)d.C * Tk) * 4 < (1L << 31)) {
                double sep_ms = 0.0;
                if (g_tune.tuning && !prof) {
                    int rc = film_separate(nullptr);
                    if (rc != FASTSVC_OK) return rc;
                    hipEvent_t e0, e1;
                    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(FASTSVC_E_HIP, "hipEventCreate");
                    hipEventRecord(e0, sf);
                    for (int r = 0; r < 3 && rc == FASTSVC_OK; ++r) rc = film_separate(nullptr);
                    hipEventRecord(e1, sf);
                    float ms = 0.f;
                    if (rc == FASTSVC_OK && hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
                    hipEventDestroy(e0); hipEventDestroy(e1);
                    if (rc != FASTSVC_OK) return rc;
                    sep_ms = ms / 3.0;
                }
                ConvParams p3 = base;
                p3.x = h; p3.x_sig = tsig; p3.x_b = tb; p3.x_T = (int)Tk; p3.xsplit = d.C;
                p3.y = ssb; p3.y_sig = 0; p3.y_b = 2 * tb;
                p3.amax_in = am_h; p3.amax_in2 = B;                 // channels [lft ; sine]: the larger of the two chains' maxima
                HIP_TRY(run_chain(d.filmc, d.filmc, blob, p3, 1, 0, sf, prof, nchain.c_str(), sep_ms, film_fused));
            }
            if (!film_fused) { const int rc = film_separate(prof); if (rc != FASTSVC_OK) return rc; }
            if (sf != stream && ctx) {
                ss_ready[k] = ctx->ev[evi++ % ctx->ev.size()];
                HIP_TRY(hipEventRecord(ss_ready[k], sf));
            }
        }
        hprev = h; Cprev = d.C; Tprev = Tk; hprev_compact = false;
    }

    // ---- up blocks ----
    if (spk) HIP_TRY(order_after(s_pre, stream));           // speaker biases and zeroed sums are ready
    const float* x = ppg;
    if (P.storage == 1) {                                          // external float32 input -> workspace bf16
        float* pa = buf("ppg_act");
        HIP_TRY(bf16::launch_act_convert(ppg, pa, (long)B * P.cfg.in_channels * F, stream));
        x = pa;
    }
    int Cx = P.cfg.in_channels;
    long Tin = F;
    bool last_done = false;
    bool last_fusable = true;
    {
        static const bool no_fuse = std::getenv("FASTSVC_NO_FUSE_LAST") != nullptr;     // A/B switch
        char key[96];
        std::snprintf(key, sizeof(key), P.storage == 1 ? "conv_last|%d|%ld|b" : "conv_last|%d|%ld", B, (long)T);
        std::lock_guard<std::mutex> lock(P.tune_mu);
        auto it = P.tuned.find(key);
        if (no_fuse || (it != P.tuned.end() && it->second.algo != 3)) last_fusable = false;
    }
    for (int i = 0; i < n; ++i) {
        const UpStage& u = P.up[i];
        const int k = n - 1 - i;
        const long Tout = Tin * u.scale;
        const std::string s = std::to_string(i);
        float* a = buf("up." + s + ".a");
        float* xr = buf("up." + s + ".xr");
        float* u1 = buf("up." + s + ".u1");
        float* xm = buf("up." + s + ".xmid");
        float* u2 = buf("up." + s + ".u2");
        float* u3 = buf("up." + s + ".u3");
        float* xo = buf("up." + s + ".out");
        float* pb = buf("up." + s + ".spk");
        double* st = reinterpret_cast<double*>(buf("up." + s + ".stats"));
        const float* ss = buf("ss." + std::to_string(k));
        const long cb = (long)u.C * Tout;
        const long stn = (long)B * u.C * 2;
        ConvParams base;
        std::memset(&base, 0, sizeof(base));
        base.B = B; base.s = 1; base.mode = MODE_DIRECT;
        base.lens = lengths; base.frames_ld = F;
        // Every tensor that feeds a FiLM-affine is stored already affined (u = scale*t + shift,
        // written by the producing epilogue, which also accumulates its InstanceNorm sums), so the
        // consuming conv stages ONE tensor and applies (u - mean) * rstd + p, LeakyReLU on the fly.
        const int aff_out = F_AFF_OUT | (spk ? F_STATS : 0);
        const int pre = F_PRE_LRELU | (spk ? F_PRE_NORM : 0);

        if (ss_ready[k]) HIP_TRY(hipStreamWaitEvent(stream, ss_ready[k], 0));   // scale/shift of stage k

        // amax rows (float32 storage): every tensor a convolution stages WITHOUT an InstanceNorm in front
        float* am_x = i == 0 ? amax_inb + 2 * B * 256 : am("up." + std::to_string(i - 1) + ".out");
        float* am_p = am("up." + s + ".spk");                      // speaker biases: bound of a normalised row
        // the block's head - conv_first and the two stretched convs behind it - as ONE launch where it has the variant
        bool head_fused = false;
        {
            ConvParams ph = base;
            ph.x = x; ph.x_b = (long)Cx * Tin; ph.x_T = (int)Tin;
            ph.y = xr; ph.y_b = cb; ph.y2 = u1; ph.y2_b = cb;
            ph.flags = F_POST_LRELU | aff_out;
            ph.ss_out = ss; ph.ss_out_b = 2 * cb; ph.st_out = st;
            ph.amax_in = am_x;
            HIP_TRY(run_uphead(u, blob, ph, stream, prof, ("up." + s + ".head").c_str(), head_fused));
        }
        // xmid = conv_d3(lrelu(norm(u1))) + xr with xr computed INSIDE that launch from `a` (run_d3x) where the variant
        // exists and the launch table does not say otherwise: the residual conv's launch and the tensor xr then go
        ConvParams pd3 = base;
        pd3.x = u1; pd3.x_b = cb; pd3.x_T = (int)Tout; pd3.T = (int)Tout;
        pd3.flags = pre | aff_out; pd3.st_in = st; pd3.spk = pb;
        pd3.amax_in = spk ? am_p : nullptr;
        pd3.no_hx = (!spk && P.storage == 0) ? 1 : 0;
        pd3.y = xm; pd3.y_b = cb; pd3.y2 = u2; pd3.y2_b = cb;
        pd3.ss_out = ss; pd3.ss_out_b = 2 * cb; pd3.st_out = st + stn;
        pd3.x2 = a; pd3.x2_b = (long)u.C * Tin; pd3.x2_T = (int)Tin; pd3.s2 = u.scale;
        pd3.amax_x2 = am_x; pd3.bnd_x2 = blob + u.first.bnd_off;   // |a| <= l1 |x| + bmax
        const std::string nd3x = "up." + s + ".d3x";
        bool d3_fused = false;
        if (!head_fused) HIP_TRY(run_d3x(u, blob, pd3, stream, nullptr, nd3x.c_str(), 0, 0.0, d3_fused));
        ConvParams p = base;
        if (!head_fused) {
        p = base;                                                  // a = conv_first(x)
        p.x = x; p.x_b = (long)Cx * Tin; p.x_T = (int)Tin;
        p.y = a; p.y_b = (long)u.C * Tin; p.T = (int)Tin;
        p.amax_in = am_x;
        HIP_TRY(run_conv(u.first, blob, p, 1, 0, 0, stream, prof, ("up." + s + ".conv_first").c_str()));

        p = base;                                                  // xr = conv_res(stretch(a))
        p.x = a; p.x_b = (long)u.C * Tin; p.x_T = (int)Tin;
        p.mode = MODE_STRETCH; p.s = u.scale;
        p.y = xr; p.y_b = cb; p.T = (int)Tout;
        p.amax_in = am_x; p.bnd_path[0] = blob + u.first.bnd_off;  // |a| <= l1 |x| + bmax  (xr is only ever a residual: not tracked)
        if (!d3_fused) {
        HIP_TRY(order_after(stream, s_side));                      // a is ready
        HIP_TRY(run_conv(u.res, blob, p, 1, 0, 0, s_side, prof, ("up." + s + ".res_stretch").c_str()));
        }

        p.flags = F_PRE_LRELU | F_POST_LRELU | aff_out;            // u1 = aff(lrelu(conv_up(stretch(lrelu(a)))))
        p.y = nullptr; p.y2 = u1; p.y2_b = cb;
        p.ss_out = ss; p.ss_out_b = 2 * cb; p.st_out = st;
        HIP_TRY(run_conv(u.up, blob, p, 1, 0, 0, stream, prof, ("up." + s + ".up_stretch").c_str()));
        }

        // tiny batches: the sums the epilogue accumulated are replaced by exact ones (see stats_exact_kernel)
        auto exact_stats = [&](const float* ut, double* stp) -> hipError_t {
            // ... and, in a small ragged batch of float32 storage, for its utterances of at most 4 frames (the launch is
            // B x C one-wave workgroups that leave at once for the longer ones: not worth it on big batches, where a
            // decode harness's length buckets never put a 1-frame utterance anyway)
            const bool small_ragged = lengths && P.storage == 0 && F <= 64;
            if (!spk || !(g_exact_f32 || small_ragged)) return hipSuccess;
            return launch_stats_exact(ut, stp, B, u.C, (int)Tout, lengths, (int)(Tout / F), 4, stream);
        };
        HIP_TRY(exact_stats(u1, st));
        p = base;                                                  // xmid = conv_d3(lrelu(norm(u1))) + xr
        p.x = u1; p.x_b = cb; p.x_T = (int)Tout; p.T = (int)Tout;
        p.flags = pre | aff_out; p.st_in = st; p.spk = pb;
        p.res = xr; p.res_b = cb;
        // Behind the norm sqrt(T) + max |p| bounds the staged row.  WITHOUT a speaker embedding there is no norm
        // (fastsvc.py:134-140): the three FiLM affines of a block multiply unnormalised activations, nothing bounds
        // them short of measuring every FiLM-affined tensor in the store-bound FiLM epilogues - that path keeps
        // these three convs on the exact f32-input MFMA kernels instead (float32 storage; bf16 has float32's range)
        p.amax_in = spk ? am_p : nullptr;
        p.no_hx = (!spk && P.storage == 0) ? 1 : 0;
        p.y = xm; p.y_b = cb; p.y2 = u2; p.y2_b = cb;              // and u2 = aff(xmid)
        p.ss_out = ss; p.ss_out_b = 2 * cb; p.st_out = st + stn;
        if (d3_fused) {
            bool ran = false;
            HIP_TRY(run_d3x(u, blob, pd3, stream, prof, nd3x.c_str(), 1, 0.0, ran));
            // (the route was decided by the query call in front of conv_first - the separate residual launch was skipped on its
            // word; a launch table that changed in between - another thread's load_tuned on this plan - must not leave xmid / u2
            // unwritten: ADVICE r5)
            if (!ran) return fail(FASTSVC_E_INVALID, "up." + s + ".d3x: the launch table changed between the route decision and the launch");
        } else {
        HIP_TRY(order_after(s_side, stream));                      // xr is ready
        HIP_TRY(run_conv(u.d3, blob, p, 1, 0, 0, stream, prof, ("up." + s + ".d3").c_str()));
        if (g_tune.tuning && !prof && !head_fused) {
            // tuning pass: the two separate launches again, timed, then the fused shapes against them
            ConvParams pr = base;
            pr.x = a; pr.x_b = (long)u.C * Tin; pr.x_T = (int)Tin;
            pr.mode = MODE_STRETCH; pr.s = u.scale;
            pr.y = xr; pr.y_b = cb; pr.T = (int)Tout;
            pr.amax_in = am_x; pr.bnd_path[0] = blob + u.first.bnd_off;
            ConvParams pq = p;
            pq.flags &= ~F_STATS;                                  // (the InstanceNorm sums are accumulated once)
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(FASTSVC_E_HIP, "hipEventCreate");
            hipEventRecord(e0, stream);
            int rc = FASTSVC_OK;
            for (int r = 0; r < 3 && rc == FASTSVC_OK; ++r) {
                if (run_conv(u.res, blob, pr, 1, 0, 0, stream, nullptr, ("up." + s + ".res_stretch").c_str()) != hipSuccess ||
                    run_conv(u.d3, blob, pq, 1, 0, 0, stream, nullptr, ("up." + s + ".d3").c_str()) != hipSuccess)
                    rc = fail(FASTSVC_E_HIP, "timing the separate residual / d3 launches");
            }
            hipEventRecord(e1, stream);
            float ms = 0.f;
            if (rc == FASTSVC_OK && hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            if (rc != FASTSVC_OK) return rc;
            bool dummy = false;
            HIP_TRY(run_d3x(u, blob, pd3, stream, nullptr, nd3x.c_str(), 2, ms / 3.0, dummy));
        }
        }
        HIP_TRY(exact_stats(u2, st + stn));

        p.x = u2; p.st_in = st + stn; p.res = nullptr;             // u3 = aff(conv_d9(lrelu(norm(u2))))
        p.y = nullptr; p.y2 = u3;
        p.st_out = st + 2 * stn;
        
        HIP_TRY(run_conv(u.d9, blob, p, 1, 0, 0, stream, prof, ("up." + s + ".d9").c_str()));
        HIP_TRY(exact_stats(u3, st + 2 * stn));

        p.x = u3; p.st_in = st + 2 * stn;                          // out = conv_d27(lrelu(norm(u3))) + xmid
        p.flags = pre; p.ss_out = nullptr; p.st_out = nullptr; p.y2 = nullptr;
        p.res = xm; p.res_b = cb;
        p.y = xo; p.y_b = cb;
        p.amax_out = i + 1 < n ? am("up." + s + ".out") : nullptr;
        if (i == n - 1 && P.cfg.out_channels == 1 && last_fusable) {
            // conv_last in the same launch where the launch has the variant (run_conv decides; the block's
            // C-channel output is then not materialised - a launch table with algorithm 0 under "conv_last|B|T"
            // keeps the two launches, e.g. to read the `up.<n-1>.out` tap)
            p.last_w = blob + P.last.w_off; p.last_b = blob + P.last.b_off;
            p.last_y = out; p.last_y_b = (long)P.cfg.out_channels * T;
            if (lengths) HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * T, stream));   // ragged batch: zero padding of the waveform
        }
        g_last_fused = false;
        HIP_TRY(run_conv(u.d27, blob, p, 1, 0, 0, stream, prof, ("up." + s + ".d27").c_str()));
        if (i == n - 1) last_done = g_last_fused;

        x = xo; Cx = u.C; Tin = Tout;
    }

    // ---- conv_last ----
    if (last_done) return FASTSVC_OK;
    if (prof) HIP_TRY(prof->begin(stream, "conv_last", "pointwise_out", 2.0 * Cx * P.cfg.out_channels * (double)T * B,
                                  4.0 * (Cx + P.cfg.out_channels) * (double)T * B));
    HIP_TRY((P.storage == 1 ? bf16::launch_pointwise_out : launch_pointwise_out)(x, blob + P.last.w_off, blob + P.last.b_off, out, B, Cx,
                                 P.cfg.out_channels, (int)T, lengths, (int)hop, stream));
    if (prof) HIP_TRY(prof->end());
    return FASTSVC_OK;
}

void fastsvc_split_half(const float* x, int64_t n, uint16_t* f16_hi, uint16_t* f16_lo, uint16_t* bf16) {
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t hi = f32_to_f16(x[i]);
        if (f16_hi) f16_hi[i] = hi;
        if (f16_lo) f16_lo[i] = f32_to_f16(x[i] - f16_to_f32(hi));
        if (bf16) bf16[i] = f32_to_bf16(x[i]);
    }
}

int fastsvc_stream_prepare(void* stream) {
    ExecCtx* c = exec_ctx_for(static_cast<hipStream_t>(stream));
    if (!c) return fail(FASTSVC_E_HIP, "could not create the helper streams / events");
    std::lock_guard<std::mutex> busy(c->busy);
    ensure_measured(static_cast<hipStream_t>(stream), c);
    return FASTSVC_OK;
}

int fastsvc_stream_release(void* stream) { return exec_ctx_release(static_cast<hipStream_t>(stream)); }

int fastsvc_plan_set_storage(fastsvc_plan* plan, int32_t dtype) {
    if (!plan || (dtype != 0 && dtype != 1)) return fail(FASTSVC_E_INVALID, "storage dtype must be 0 (float32) or 1 (bfloat16)");
    plan->storage = dtype;
    return FASTSVC_OK;
}

int fastsvc_plan_get_storage(const fastsvc_plan* plan) { return plan ? plan->storage : 0; }

int fastsvc_tuned_count(const fastsvc_plan* plan) {
    if (!plan) return 0;
    std::lock_guard<std::mutex> lock(plan->tune_mu);
    return (int)plan->tuned.size();
}

int fastsvc_tuned_get(const fastsvc_plan* plan, int32_t index, char* key_out, int32_t shape_out[5]) {
    if (!plan || !key_out || !shape_out) return fail(FASTSVC_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(plan->tune_mu);
    if (index < 0 || index >= (int32_t)plan->tuned.size()) return fail(FASTSVC_E_INVALID, "index out of range");
    auto it = plan->tuned.begin();
    std::advance(it, index);
    std::snprintf(key_out, 96, "%s", it->first.c_str());
    shape_out[0] = it->second.NW; shape_out[1] = it->second.WM; shape_out[2] = it->second.WN; shape_out[3] = it->second.tpw;
    shape_out[4] = it->second.algo;
    return FASTSVC_OK;
}

int fastsvc_tuned_set(const fastsvc_plan* plan, const char* key, const int32_t shape[5]) {
    if (!plan || !key || !shape) return fail(FASTSVC_E_INVALID, "null argument");
    if (std::strlen(key) >= 96) return fail(FASTSVC_E_INVALID, "key too long");
    std::lock_guard<std::mutex> lock(plan->tune_mu);
    plan->tuned[key] = fastsvc_plan::Choice{shape[0], shape[1], shape[2], shape[3], shape[4]};
    plan->priors.clear();
    return FASTSVC_OK;
}

int fastsvc_forward(const fastsvc_plan* plan, const void* dev_blob,
                    const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                    float* out, int32_t B, int32_t F, const int32_t* lengths,
                    void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(plan, dev_blob, ppg, sine, lft, spk_emb, out, B, F, lengths, workspace,
                        workspace_bytes, stream, nullptr);
}

int fastsvc_autotune(const fastsvc_plan* plan, const void* dev_blob,
                     const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                     float* out, int32_t B, int32_t F,
                     void* workspace, size_t workspace_bytes, void* stream, int32_t* n_trials) {
    g_tune.tuning = true;
    g_tune.trials = 0;
    const int rc = forward_impl(plan, dev_blob, ppg, sine, lft, spk_emb, out, B, F, nullptr, workspace,
                                workspace_bytes, stream, nullptr);
    g_tune.tuning = false;
    if (n_trials) *n_trials = g_tune.trials;
    if (rc != FASTSVC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return FASTSVC_OK;
}

int fastsvc_forward_profile(const fastsvc_plan* plan, const void* dev_blob,
                            const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                            float* out, int32_t B, int32_t F, const int32_t* lengths,
                            void* workspace, size_t workspace_bytes, void* stream,
                            fastsvc_launch_record* records, int32_t max_records, int32_t* n_records) {
    if (!records || !n_records || max_records < 1) return fail(FASTSVC_E_INVALID, "null argument");
    Profiler prof;
    prof.stream = static_cast<hipStream_t>(stream);
    const int rc = forward_impl(plan, dev_blob, ppg, sine, lft, spk_emb, out, B, F, lengths, workspace,
                                workspace_bytes, stream, &prof);
    if (rc != FASTSVC_OK) return rc;
    HIP_TRY(prof.finish());
    const int n = (int)prof.recs.size() < max_records ? (int)prof.recs.size() : max_records;
    for (int i = 0; i < n; ++i) records[i] = prof.recs[i];
    *n_records = n;
    return FASTSVC_OK;
}

}  // extern "C"
