// fastsvc_kernels.h - launch-side interface between the plan/orchestration code (fastsvc_plan.cpp)
// and the gfx950 kernels (fastsvc_kernels.hip).  Internal; the public ABI is include/fastsvc_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fastsvc {

// input index modes of the generic convolution (how output-rate column t maps to the source row)
enum : int {
    MODE_DIRECT = 0,    // src = t
    MODE_DECIMATE = 1,  // src = t * s        Squeeze2d == x[..., ::s]        (upsample.py:53-74)
    MODE_STRETCH = 2,   // src = t / s        Stretch2d == repeat_interleave  (upsample.py:21-50)
    // Stretch2d + k=3 d=1 conv computed at the INPUT rate (SURVEY note P).  With j = t / s,
    // phase = t % s and the stretched signal xs[t] = x[j]:
    //     out[s*j + phase] = z[j] + (phase == 0 ? a[j] : 0) + (phase == s-1 ? c[j] : 0)
    //     z = (W0+W1+W2) x[j],  a = W0 (x[j-1] - x[j]),  c = W2 (x[j+1] - x[j])
    // i.e. three input-rate 1-tap products instead of three output-rate ones: s times fewer MACs.
    // Geometry: T (tiles, x_T) is the INPUT length, the output tensors have T * s columns; the
    // packed weights hold W0 | W0+W1+W2 | W2 in the three tap slots (fastsvc_plan.cpp packer).
    MODE_POLY = 3,
    // k=3 conv with dilation d in {1,2,4} as Winograd F(2,3) along time: per output pair
    // (t, t+d) and input channel, with d0..d3 = x[t-d], x[t], x[t+d], x[t+2d],
    //     m0 = (d0-d2) g0, m1 = (d1+d2) g1, m2 = (d2-d1) g2, m3 = (d1-d3) g3,
    //     y[t] = m0+m1+m2,  y[t+d] = m1-m2-m3,   g = w0 | (w0+w1+w2)/2 | (w0-w1+w2)/2 | w2
    // four products per two outputs instead of six.  A 16-row MFMA tile holds 16 pairs = 32
    // consecutive outputs (pair m = p*d + r  <->  t = 2d*p + r); the window is staged de-interleaved
    // into 2d phase planes (x[t] -> plane t % 2d, position t / 2d) so every component is a
    // unit-stride LDS read.  ps = plane stride in floats; packed weights hold the four g planes.
    MODE_WINO = 4,
    // First two convs of a down-sampling stage in ONE launch (fastsvc.py:164-173): both read the
    // decimated input h[..., ::s];  y = Conv3_d1(lrelu(hd)) + bias  and  y2 = Conv1x1(hd) + bias2.
    // The window is staged raw, the consumer applies LeakyReLU to the three tap operands and feeds a
    // second accumulator set with the raw centre column; 4 weight "taps" per k-group (w0 w1 w2 | w1x1).
    MODE_DEC2 = 5,
    // Two k=3 convs back to back in ONE launch (half-precision MFMA kernels only; the c2 -> c3 pair of a down
    // stage, fastsvc.py:174-177):  y = ConvB_dil2(lrelu(ConvA_dil(pre(x)) + bias_mid)) + bias [+ epilogue].
    // The intermediate tile (with ConvB's halo, zero outside the utterance = ConvB's "same" padding) never
    // leaves LDS: one launch boundary and one tensor write + read less.
    MODE_CHAIN = 6,
    // MODE_CHAIN behind the FIRST conv of down stage 0 (k=3, 1 input channel, fastsvc.py:172-173): x is the raw
    // 1-channel signal (always float32) and the staging waves compute lrelu(conv3(lrelu(x)) + b1) on the fly
    // instead of loading a C-channel tensor - the whole stage is one launch that only writes its output.
    MODE_CHAIN1 = 7,
    // The head of an up block in ONE launch (half-precision MFMA kernels, float32 storage; fastsvc.py:92-97):
    //     a  = conv_first(x)                       k=3, d=1, C_in -> C at the block's INPUT rate - never leaves LDS
    //     y  = Conv3(Stretch_s(a))        + bias   the stretched residual conv (polyphase, MODE_POLY)      -> xr
    //     y2 = scale * lrelu(Conv3(Stretch_s(lrelu(a))) + bias2) + shift   (+ InstanceNorm partial sums)    -> u1
    // Two copies of the intermediate tile (raw, LeakyReLU'd) serve the two polyphase convs; three launches and the
    // write + two reads of `a` less.  whx: [conv_first units | residual conv's polyphase units | up conv's];
    // whx_inv: [first | residual | up | l1_first, bmax_first, 0, 0]; bias = residual conv's, bias2 = up conv's,
    // bias_mid = conv_first's; T = x_T = input columns, ldy = output pitch, s = stretch factor.
    MODE_UPHEAD = 8
};

enum : int {
    F_PRE_LRELU = 1,    // LeakyReLU(0.2) on the (transformed) input before the conv
    F_PRE_AFFINE = 2,   // u = scale * x + shift             (fastsvc.py:131-132)
    F_PRE_NORM = 4,     // u = (u - mean) * rstd + p         (fastsvc.py:134-139)
    F_POST_LRELU = 8,   // LeakyReLU(0.2) on conv + bias
    F_STATS = 16,       // accumulate sum / sum-of-squares of (scale_out * y + shift_out) per (b, co)
    F_AFF_OUT = 32      // also write u = scale_out * y + shift_out to y2 (the next conv's input)
};

constexpr float LRELU_SLOPE = 0.2f;
constexpr double IN_EPS = 1e-5;

// All strides are in float elements.  "sig" is the conditioning-signal index (0 = lft, 1 = sine)
// for the dual-signal launches of the down-sampling / FiLM chains; nsig == 1 otherwise.
struct ConvParams {
    // source tensor (nsig, B, CIN, x_T)
    const float* x;
    long x_sig, x_b;
    int x_T;
    int CIN, KC, nchunks;        // KC = channels staged per chunk (multiple of 4); nchunks*KC >= CIN
    // packed weights / bias
    const float* w;
    long w_sig;
    const float* bias;
    long bias_sig;
    int Q;                       // k-steps per 16*MW-row group = ntaps * nchunks * KC / 4
    // half-precision-MFMA kernels (fastsvc_hx.hip): pre-split weight fragments
    // [group of 16*MW channels][32-channel chunk][tap][16-channel tile][hi, lo][lane][8 halves]
    const void* whx;
    long whx_sig;                // lft -> sine stride in BYTES (paired convs)
    int nch32;                   // 32-channel K chunks = ceil(CIN / 32)
    int stagger;                 // start delay (x 1024 cycles) of the second workgroup of a CU (two-per-CU variants)
    // MODE_CHAIN: the second conv (CMID -> COUT, dilation dil2); the first one maps CIN -> CMID with `dil`
    const float* bias_mid;       // bias of the first conv (added before the LeakyReLU in between)
    long bias_mid_sig;
    // conv_last fused behind the last block's conv (half-precision MFMA kernels, residual epilogue, one channel
    // group, one output channel; fastsvc.py:301,330): the wave that holds all C channels of its time steps also
    // reduces them, wave = last_w . (conv + res) + last_b, and the C-channel tensor is not written (y = null)
    const float* last_w;         // (1, C)
    const float* last_b;         // (1)
    float* last_y;               // (B, 1, ldy) float32
    long last_y_b;
    // MODE_CHAIN: input channels split over the two signals' tensors (the FiLM net of a stage reads both chains'
    // outputs as ONE 2C-channel input): channel cc lives in signal cc / xsplit (stride x_sig) at row cc % xsplit;
    // 0 = one tensor
    int xsplit;
    const float* in1_w;          // MODE_CHAIN1: first conv's raw weights (CIN, 1, 3) and bias (CIN)
    const float* in1_b;
    long in1_w_sig, in1_b_sig;
    int CMID, nch32b, dil2;      // nch32b = ceil(CMID / 32); whx holds [nch32 units of A | nch32b units of B] per group
    int ngroups;                 // number of 16*MW-row groups
    // ---- dynamic range of the split-binary16 products (float32 storage, fastsvc_hx.hip) ----
    // binary16 has an absolute floor (2^-24) and ceiling (65504) float32 does not have, so both operands are moved
    // into its range by EXACT power-of-two factors before the split and the factors leave again in the epilogue:
    //   weights: per output channel, chosen by the packer (max |w| of the channel -> [2^14, 2^15)); whx_inv holds the
    //            inverse factors: [16*MW*ngroups] floats per table; MODE_DEC2: [k=3 conv | 1x1 conv];
    //            MODE_CHAIN: [first conv | second conv | l1_first, bmax_first, l1_in1, bmax_in1] (l1 = largest
    //            absolute row sum of the first conv's weights, bmax = its largest |bias|: they bound the intermediate
    //            tensor that never leaves LDS; in1: the 1 -> C conv the staging waves compute in MODE_CHAIN1)
    //   activations: per (tensor, utterance), from the largest magnitude the PRODUCING kernel saw (amax slots in the
    //            workspace: every epilogue keeps a running max of what it writes and every wave ends with one
    //            atomic max; the raw inputs are scanned by amax_inputs_kernel).  Entry sig * amax_in_sig + b of
    //            amax_in (and, with amax_in2 > 0, entry amax_in2 + b for the second signal's tensor: the larger
    //            bound counts), 8 floats each - the value is their max; behind an InstanceNorm (F_PRE_NORM) amax_in is the row of the speaker biases p and the
    //            bound of a normalised row, sqrt(x_T) + max |p|, is used.
    const float* whx_inv;
    long whx_inv_sig;            // lft -> sine stride in floats
    const float* amax_in;
    int amax_in_sig, amax_in2;
    // the staged tensor is not the measured one but up to two convolutions downstream of it: bound = bound * l1 + bmax
    // per layer on the path ((l1, bmax) pairs written by the packer; second signal bnd_sig floats further).  Keeps the
    // number of MEASURED tensors - each costs its producer device-scope atomics - to one per stage / block.
    const float* bnd_path[2];
    long bnd_sig[2];
    // ---- second operand of a MODE_DIRECT launch (half-precision MFMA kernels; the middle conv of an up block with the
    // block's stretched residual conv folded in, fastsvc.py:94-100):  y = Conv3_dil(pre(x)) + Conv3_1(Stretch_s2(x2)) +
    // bias + bias2.  x2 is the RAW (B, CIN, x2_T) tensor at 1 / s2 of the output rate (x2_T * s2 == T; a ragged batch:
    // lens[b] * x2len_mul columns); the staging waves write every input column s2 times into a second LDS window and the
    // consumers run three more taps (dilation 1) on it - the residual tensor is never materialised.  whx then holds, per
    // channel group, the units [conv chunk 0 | x2 conv chunk 0 | conv chunk 1 | x2 conv chunk 1 ...] and whx_inv the two
    // tables [conv | x2 conv]; amax_x2 / bnd_x2: the measured row and the (l1, bmax) pair that bound x2 (float32 storage)
    const float* x2;
    long x2_b;
    int x2_T, ldx2, s2, x2len_mul;
    const float* amax_x2;
    const float* bnd_x2;
    float* amax_out;             // entry sig * amax_out_sig + b: largest |value| of y (residual epilogues and the fused
    int amax_out_sig;            //   pair's final one measure; null: not tracked)
    int no_hx;                   // host side only: keep this launch on the exact f32-input MFMA kernels (run_conv)
    // destination (nsig, B, COUT, T); y may be null when only the FiLM-affine output y2 is needed
    float* y;
    long y_sig, y_b;
    int T, COUT;
    float* y2;                   // (B, COUT, T): scale_out * y + shift_out  (F_AFF_OUT); MODE_DEC2: the 1x1 output
    long y2_b, y2_sig;
    const float* bias2;          // MODE_DEC2: bias of the 1x1 conv
    long bias2_sig;
    // optional residual, same geometry as y
    const float* res;
    long res_sig, res_b;
    // optional rank-1 residual r[co][t] = r1w[co] * r1x[t] + r1b[co] (1x1 conv with C_in == 1)
    const float* r1x;
    long r1x_sig, r1x_b;
    const float* r1w;
    const float* r1b;
    long r1_sig;
    // FiLM / InstanceNorm prologue on the input channels
    const float* ss_in;          // (B, 2*CIN, x_T): rows [0,CIN) scale, [CIN,2CIN) shift
    long ss_in_b;
    const double* st_in;         // (B, CIN, 2): sum, sum of squares over x_T
    const float* spk;            // (B, CIN) speaker bias p
    // statistics epilogue on the output channels
    const float* ss_out;         // (B, 2*COUT, T)
    long ss_out_b;
    double* st_out;              // (B, COUT, 2)
    int ntaps, dil, mode, s, flags;
    int B;                       // batch per signal; gridDim.z = nsig * B
    int xs;                      // LDS row stride in floats (== 16 mod 32, >= NT + 2*halo)
    int ps;                      // MODE_WINO: phase-plane stride in floats inside an LDS row
    // Row pitches (floats) of the input / output tensors.  T and x_T are the VALID row lengths; they
    // equal the pitches unless the batch is ragged: with `lens` (device, B frame counts) the kernels
    // use T = lens[b] * len_mul and x_T = lens[b] * xlen_mul for utterance b (masks, tiles,
    // InstanceNorm length) while rows keep the pitch of the longest utterance.
    int ldx, ldy;
    const int* lens;
    int len_mul, xlen_mul;
    int frames_ld;               // host side only: padded frame count F (run_conv derives len_mul = T / F)
    // FASTSVC_TIMELINE builds only (tools/timeline.py): per-wave cycle stamps, [workgroup][wave][64]
    unsigned long long* tl;
    int tl_wgs;                  // workgroups (linear id < tl_wgs) that record
    int vec;                     // 1: T % 4 == 0 and all row bases 16-byte aligned -> float4 epilogue
    int tpw;                     // pipelined kernel: consecutive time tiles walked by one workgroup
    int dbg;                     // ablation switches for profiling (FASTSVC_DBG env var); 0 in production
};

// A whole conditioning stage 0 as ONE launch (fastsvc_cond.hip; fastsvc.py:164-193 + :220-232 for both signals):
// raw signals -> [scale ; shift] of the stage (ss) and the COMPACT decimated chain output h[..., ::hd_s] (hd) that
// the next stage reads; no other tensor of the stage reaches memory.  Index s = 0 loudness, 1 sine excitation.
struct CondStage0Params {
    const float* x;              // raw signals, float32: signal s row b at x + s * x_sig + b * x_b
    long x_sig, x_b;
    int B, C, T, ld;             // T: padded columns of the batch, ld: row pitch of ss (elements)
    const int* lens;             // ragged batch: valid columns of utterance b = lens[b] * len_mul
    int len_mul;
    const float* in1_w[2];       // c1: (C, 1, 3) plain taps, bias (C)
    const float* in1_b[2];
    const float* r1w[2];         // 1x1 residual conv (C_in = 1): weight (C), bias (C)
    const float* r1b[2];
    const void* w[3][2];         // c2 / c3 / film.conv: half-precision fragments [tap][16-channel tile][piece][lane][8]
    const float* bias[3][2];
    const float* winv[3][2];     // float32 storage: inverse per-channel weight scales of those fragments
    const float* bnd[4][2];      // float32 storage: (l1, bmax) of c1, c2, c3, film.conv (bounds of the LDS-resident tensors)
    const float* bnd_r[2];       //   ... and of the 1x1 residual conv
    const float* cbnd[2];        // float32 storage: per-channel bounds [tensor c1 c2 h u][alpha | beta][C]: |t[c]| <= alpha amax + beta (packer)
    const void* w5;              // heads: [32-channel chunk][tap][16-channel tile][piece][lane][8]
    const float* b5;             // (2C): lft + sine biases summed
    const float* winv5;
    const float* amax_in;        // float32 storage: amax entries of the raw signals [s * B + b]
    float* amax_hd;              // float32 storage: amax entries of hd [s * B + b] (largest |value| written)
    void* ss;                    // (B, 2C, ld) activation storage type
    long ss_b;
    void* hd;                    // (2B, C, hd_ld): signal s utterance b at hd + (s * hd_sig + b * hd_b) elements; null: not written
    long hd_sig, hd_b;
    int hd_ld, hd_s;
    int tpw;                     // consecutive time tiles per workgroup
    int small;                   // 1: the short-tile instance (more workgroups for small batches)
};

// A whole conditioning stage k >= 1 with C = 48 channels (C_in = 24) as ONE launch (fastsvc_cond.hip): the COMPACT decimated
// output of the stage before (hd_k = h_{k-1}[..., ::s_k], written by the launch of stage k-1) -> [scale ; shift] of the
// stage (ss) and the compact h_k[..., ::hd_s] for the stage after.  Index s = 0 loudness, 1 sine excitation.
struct CondStage1Params {
    const void* x;               // (2B, Cin, ldx) activation storage: signal s utterance b at (s * x_sig + b * x_b) elements
    long x_sig, x_b;
    int ldx;
    int B, C, Cin, T, ld;        // T: padded columns at this stage's rate, ld: row pitch of ss (elements)
    const int* lens;             // ragged batch: valid columns of utterance b = lens[b] * len_mul
    int len_mul;
    const void* w1[2];           // first k=3 conv + 1x1 residual conv (MODE_DEC2 fragments): [slot w0 w1 w2 | w1x1][tile][piece]
    const float* b1[2];          // bias of the k=3 conv
    const float* br[2];          // bias of the 1x1 conv
    const float* winv1[2];       // float32 storage: inverse weight scales [k=3 conv | 1x1 conv], 16 * ntiles each
    const void* w[3][2];         // c2 / c3 / film.conv: [32-channel chunk][tap][tile][piece]
    const float* bias[3][2];
    const float* winv[3][2];
    const float* bnd[4][2];      // float32 storage: (l1, bmax) of c1, c2, c3, film.conv
    const float* bnd_r[2];
    const float* cbnd[2];        // float32 storage: per-channel bounds [c1 c2 h u][alpha | beta][C] (packer)
    const void* w5;              // heads 2C -> 2C: [group of 3 tiles][chunk][tap][tile][piece]
    const float* b5;
    const float* winv5;
    const float* amax_in;        // float32 storage: amax entries of x [s * B + b]
    float* amax_hd;              // float32 storage: amax entries of hd
    void* ss;                    // (B, 2C, ld)
    long ss_b;
    void* hd;                    // (2B, C, hd_ld) compact h[..., ::hd_s]; null: not written
    long hd_sig, hd_b;
    int hd_ld, hd_s;
    int tpw;
    int small;
};

enum : int { DBG_NO_LOAD = 1, DBG_NO_MFMA = 2, DBG_NO_EPILOGUE = 4, DBG_NO_COMMIT = 8, DBG_NO_WEIGHTS = 16 };
// The switches exist only in the diagnostic build (-DFASTSVC_DEBUG_SWITCHES: `build --timeline`).  A run-time branch
// around a pipeline stage is not free even when never taken: hipcc's wait-count bookkeeping merges the path that
// skips the stage, e.g. with `if (dbg & NO_COMMIT) return;` in front of the staging waves' commit it had to assume
// the window loads might still be in flight at the loop head and put `s_waitcnt vmcnt(0)` in front of every second
// unit's loads - the two-deep prefetch was one deep.
#ifdef FASTSVC_DEBUG_SWITCHES
#define FASTSVC_DBG_ON(p, flag) (((p).dbg & (flag)) != 0)
#else
#define FASTSVC_DBG_ON(p, flag) false
#endif

struct ConvLaunch {
    int MW;      // 16-channel tiles per wave (1, 2 or 3), fixed by the packed weight layout
    int NW;      // 16-column time tiles per wave (1, 2 or 4)
    int WM, WN;  // waves per workgroup along output channels / time (WM * WN == 4)
    int nsig;
    int pipe;    // 1: pipelined float4 kernel, 0: generic scalar-staging kernel (WM=1, WN=4), 2: conv_hx, 3: conv_wx
};

// the pipelined kernel needs 24-channel K chunks and no fused input affine; any row length
bool conv_pipe_supported(const ConvParams& p);
// tile shapes the polyphase (MODE_POLY) variant of the pipelined kernel is compiled for
bool conv_poly_shape(int MW, int NW, int WM, int WN);
// workgroups per CU allowed by the register budget of the pipelined variant (epilogue kind:
// 0 generic, 1 plain, 2 residual, 3 rank-1 residual, 4 FiLM-affine)
int conv_ws_resident(int MW, int NW, int mode, int epi_kind);
// whether that variant handles rows that are not a multiple of 4 long (S: polyphase stretch, else 1)
bool conv_ws_tail_ok(int MW, int NW, int mode, int epi_kind, int S);

// generic k in {1,3} dilated conv, MFMA f32 16x16x4
hipError_t launch_conv(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);

// the same convolutions on the 16x16x32 half-precision MFMA (fastsvc_hx.hip): float32 storage = split-half
// (hi + lo binary16 pieces, three products, fp32-class), bfloat16 storage = one bf16 product.  Needs
// p.whx / p.nch32, rows that are a multiple of 4 long; cfg.pipe is ignored.
hipError_t launch_conv_hx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);
// whether launch_conv_hx is compiled for this mode / shape
bool conv_hx_shape(int mode, int MW, int NW, int WM, int WN);
// whether the half-precision-MFMA instances of this mode / epilogue kind handle rows whose own length (ragged batch at
// the frame rate or twice it) is not a multiple of 4 - the pitches must be
bool conv_hx_tail_ok(int mode, int MW, int epi_kind, int S);
// whether the MODE_DIRECT instances with a second, stretched operand (ConvParams::x2) exist for this channel-tile count,
// number of K chunks and stretch factor
bool conv_hx_x2_ok(int MW, int nch32, int s2);

// the wide-layer kernel in which every wave multiplies (fastsvc_wx.hip): MODE_DIRECT, 48-channel groups (MW = 3), the
// weights once per workgroup and unit through LDS; bfloat16 storage.  cfg.pipe is ignored; same ConvParams as launch_conv_hx.
hipError_t launch_conv_wx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);
bool conv_wx_shape(int mode, int MW, int NW, int WM, int WN);
bool conv_wx_fits(int nch32, int dil);      // window buffers + two units of weights + patches fit the CU's LDS

// down-sampling stage 0, first conv (C_in = 1, k = 3, d = 1, LeakyReLU on the input):
//   y[sig][b][co][t] = bias[co] + sum_tap w[co][tap] * lrelu(x[sig][b][t + tap - 1])
//   x: signal 0 (B, T); signal 1 starts x_sig floats further (the caller's two tensors, no copy)
//   lens / len_mul: ragged batches (valid length of utterance b = lens[b] * len_mul, rows keep pitch T)
hipError_t launch_in1_conv(const float* x, long x_sig, const float* w, const float* bias, long w_sig, long b_sig,
                           float* y, int nsig, int B, int C, int T, const int* lens, int len_mul, hipStream_t stream,
                           float* amax_out = nullptr);

// Largest magnitude of every input row (float32 storage: scale of the split-binary16 staging, see ConvParams):
//   amax_in[sig * B + b] = max |signal[sig][b][0 .. T_b)|,  amax_in[2B + b] = max |ppg[b][:, 0 .. F_b)|
// Every (tensor, utterance) entry is AMAX_W = 8 floats wide (writers spread over the slots, readers take their max):
// the scan stores 8 partial maxima per input row (no atomics, nothing to zero first) and zeroes the `nzero` floats
// at `zero` - the intermediate tensors' entries, which later kernels of the forward accumulate into by atomic max.
hipError_t launch_noop(hipStream_t stream);          // empty kernel (stream calibration)
// exact (float64) InstanceNorm sums of a float32 (B, C, ld) tensor over each row's own length -> st (B, C, 2), overwriting
// (ragged batch: rows of utterances longer than max_frames are left as the conv accumulated them)
hipError_t launch_stats_exact(const float* u, double* st, int B, int C, int ld, const int* lens, int len_mul, int max_frames,
                              hipStream_t stream);
hipError_t launch_amax_inputs(const float* sig, long sig_stride, const float* ppg, int B, int C, int F, int hop,
                              const int* lens, float* amax_in, float* zero, int nzero, hipStream_t stream);

// conv_last: 1x1, y[b][o][t] = bias[o] + sum_c w[o][c] * x[b][c][t]
hipError_t launch_pointwise_out(const float* x, const float* w, const float* bias, float* y,
                                int B, int C, int O, int T, const int* lens, int len_mul, hipStream_t stream);

hipError_t launch_cond_stage0(const CondStage0Params& p, hipStream_t stream);
int cond_stage0_tile_columns(int small);
hipError_t launch_cond_stage1(const CondStage1Params& p, hipStream_t stream);
int cond_stage1_tile_columns(int small);

// speaker bias for all up blocks: p[blk][b][c] = bias + W[c] . (e / max(||e||, 1e-12))
struct SpkBlock {
    const float* w;     // (C, E)
    const float* bias;  // (C)
    float* out;         // (B, C)
    int C;
    float* amax;        // float32 storage: amax row of the speaker biases, (B, 8) - see ConvParams; null otherwise
};
hipError_t launch_spk_proj(const float* emb, const SpkBlock* blocks, int nblocks, int B, int E,
                           hipStream_t stream);

// The same launchers compiled with bfloat16 activation storage (fastsvc_kernels.hip built with
// -DFASTSVC_ACT_BF16): every pointer to an activation tensor then addresses bfloat16 elements (strides
// stay in elements); weights, biases, the raw signals (r1x), statistics and speaker biases stay float32.
namespace bf16 {
hipError_t launch_conv(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);
hipError_t launch_conv_hx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);
hipError_t launch_conv_wx(const ConvParams& p, const ConvLaunch& cfg, hipStream_t stream);
hipError_t launch_in1_conv(const float* x, long x_sig, const float* w, const float* bias, long w_sig, long b_sig,
                           float* y, int nsig, int B, int C, int T, const int* lens, int len_mul, hipStream_t stream,
                           float* amax_out = nullptr);
hipError_t launch_pointwise_out(const float* x, const float* w, const float* bias, float* y,
                                int B, int C, int O, int T, const int* lens, int len_mul, hipStream_t stream);
hipError_t launch_act_convert(const float* src, float* dst_bf16, long n, hipStream_t stream);
hipError_t launch_cond_stage0(const CondStage0Params& p, hipStream_t stream);
hipError_t launch_cond_stage1(const CondStage1Params& p, hipStream_t stream);
}  // namespace bf16

}  // namespace fastsvc
