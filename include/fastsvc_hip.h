/*
 * fastsvc_hip.h - C ABI of the MI355X-native FastSVC generator forward pass (gfx950 / CDNA4).
 *
 * Drop-in boundary for ONE path of lesterphillip/SVCC23_FastSVC (package `harana`):
 *
 *     harana/models/fastsvc.py:305-332   FastSVCGenerator.forward(x, s, l, spk_emb=None)
 *
 * plus the weight preparation that `decode_fastsvc.py:140-143` performs before it
 * (`load_model` -> `remove_weight_norm()` -> `.to(device)`).  The reference is pure Python on
 * PyTorch, so there is no FFI in it today; these entry points are what a ctypes / torch C++
 * binding inside `FastSVCGenerator.forward` binds (see INTEGRATION.md, and the binding shipped in
 * svcc23_fastsvc_amd/engine.py).
 *
 * Conventions
 *   - plain C types only; no torch types.  All tensors are contiguous float32, channel-major
 *     (B, C, T) exactly as the reference passes them (fastsvc.py:305-316).
 *   - device memory is owned by the caller (PyTorch's caching allocator in the shipped host
 *     code): the packed weight blob, the workspace, the inputs and the output.  The plan object
 *     is host-only and immutable after creation, so one plan may serve several devices/streams.
 *   - every launch is ordered with respect to the hipStream_t passed in; no hidden synchronisation - with ONE
 *     exception: large calls (from ~1.5e5 output samples) may fork kernels off the critical path onto a helper
 *     stream, forked from and joined back into that stream with events, and the helper streams / events of a
 *     (device, stream) pair are created, and the cost of a fork + join through them MEASURED (which synchronises
 *     the stream once), by the first such forward on that pair - or ahead of time by fastsvc_stream_prepare(),
 *     e.g. before graph capture (during capture nothing is created or measured: one stream).  Where the fork +
 *     join is slow the forward stays on the one stream.  fastsvc_stream_release() frees the pair's
 *     context (streams handed to forward must outlive it).  Concurrent forwards on DIFFERENT streams are
 *     independent; forwards issued from several host threads on the SAME stream are serialised while they enqueue.
 *   - return value 0 = success; negative = FASTSVC_E_*; fastsvc_last_error() gives the text.
 */
#ifndef FASTSVC_HIP_H
#define FASTSVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTSVC_ABI_VERSION 1
#define FASTSVC_MAX_STAGES 8

enum {
    FASTSVC_OK = 0,
    FASTSVC_E_INVALID = -1,      /* bad argument / shape invariant violated (reference: ValueError / RuntimeError) */
    FASTSVC_E_MISSING = -2,      /* a state-dict tensor is missing or has the wrong element count */
    FASTSVC_E_WORKSPACE = -3,    /* workspace too small */
    FASTSVC_E_HIP = -4,          /* a HIP runtime call or kernel launch failed */
    FASTSVC_E_UNSUPPORTED = -5   /* valid in the reference, not implemented here */
};

/* Constructor kwargs of FastSVCGenerator (fastsvc.py:238-246; egs/svcc23/fastsvc1/conf/fastsvc.yaml:23-29). */
typedef struct fastsvc_config {
    int32_t in_channels;                              /* 144 */
    int32_t n_stages;                                 /* len(mid_channels) == len(upsampling_scales) = 4 */
    int32_t mid_channels[FASTSVC_MAX_STAGES];         /* 192, 96, 48, 24 */
    int32_t upsampling_scales[FASTSVC_MAX_STAGES];    /* 2, 4, 4, 5 */
    int32_t out_channels;                             /* 1 */
    int32_t spk_emb_size;                             /* 512 */
    int32_t use_spk_emb;                              /* 1 */
} fastsvc_config;

/* One entry of the generator's state_dict, host memory, float32.  `name` is the reference key
 * (SURVEY.md 8(b)), e.g. "upsampling_nets.0.conv_first.weight_g". */
typedef struct fastsvc_tensor {
    const char* name;
    const float* data;
    int64_t numel;
} fastsvc_tensor;

typedef struct fastsvc_plan fastsvc_plan;   /* opaque, host-only */

int fastsvc_abi_version(void);
const char* fastsvc_last_error(void);       /* thread-local text of the last failure */

/* Replaces FastSVCGenerator.__init__ (fastsvc.py:238-303): builds the layer table, the packed
 * weight-blob layout and the workspace layout.  No GPU needed. */
int fastsvc_plan_create(const fastsvc_config* cfg, fastsvc_plan** out_plan);
void fastsvc_plan_destroy(fastsvc_plan* plan);

/* Activation storage in the workspace: 0 = float32 (default; the parity path), 1 = bfloat16 - every
 * intermediate tensor (conv outputs, FiLM-affined tensors, scale / shift) is stored as bfloat16, which
 * halves the traffic of the HBM-bound layers and the workspace; arithmetic (fp32 MFMA), weights,
 * InstanceNorm sums, inputs and output stay float32.  Accuracy is that of bf16 activations (about 1e-2
 * of the output range), so this is the mode for BASELINE config 3, not for the 1e-3 parity bar.
 * Set it before fastsvc_workspace_bytes / fastsvc_forward; needs F % 4 == 0 and the yaml channel counts. */
int fastsvc_plan_set_storage(fastsvc_plan* plan, int32_t dtype);
int fastsvc_plan_get_storage(const fastsvc_plan* plan);

/* Size in bytes of the packed weight blob (device resident; also what rank 0 broadcasts over RCCL). */
size_t fastsvc_weight_blob_bytes(const fastsvc_plan* plan);

/* Replaces load_state_dict + remove_weight_norm (harana/utils/utils.py:243-280, fastsvc.py:342-352):
 * takes the state dict in EITHER layout - `<layer>.weight_g` + `<layer>.weight_v` (checkpoint
 * layout, folded here as w = g * v / ||v|| per output channel) or `<layer>.weight` (after
 * remove_weight_norm) - plus `<layer>.bias`, and writes the kernel-layout blob (MFMA fragment
 * order, FiLM heads concatenated, zero padded) into `host_blob` (fastsvc_weight_blob_bytes bytes,
 * host memory).  Pure host code; the caller uploads the blob.  strict: every tensor must be present. */
int fastsvc_pack_weights(const fastsvc_plan* plan, const fastsvc_tensor* tensors, int32_t n_tensors,
                         void* host_blob);

/* Workspace layout of the plan's later fastsvc_workspace_bytes / fastsvc_forward calls: 0 (default) = every
 * intermediate has its own buffer (all fastsvc_workspace_tap tensors stay readable after a forward); 1 = compact,
 * intermediates of different stages whose lifetimes cannot overlap share buffers (about 40 % less at 64 x 10 s;
 * taps of shared buffers then hold the last stage's tensor).  Set it before sizing the workspace. */
int fastsvc_plan_set_workspace_mode(fastsvc_plan* plan, int32_t compact);

/* Device scratch needed by one forward of B utterances of F frames each (T = F * prod(scales)). */
size_t fastsvc_workspace_bytes(const fastsvc_plan* plan, int32_t B, int32_t F);

/* Replaces FastSVCGenerator.forward (fastsvc.py:305-332).
 *   ppg      (B, in_channels, F)   device
 *   sine     (B, 1, T)             device   T = F * prod(upsampling_scales)
 *   lft      (B, 1, T)             device
 *   spk_emb  (B, spk_emb_size)     device, or NULL (reference: spk_emb=None skips InstanceNorm and
 *                                  the speaker bias, fastsvc.py:134-140)
 *   out      (B, out_channels, T)  device, written
 *   lengths  NULL (all utterances F frames) or DEVICE pointer to B int32 frame counts, 1 <= lengths[b] <= F:
 *            a ragged batch.  Inputs and outputs keep the padded shapes above; utterance b is computed
 *            exactly as if it were run alone with lengths[b] frames (zero "same" padding at its own
 *            end, InstanceNorm over its own length); out[b, :, lengths[b]*hop:] is set to zero and the
 *            padding of the inputs is never read.  (The reference batches equal-length crops only.)
 *   dev_blob packed weights on the same device; workspace >= fastsvc_workspace_bytes(B, F)
 *   stream   hipStream_t (void* to keep this header free of HIP includes)
 * Launches are asynchronous on `stream`; the caller synchronises. */
int fastsvc_forward(const fastsvc_plan* plan, const void* dev_blob,
                    const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                    float* out, int32_t B, int32_t F, const int32_t* lengths,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Creates (and calibrates, see the conventions above: synchronises `stream`) the helper streams / events
 * fastsvc_forward may use for `stream` on the current device, so that the forward itself allocates nothing and
 * never synchronises (call once per stream, e.g. before graph capture or a latency-critical first call).
 * Idempotent. */
int fastsvc_stream_prepare(void* stream);

/* Frees the helper streams / events held for `stream` on the current device (after waiting for whatever
 * they still run).  Call it BEFORE destroying a stream that forwards were issued on: contexts are keyed by
 * the stream handle, are otherwise kept for the life of the process, and a new stream that reuses the
 * handle value would inherit the old one's.  No forward may be in flight on `stream` from another host
 * thread.  Returns 1 if a context was freed, 0 if there was none. */
int fastsvc_stream_release(void* stream);

/* Host-side number formats of the half-precision-MFMA kernels (csrc/fastsvc_hx.hip), exported so that the
 * packer's conversions can be pinned against an independent implementation (tests/test_boundary.py):
 * for each of n floats x:  f16_hi = binary16(x), f16_lo = binary16(x - f16_hi)  (the split-half pieces:
 * x = hi + lo to 22 significand bits) and bf16 = bfloat16(x); all round-to-nearest-even, raw bit patterns.
 * Any output pointer may be NULL. */
void fastsvc_split_half(const float* x, int64_t n, uint16_t* f16_hi, uint16_t* f16_lo, uint16_t* bf16);

/* Test / profiling support: location of a named intermediate tensor inside the workspace after a
 * forward (names as in oracle/fastsvc_oracle.py taps: "down_lft.0", "scale.2", "up.1.xmid", ...).
 * Returns 0 and fills byte offset / element count / shape (up to 3 dims: B, C, T). */
int fastsvc_workspace_tap(const fastsvc_plan* plan, int32_t B, int32_t F, const char* tap_name,
                          size_t* byte_offset, int64_t* numel, int64_t shape3[3]);

/* Optional device-side autotuning for one problem size (the analogue of the reference's
 * `torch.backends.cudnn.benchmark = True`, train_fastsvc.py:617): runs one forward in which every
 * convolution times its candidate launch shapes (time tile x channel split x tiles per workgroup)
 * on `stream`, keeps the fastest per layer in the plan (thread-safe cache keyed by layer, B, T) and
 * synchronises the stream.  `out` holds a valid forward result afterwards.  Without this call the
 * launch shapes come from a static cost model.  n_trials (nullable) returns the number of timed
 * trial configurations. */
int fastsvc_autotune(const fastsvc_plan* plan, const void* dev_blob,
                     const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                     float* out, int32_t B, int32_t F,
                     void* workspace, size_t workspace_bytes, void* stream, int32_t* n_trials);

/* Launch-shape table.  fastsvc_autotune stores its winners in the plan under keys
 * "<layer>|<B>|<T>" ("<layer>|<B>|<T>|b" while the plan uses bfloat16 activation storage, whose variants
 * compile under different register budgets); these entry points export them and load them back (a table measured once on an
 * MI355X ships as svcc23_fastsvc_amd/tuned_mi355x.json, so production runs need no trial launches).
 * An entry whose shape is not compiled for that layer is ignored at launch time (cost model instead).
 *   fastsvc_tuned_count: number of entries;
 *   fastsvc_tuned_get:   entry `index` in key order; key_out holds >= 96 bytes; shape = NW, WM, WN,
 *                        tiles per workgroup, algorithm (0 = as launched, 1 = Winograd F(2,3) along time);
 *   fastsvc_tuned_set:   insert / replace one entry. */
int fastsvc_tuned_count(const fastsvc_plan* plan);
int fastsvc_tuned_get(const fastsvc_plan* plan, int32_t index, char* key_out, int32_t shape_out[5]);
int fastsvc_tuned_set(const fastsvc_plan* plan, const char* key, const int32_t shape[5]);

/* Per-launch timing of one forward (bench.py roofline accounting).  Same arguments as
 * fastsvc_forward; brackets every kernel launch with hipEvents on `stream`, synchronises the
 * stream at the end and fills one record per launch: the layer it computes, the kernel symbol
 * (template instance) it ran, its algorithmic FLOPs (2*MAC, padding excluded) and algorithmic HBM
 * bytes (each operand tensor once + packed weights once), and the measured duration.  The profiled
 * forward runs on `stream` ONLY (no helper streams), so every kernel is timed running alone. */
typedef struct fastsvc_launch_record {
    char layer[64];
    char kernel[40];
    double flops;
    double bytes;
    float ms;
} fastsvc_launch_record;

int fastsvc_forward_profile(const fastsvc_plan* plan, const void* dev_blob,
                            const float* ppg, const float* sine, const float* lft, const float* spk_emb,
                            float* out, int32_t B, int32_t F, const int32_t* lengths,
                            void* workspace, size_t workspace_bytes, void* stream,
                            fastsvc_launch_record* records, int32_t max_records, int32_t* n_records);

/* Number of kernel launches one forward enqueues at most (fused launches of the conditioning chains, chosen per
 * call from its shapes and the launch table, lower it by up to n_stages + 1) and algorithmic FLOPs (2*MAC of every conv /
 * linear, de-duplicated dataflow) per output sample - used by bench.py's roofline accounting. */
int fastsvc_forward_launch_count(const fastsvc_plan* plan, int32_t with_spk_emb);
double fastsvc_flops_per_sample(const fastsvc_plan* plan);

/* ---- the step right before the forward inside `inference()` ----
 * Replaces SignalGenerator.__call__ (harana/utils/features.py:144-213; called at fastsvc.py:381):
 *   f0   (B, 1, F) device, Hz, 0 = unvoiced
 *   out  (B, ntypes, F*hop) device; channel k is signal type types[k]: 0 = "noise", 1 = "sine"
 *        (NSF sine + voiced/unvoiced noise, features.py:177-197), 2 = "uv"
 *   scratch  fastsvc_signal_scratch_bytes(B, F) device bytes (per-frame phase prefix, f64)
 * Deterministic for a given seed (the reference draws torch.randn). Asynchronous on `stream`. */
size_t fastsvc_signal_scratch_bytes(int32_t B, int32_t F);
int fastsvc_signal_generate(const float* f0, float* out, void* scratch, int32_t B, int32_t F, int32_t hop,
                            float sample_rate, float sine_amp, float noise_amp,
                            const int32_t* types, int32_t ntypes, uint64_t seed, void* stream);

/* Batch assembly for ragged batches (csrc/fastsvc_stage.hip; the reference decodes one utterance at a time,
 * decode_fastsvc.py:160-200, and has no counterpart): utterance b's tensor is a (C, lens[b]) float32 block at
 * src[b] ON THE DEVICE whose rows are pitches[b] elements apart; dst (B, C, width) receives them zero-padded to
 * `width` columns (lens[b] <= width).  `src`, `lens`, `pitches` are HOST arrays of B entries, read during the call
 * (their values travel in the kernel arguments: nothing to keep alive, no table upload).  One launch per 64
 * utterances on `stream`. */
int fastsvc_gather_padded(const float* const* src, const int32_t* lens, const int32_t* pitches, float* dst,
                          int32_t B, int32_t C, int32_t width, void* stream);

/* ---- the producer of the generator's loudness input ----
 * Replaces loudness_extract(audio, sampling_rate, hop_length) (harana/bin/preprocess_fastsvc.py:60-75; librosa
 * 0.8.1 stft n_fft 2048 / periodic Hann / reflect padding, perceptual (A) weighting with the 80 dB floor below
 * the utterance maximum, db_to_amplitude, log(mean over bins + 1e-5), nearest stretch by the hop):
 *   audio (B, T) device float32, one utterance per row;  out (B, frames * hop), frames = 1 + T / hop
 *   scratch  fastsvc_loudness_scratch_bytes(B, T, hop) device bytes (power spectrogram + per-utterance maximum)
 * Asynchronous on `stream`. */
int32_t fastsvc_loudness_frames(int32_t T, int32_t hop);
size_t fastsvc_loudness_scratch_bytes(int32_t B, int32_t T, int32_t hop);
int fastsvc_loudness_extract(const float* audio, float* out, void* scratch, int32_t B, int32_t T, int32_t hop,
                             float sample_rate, void* stream);

/* ---- the multi-resolution STFT loss of the training step, forward and backward ----
 * Replaces MultiResolutionSTFTLoss.forward(x, y) (harana/losses/stft_loss.py:131-180; STFTLoss :100-128, the magnitude
 * stft() :21-51, SpectralConvergenceLoss :54-74, LogSTFTMagnitudeLoss :77-97) and the gradient autograd derives from
 * it for the predicted signal x (the call site: harana/bin/train_fastsvc.py:163-170):
 *   x, y      (B, T) device float32: predicted / ground-truth waveforms, one per row
 *   n_res resolutions: fft_sizes[i] (a power of two in [8, 2048]: anything else FASTSVC_E_UNSUPPORTED), hop_sizes[i],
 *             win_lengths[i] <= fft_sizes[i] (HOST arrays); windows[i] = device pointer to win_lengths[i] window values
 *             (HOST array of pointers: whatever getattr(torch, window)(win_length) gives; centred in the frame)
 *   frames are centred with reflect padding (fft_size / 2 < T, else FASTSVC_E_INVALID - torch raises there too)
 *   loss      device float32 [2]: (spectral convergence, log STFT magnitude), each averaged over the resolutions
 *   scratch   fastsvc_stft_loss_scratch_bytes(...) device bytes; the forward leaves the per-resolution sums there,
 *             the backward of the SAME (x, y) reads them and uses the rest for the frame gradients
 *   grad_loss device float32 [2]: incoming gradients of (sc, mag);  grad_x (B, T): d(grad_loss . loss) / dx
 * No atomics: loss and gradient are bit-reproducible.  Asynchronous on `stream`. */
size_t fastsvc_stft_loss_scratch_bytes(int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes, const int32_t* hop_sizes);
int fastsvc_stft_loss_forward(const float* x, const float* y, int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes,
                              const int32_t* hop_sizes, const int32_t* win_lengths, const float* const* windows,
                              float* loss, void* scratch, void* stream);
int fastsvc_stft_loss_backward(const float* x, const float* y, int32_t B, int32_t T, int32_t n_res, const int32_t* fft_sizes,
                               const int32_t* hop_sizes, const int32_t* win_lengths, const float* const* windows,
                               const float* grad_loss, float* grad_x, void* scratch, void* stream);

/* ---- the grouped strided convolutions of the recipe's discriminator, forward and backward ----
 * Replaces, for the downsampling layers of MelGANDiscriminator (harana/models/fastsvc.py:386-520: Conv1d(c, min(4c, 512),
 * kernel_size = 41, stride = 4, padding = 20, groups = c / 4) + LeakyReLU; yaml egs/svcc23/fastsvc1/conf/fastsvc.yaml:34-52), what
 * the trainer's discriminator calls and their autograd run (harana/bin/train_fastsvc.py:172-175,207-224):
 *   fastsvc_gconv1d_forward          y = lrelu_slope(bias + grouped_conv(x, w))        x (B, Cin, T), w (Cout, Cin / groups, K),
 *                                    y (B, Cout, Tout), Tout = (T + 2 pad - K) / stride + 1; slope = 1: no activation
 *   fastsvc_gconv1d_backward_data    dx = grouped_conv_transpose(dy * lrelu'(y_act), w)  (y_act = the forward's output, or NULL)
 *   fastsvc_gconv1d_backward_weight  dw[o, i, k] = sum_{b, t} dy'[b, o, t] x[b, g(o) Ig + i, stride t + k - pad], dbias[o] = sum dy'
 *                                    (dbias may be NULL); scratch: fastsvc_gconv1d_backward_weight_scratch_bytes(B, Cout, T)
 *                                    device bytes (per-slab partial sums added in a fixed order: bit-reproducible)
 * Device float32, contiguous.  Supported (fastsvc_gconv1d_supported != 0): Cin / groups = 4, Cout / groups in {8, 16}, K = 41,
 * stride = 4, pad = 20; anything else FASTSVC_E_UNSUPPORTED (the caller keeps its own convolution).  Asynchronous on `stream`. */
int fastsvc_gconv1d_supported(int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad);
int fastsvc_gconv1d_forward(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t Cout,
                            int32_t groups, int32_t T, int32_t K, int32_t stride, int32_t pad, float slope, void* stream);
int fastsvc_gconv1d_backward_data(const float* dy, const float* y_act, const float* w, float* dx, int32_t B, int32_t Cin,
                                  int32_t Cout, int32_t groups, int32_t T, int32_t K, int32_t stride, int32_t pad, float slope,
                                  void* stream);
size_t fastsvc_gconv1d_backward_weight_scratch_bytes(int32_t B, int32_t Cout, int32_t T);
int fastsvc_gconv1d_backward_weight(const float* x, const float* dy, const float* y_act, float* dw, float* dbias, void* scratch,
                                    int32_t B, int32_t Cin, int32_t Cout, int32_t groups, int32_t T, int32_t K, int32_t stride,
                                    int32_t pad, float slope, void* stream);

/* ---- the convolutions of the generator's backward pass (float32 matrix-core kernels) ----
 * What autograd runs for every Conv1d / Conv2d(1 x k) of the generator when the reference trainer calls
 * gen_loss.backward() (harana/bin/train_fastsvc.py:183; the layers: harana/layers/residual_block.py:27-48,
 * harana/models/fastsvc.py:34-232 - all stride 1, "same" zero padding, k = 1 or 3, dilation d with (k / 2) d <= 27):
 *   fastsvc_conv1d_forward   y[b, o, t] = bias[o] + sum_{i, k} w[o, i, k] x[b, i, t + (k - k/2) d]
 *       x (B, Cin, T), y (B, Cout, T), w (Cout, Cin, K), bias (Cout) or NULL - device float32, contiguous.
 *       transposed != 0: w is read as (Cin, Cout, K) with flipped taps - with x = dy and w the forward weight this IS the
 *       backward-data convolution dx = conv_transpose(dy, w) (Cin = the forward's Cout, Cout = the forward's Cin).
 *   fastsvc_conv1d_backward_weight   dw[o, i, k] = sum_{b, t} dy[b, o, t] x[b, i, t + (k - k/2) d],  dbias[o] = sum dy
 *       (dbias may be NULL); both are overwritten.  scratch: fastsvc_conv1d_backward_weight_scratch_bytes(...) device
 *       bytes (per-slab partial sums, added in a fixed order: bit-reproducible).
 * Anything else (even k, k > 3, longer halos): FASTSVC_E_UNSUPPORTED.  Asynchronous on `stream`. */
int fastsvc_conv1d_forward(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t Cout,
                           int32_t T, int32_t K, int32_t dilation, int32_t transposed, void* stream);
size_t fastsvc_conv1d_backward_weight_scratch_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t T, int32_t K);
int fastsvc_conv1d_backward_weight(const float* x, const float* dy, float* dw, float* dbias, void* scratch, int32_t B, int32_t Cin,
                                   int32_t Cout, int32_t T, int32_t K, int32_t dilation, void* stream);

/* ---- FiLM affine + InstanceNorm + speaker bias + LeakyReLU of the up blocks, forward and backward ----
 * `_feature_affine` with a speaker embedding (harana/models/fastsvc.py:115-139) followed by the LeakyReLU that opens the
 * next conv block (fastsvc.py:60-83), as ONE node of the training graph:
 *   out = lrelu((u - mean_t u) / sqrt(var_t u + eps) + bias[row]),  u = scale * x + shift
 *   x, scale, shift, out (rows, T) device float32 with rows = B * C; bias (rows) = emb_projector(normalize(spk_emb));
 *   mean, rstd (rows) are written by the forward and read by the backward of the same inputs
 *   backward: dx, dscale, dshift (rows, T) and dbias (rows) from dout
 * Asynchronous on `stream`. */
int fastsvc_film_norm_forward(const float* x, const float* scale, const float* shift, const float* bias, float* out, float* mean,
                              float* rstd, int32_t rows, int32_t T, float eps, float slope, void* stream);
int fastsvc_film_norm_backward(const float* dout, const float* x, const float* scale, const float* shift, const float* bias,
                               const float* mean, const float* rstd, float* dx, float* dscale, float* dshift, float* dbias,
                               int32_t rows, int32_t T, float slope, void* stream);

/* ---- weight normalisation w = g * v / ||v|| of up to 56 layers in ONE launch, forward and backward ----
 * torch.nn.utils.weight_norm as the reference applies it to every conv (harana/models/fastsvc.py:354-362; norm over all
 * dims but 0) and what autograd derives from it.  HOST arrays of n device pointers / sizes:
 *   v[i] (rows[i], cols[i]), g[i] (rows[i]);  forward writes w[i] (rows, cols) and norm[i] (rows);
 *   backward reads dw[i], norm[i] and writes dv[i] (rows, cols), dg[i] (rows).   n > 56: FASTSVC_E_UNSUPPORTED (call again for the rest). */
int fastsvc_weight_norm_forward(int32_t n, const float* const* v, const float* const* g, float* const* w, float* const* norm,
                                const int32_t* rows, const int32_t* cols, void* stream);
int fastsvc_weight_norm_backward(int32_t n, const float* const* v, const float* const* g, const float* const* dw,
                                 const float* const* norm, float* const* dv, float* const* dg, const int32_t* rows,
                                 const int32_t* cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTSVC_HIP_H */
