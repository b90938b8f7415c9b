"""ctypes binding of the C ABI in ``include/fastsvc_hip.h`` (``libfastsvc_hip.so``).

PyTorch is used for plumbing only: device memory (weight blob, workspace, outputs) comes from its
caching allocator and kernels are enqueued on its current HIP stream.  There is NO CPU fallback:
if the shared library is missing or the tensors are not on a GPU, this module raises.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .synth import GeneratorConfig

_LIB = None
# FASTSVC_HIP_LIB: load another build of the same ABI (tools/timeline.py points it at the stamped
# diagnostic library); there is no fallback of any kind - a missing library is an error
_LIB_PATH = os.environ.get("FASTSVC_HIP_LIB") or \
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfastsvc_hip.so")
MAX_STAGES = 8
# launch shapes measured once on an MI355X by tools/tune_shapes.py (fastsvc_autotune winners for
# the BASELINE.json workloads); other (B, F) fall back to the static cost model or model.autotune
TUNED_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_mi355x.json")

# every symbol include/fastsvc_hip.h declares (checked by tests/test_boundary.py)
ABI_SYMBOLS = (
    "fastsvc_abi_version", "fastsvc_last_error", "fastsvc_plan_create", "fastsvc_plan_destroy",
    "fastsvc_plan_set_storage", "fastsvc_plan_get_storage",
    "fastsvc_weight_blob_bytes", "fastsvc_pack_weights", "fastsvc_workspace_bytes",
    "fastsvc_forward", "fastsvc_autotune", "fastsvc_tuned_count", "fastsvc_tuned_get", "fastsvc_tuned_set",
    "fastsvc_forward_profile", "fastsvc_workspace_tap", "fastsvc_forward_launch_count",
    "fastsvc_flops_per_sample", "fastsvc_signal_scratch_bytes", "fastsvc_signal_generate",
    "fastsvc_stream_prepare", "fastsvc_stream_release", "fastsvc_split_half", "fastsvc_plan_set_workspace_mode",
    "fastsvc_loudness_frames", "fastsvc_loudness_scratch_bytes", "fastsvc_loudness_extract",
    "fastsvc_gather_padded",
    "fastsvc_stft_loss_scratch_bytes", "fastsvc_stft_loss_forward", "fastsvc_stft_loss_backward",
    "fastsvc_conv1d_forward", "fastsvc_conv1d_backward_weight", "fastsvc_conv1d_backward_weight_scratch_bytes",
    "fastsvc_film_norm_forward", "fastsvc_film_norm_backward", "fastsvc_weight_norm_forward", "fastsvc_weight_norm_backward",
    "fastsvc_gconv1d_supported", "fastsvc_gconv1d_forward", "fastsvc_gconv1d_backward_data",
    "fastsvc_gconv1d_backward_weight_scratch_bytes", "fastsvc_gconv1d_backward_weight",
)


class FastSVCError(RuntimeError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [
        ("in_channels", ctypes.c_int32),
        ("n_stages", ctypes.c_int32),
        ("mid_channels", ctypes.c_int32 * MAX_STAGES),
        ("upsampling_scales", ctypes.c_int32 * MAX_STAGES),
        ("out_channels", ctypes.c_int32),
        ("spk_emb_size", ctypes.c_int32),
        ("use_spk_emb", ctypes.c_int32),
    ]


class _LaunchRecord(ctypes.Structure):
    _fields_ = [("layer", ctypes.c_char * 64), ("kernel", ctypes.c_char * 40),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double), ("ms", ctypes.c_float)]


class _Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float)), ("numel", ctypes.c_int64)]


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the in-tree gfx950 library; raise (never fall back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise FastSVCError(
            f"{_LIB_PATH} not found: build it with `python -m svcc23_fastsvc_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = ctypes.CDLL(_LIB_PATH)
    vp, sz, i32, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int64
    lib.fastsvc_abi_version.restype = ctypes.c_int
    lib.fastsvc_last_error.restype = ctypes.c_char_p
    lib.fastsvc_plan_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(vp)]
    lib.fastsvc_plan_create.restype = ctypes.c_int
    lib.fastsvc_plan_destroy.argtypes = [vp]
    lib.fastsvc_plan_destroy.restype = None
    lib.fastsvc_plan_set_storage.argtypes = [vp, i32]
    lib.fastsvc_plan_set_storage.restype = ctypes.c_int
    lib.fastsvc_plan_get_storage.argtypes = [vp]
    lib.fastsvc_plan_get_storage.restype = ctypes.c_int
    lib.fastsvc_weight_blob_bytes.argtypes = [vp]
    lib.fastsvc_weight_blob_bytes.restype = sz
    lib.fastsvc_pack_weights.argtypes = [vp, ctypes.POINTER(_Tensor), i32, vp]
    lib.fastsvc_pack_weights.restype = ctypes.c_int
    lib.fastsvc_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fastsvc_workspace_bytes.restype = sz
    lib.fastsvc_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.fastsvc_forward.restype = ctypes.c_int
    lib.fastsvc_loudness_frames.argtypes = [i32, i32]
    lib.fastsvc_loudness_frames.restype = i32
    lib.fastsvc_loudness_scratch_bytes.argtypes = [i32, i32, i32]
    lib.fastsvc_loudness_scratch_bytes.restype = sz
    lib.fastsvc_loudness_extract.argtypes = [vp, vp, vp, i32, i32, i32, ctypes.c_float, vp]
    lib.fastsvc_loudness_extract.restype = ctypes.c_int
    lib.fastsvc_stft_loss_scratch_bytes.argtypes = [i32, i32, i32, vp, vp]
    lib.fastsvc_stft_loss_scratch_bytes.restype = sz
    lib.fastsvc_stft_loss_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.fastsvc_stft_loss_forward.restype = ctypes.c_int
    lib.fastsvc_stft_loss_backward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.fastsvc_stft_loss_backward.restype = ctypes.c_int
    lib.fastsvc_conv1d_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.fastsvc_conv1d_forward.restype = ctypes.c_int
    lib.fastsvc_conv1d_backward_weight_scratch_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.fastsvc_conv1d_backward_weight_scratch_bytes.restype = sz
    lib.fastsvc_conv1d_backward_weight.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.fastsvc_conv1d_backward_weight.restype = ctypes.c_int
    lib.fastsvc_gconv1d_supported.argtypes = [i32] * 6
    lib.fastsvc_gconv1d_supported.restype = ctypes.c_int
    lib.fastsvc_gconv1d_forward.argtypes = [vp, vp, vp, vp] + [i32] * 8 + [ctypes.c_float, vp]
    lib.fastsvc_gconv1d_forward.restype = ctypes.c_int
    lib.fastsvc_gconv1d_backward_data.argtypes = [vp, vp, vp, vp] + [i32] * 8 + [ctypes.c_float, vp]
    lib.fastsvc_gconv1d_backward_data.restype = ctypes.c_int
    lib.fastsvc_gconv1d_backward_weight_scratch_bytes.argtypes = [i32, i32, i32]
    lib.fastsvc_gconv1d_backward_weight_scratch_bytes.restype = sz
    lib.fastsvc_gconv1d_backward_weight.argtypes = [vp, vp, vp, vp, vp, vp] + [i32] * 8 + [ctypes.c_float, vp]
    lib.fastsvc_gconv1d_backward_weight.restype = ctypes.c_int
    lib.fastsvc_film_norm_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, ctypes.c_float, ctypes.c_float, vp]
    lib.fastsvc_film_norm_forward.restype = ctypes.c_int
    lib.fastsvc_film_norm_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, ctypes.c_float, vp]
    lib.fastsvc_film_norm_backward.restype = ctypes.c_int
    lib.fastsvc_weight_norm_forward.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.fastsvc_weight_norm_forward.restype = ctypes.c_int
    lib.fastsvc_weight_norm_backward.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.fastsvc_weight_norm_backward.restype = ctypes.c_int
    lib.fastsvc_split_half.argtypes = [vp, i64, vp, vp, vp]
    lib.fastsvc_split_half.restype = None
    lib.fastsvc_stream_prepare.argtypes = [vp]
    lib.fastsvc_stream_prepare.restype = ctypes.c_int
    lib.fastsvc_stream_release.argtypes = [vp]
    lib.fastsvc_stream_release.restype = ctypes.c_int
    lib.fastsvc_gather_padded.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32), vp, i32, i32, i32, vp]
    lib.fastsvc_gather_padded.restype = ctypes.c_int
    lib.fastsvc_autotune.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, sz, vp, ctypes.POINTER(i32)]
    lib.fastsvc_autotune.restype = ctypes.c_int
    lib.fastsvc_tuned_count.argtypes = [vp]
    lib.fastsvc_tuned_count.restype = ctypes.c_int
    lib.fastsvc_tuned_get.argtypes = [vp, i32, ctypes.c_char_p, ctypes.POINTER(i32 * 5)]
    lib.fastsvc_tuned_get.restype = ctypes.c_int
    lib.fastsvc_tuned_set.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32 * 5)]
    lib.fastsvc_tuned_set.restype = ctypes.c_int
    lib.fastsvc_forward_profile.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, sz, vp,
                                            ctypes.POINTER(_LaunchRecord), i32, ctypes.POINTER(i32)]
    lib.fastsvc_forward_profile.restype = ctypes.c_int
    lib.fastsvc_workspace_tap.argtypes = [vp, i32, i32, ctypes.c_char_p, ctypes.POINTER(sz),
                                          ctypes.POINTER(i64), ctypes.POINTER(i64 * 3)]
    lib.fastsvc_workspace_tap.restype = ctypes.c_int
    lib.fastsvc_plan_set_workspace_mode.argtypes = [vp, i32]
    lib.fastsvc_plan_set_workspace_mode.restype = ctypes.c_int
    lib.fastsvc_forward_launch_count.argtypes = [vp, i32]
    lib.fastsvc_forward_launch_count.restype = ctypes.c_int
    lib.fastsvc_flops_per_sample.argtypes = [vp]
    lib.fastsvc_flops_per_sample.restype = ctypes.c_double
    lib.fastsvc_signal_scratch_bytes.argtypes = [i32, i32]
    lib.fastsvc_signal_scratch_bytes.restype = sz
    lib.fastsvc_signal_generate.argtypes = [vp, vp, vp, i32, i32, i32, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.POINTER(i32), i32, ctypes.c_uint64, vp]
    lib.fastsvc_signal_generate.restype = ctypes.c_int
    if lib.fastsvc_abi_version() != 1:
        raise FastSVCError("libfastsvc_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def gather_padded(rows: Sequence[torch.Tensor], width: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Assemble a zero-padded batch from B device tensors of shape (C, len_b) (len_b <= width; float32, unit stride
    along time, any row pitch - views of larger tensors are fine): -> (B, C, width).  One HIP launch per 64
    utterances on the current stream (fastsvc_gather_padded, csrc/fastsvc_stage.hip); fails loudly off the GPU."""
    lib = load_library()
    B = len(rows)
    if B == 0:
        raise ValueError("gather_padded needs at least one utterance")
    first = rows[0]
    if not first.is_cuda:
        raise FastSVCError("gather_padded needs GPU tensors (no CPU fallback); got " + str(first.device))
    C = int(first.shape[0])
    for t in rows:
        if t.dim() != 2 or t.shape[0] != C or t.dtype != torch.float32 or t.device != first.device or \
                (t.shape[1] > 1 and t.stride(1) != 1) or t.shape[1] > width:
            raise ValueError(f"gather_padded: every utterance must be a float32 (C={C}, len <= {width}) tensor with unit "
                             f"time stride on {first.device}; got {tuple(t.shape)} {t.dtype} strides {t.stride()}")
    if out is None:
        out = torch.empty((B, C, width), dtype=torch.float32, device=first.device)
    elif tuple(out.shape) != (B, C, width) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != first.device:
        raise ValueError(f"out must be a contiguous float32 {(B, C, width)} tensor on {first.device}")
    src = (ctypes.c_void_p * B)(*[t.data_ptr() for t in rows])
    lens = (ctypes.c_int32 * B)(*[int(t.shape[1]) for t in rows])
    pitches = (ctypes.c_int32 * B)(*[int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[1]), 1) for t in rows])
    with torch.cuda.device(first.device):
        stream = torch.cuda.current_stream(first.device).cuda_stream
        _check(lib, lib.fastsvc_gather_padded(src, lens, pitches, ctypes.c_void_p(out.data_ptr()), B, C, width,
                                              ctypes.c_void_p(stream)), "fastsvc_gather_padded")
    return out


def _check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.fastsvc_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise FastSVCError(f"{what} failed ({rc}): {msg}")


class Plan:
    """Host-only plan (layer table + blob / workspace layout) for one generator configuration."""

    def __init__(self, cfg: GeneratorConfig, load_shipped_table: bool = True, storage: str = "float32",
                 compact_workspace: bool = False):
        """``storage``: "float32" (default, the parity path) or "bfloat16" - every workspace tensor is
        stored as bf16 (half the HBM traffic of the narrow layers, half the workspace; fp32 arithmetic;
        bf16-activation accuracy, frame counts must be multiples of 4).
        ``compact_workspace``: intermediates of different stages share buffers (about 40 % less memory for long
        batches); the ``tap`` of a shared buffer then holds the last stage's tensor only."""
        if storage not in ("float32", "bfloat16"):
            raise ValueError("storage must be 'float32' or 'bfloat16'")
        self.storage = storage
        self.cfg = cfg
        self.lib = load_library()
        if cfg.n_stages > MAX_STAGES or len(cfg.upsampling_scales) != cfg.n_stages:
            raise ValueError("mid_channels / upsampling_scales must have equal length <= 8")
        c = _Config()
        c.in_channels = cfg.in_channels
        c.n_stages = cfg.n_stages
        for i in range(cfg.n_stages):
            c.mid_channels[i] = cfg.mid_channels[i]
            c.upsampling_scales[i] = cfg.upsampling_scales[i]
        c.out_channels = cfg.out_channels
        c.spk_emb_size = cfg.spk_emb_size
        c.use_spk_emb = 1 if cfg.use_spk_emb else 0
        handle = ctypes.c_void_p()
        _check(self.lib, self.lib.fastsvc_plan_create(ctypes.byref(c), ctypes.byref(handle)), "fastsvc_plan_create")
        self._h = handle
        if storage == "bfloat16":
            _check(self.lib, self.lib.fastsvc_plan_set_storage(handle, 1), "fastsvc_plan_set_storage")
            # (bfloat16 launches look their shapes up under "<layer>|<B>|<T>|b": separate entries of the same table)
        self.compact_workspace = bool(compact_workspace)
        if compact_workspace:
            _check(self.lib, self.lib.fastsvc_plan_set_workspace_mode(handle, 1), "fastsvc_plan_set_workspace_mode")
        self.last_autotune_trials = 0
        self.pad_odd_lengths = True          # float32 storage: run F % 4 != 0 batches padded (see padded_frames)
        if load_shipped_table:
            self.load_tuned_file(TUNED_TABLE_PATH, missing_ok=True)

    # ---- launch-shape table (fastsvc_autotune winners) ----
    def config_signature(self) -> str:
        c = self.cfg
        return "in%d_mid%s_up%s_out%d_spk%d" % (c.in_channels, "-".join(map(str, c.mid_channels)),
                                                "-".join(map(str, c.upsampling_scales)), c.out_channels,
                                                c.spk_emb_size if c.use_spk_emb else 0)

    def tuned_shapes(self) -> dict:
        """{"<layer>|<B>|<T>": [NW, WM, WN, tiles_per_workgroup, algorithm]} currently held by the plan."""
        out = {}
        key = ctypes.create_string_buffer(96)
        shape = (ctypes.c_int32 * 5)()
        for i in range(int(self.lib.fastsvc_tuned_count(self._h))):
            _check(self.lib, self.lib.fastsvc_tuned_get(self._h, i, key, ctypes.byref(shape)), "fastsvc_tuned_get")
            out[key.value.decode()] = [int(v) for v in shape]
        return out

    def load_tuned(self, table: Mapping[str, Sequence[int]]) -> int:
        for k, v in table.items():
            v = list(v) + [0] * (5 - len(v))                  # older 4-entry tables: algorithm 0
            shape = (ctypes.c_int32 * 5)(*[int(x) for x in v])
            _check(self.lib, self.lib.fastsvc_tuned_set(self._h, k.encode(), ctypes.byref(shape)), "fastsvc_tuned_set")
        return len(table)

    def keep_last_block_output(self, B: int, F: int) -> None:
        """By default ``conv_last`` rides on the last block's final launch and the block's C-channel output is not
        written.  This keeps the two launches for batches of this shape (launch-table entry with algorithm 0
        under ``conv_last|B|T``), so that the ``up.<n-1>.out`` workspace tap holds the tensor."""
        T = F
        for sc in self.cfg.upsampling_scales:
            T *= int(sc)
        self.load_tuned({f"conv_last|{B}|{T}": [1, 1, 4, 1, 0], f"conv_last|{B}|{T}|b": [1, 1, 4, 1, 0]})

    def fuse_block_heads(self, B: int, F: int, fused: bool = True) -> None:
        """The head of an up block - ``conv_first`` and the two stretched convs behind it - can run as ONE launch
        (kernel mode 8, float32 storage, rows a multiple of 4 long: the tensor ``a`` between them never leaves LDS).
        It costs about what the three launches cost, so it runs only where the launch table holds algorithm 3 under
        ``up.<i>.head|B|T_in``; this sets those entries for a batch shape (``fused=False``: algorithm 0, the three
        launches whatever a loaded table says - the ``up.<i>.a`` taps then hold the tensor)."""
        T = F
        table = {}
        for i, sc in enumerate(self.cfg.upsampling_scales):
            table[f"up.{i}.head|{B}|{T}"] = [2, 1, 4, 1, 3 if fused else 0]
            T *= int(sc)
        self.load_tuned(table)

    def keep_block_heads_separate(self, B: int, F: int) -> None:
        self.fuse_block_heads(B, F, fused=False)
        self.keep_residual_convs_separate(B, F)

    def keep_residual_convs_separate(self, B: int, F: int) -> None:
        """By default the stretched residual conv of an up block is folded into the block's d = 3 conv (launch
        ``up.<i>.d3x``: the tensor ``xr`` is never written).  This keeps the two launches for batches of this shape
        (algorithm 0 under ``up.<i>.d3x|B|T_out``), so that the ``up.<i>.xr`` workspace taps hold the tensor."""
        T = F
        table = {}
        for i, sc in enumerate(self.cfg.upsampling_scales):
            T *= int(sc)
            table[f"up.{i}.d3x|{B}|{T}"] = [2, 1, 4, 1, 0]
            table[f"up.{i}.d3x|{B}|{T}|b"] = [2, 1, 4, 1, 0]
        self.load_tuned(table)

    def load_tuned_file(self, path: str, missing_ok: bool = False) -> int:
        """Load this configuration's section of a tuned-shape JSON file (tools/tune_shapes.py)."""
        if not os.path.exists(path):
            if missing_ok:
                return 0
            raise FileNotFoundError(path)
        with open(path) as f:
            doc = json.load(f)
        return self.load_tuned(doc.get("tables", {}).get(self.config_signature(), {}))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.fastsvc_plan_destroy(h)
            self._h = None

    @property
    def arithmetic(self) -> str:
        """The type the path computes in, for bench.py's `dtype` (FASTSVC_HX=0 selects the f32-input MFMA kernels
        everywhere; by default every k=3 convolution whose rows are a multiple of 4 long runs on the
        half-precision MFMA kernels of csrc/fastsvc_hx.hip)."""
        hx = os.environ.get("FASTSVC_HX", "1") != "0"
        if self.storage == "float32":
            return ("f32 (conv products as split-binary16 pairs on the f16 MFMA, x*w = xh*wh + xh*wl + xl*wh with f32 "
                    "accumulation; weights scaled per output channel and activations per tensor and utterance by exact "
                    "powers of two into binary16's range, undone in the epilogue: fp32-class, 4e-6 of the f32-MFMA path, "
                    "held over 2^-20..2^8 input / weight scales by tests/test_dynamic_range_gpu.py)") if hx else "f32"
        return ("bf16 (bf16 MFMA products, f32 accumulate, bf16 activation storage)" if hx
                else "f32 arithmetic, bf16 activation storage")

    @property
    def blob_bytes(self) -> int:
        return int(self.lib.fastsvc_weight_blob_bytes(self._h))

    @property
    def flops_per_sample(self) -> float:
        return float(self.lib.fastsvc_flops_per_sample(self._h))

    def launch_count(self, with_spk: bool = True) -> int:
        return int(self.lib.fastsvc_forward_launch_count(self._h, 1 if with_spk else 0))

    def prepare_stream(self, device=None) -> None:
        """Create the helper streams / events of the current HIP stream of `device` ahead of the first
        forward on it (so that forward allocates nothing)."""
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _check(self.lib, self.lib.fastsvc_stream_prepare(ctypes.c_void_p(stream)), "fastsvc_stream_prepare")

    def release_stream(self, stream=None, device=None) -> bool:
        """Free the helper streams / events the library holds for `stream` (a torch.cuda.Stream; default: the
        current one) - to be called before a short-lived stream that ran forwards is dropped."""
        with torch.cuda.device(device):
            st = (stream or torch.cuda.current_stream(device)).cuda_stream
            return bool(self.lib.fastsvc_stream_release(ctypes.c_void_p(st)))

    def workspace_bytes(self, B: int, F: int) -> int:
        return int(self.lib.fastsvc_workspace_bytes(self._h, B, F))

    def padded_frames(self, F: int) -> int:
        """Frame count the library actually runs for an F-frame batch: the next multiple of 4 (the padded batch is run
        as a ragged one).  bfloat16 storage needs it; float32 storage takes any F, but rows that are not a multiple of
        4 long (F-rate and 2F-rate tensors) send their layers to the slower gathered kernels - 64 x 1499 frames took
        27.0 ms against 22.5 ms for 64 x 1500 (tools/ragged_check.py) - so `Plan.forward` pads there too."""
        return F + ((-F) % 4)

    def pack_prefetch(self, state_dict: Mapping[str, object]):
        """Start the device-to-host copy of a state dict's device tensors (one gather, one asynchronous copy into a
        page-locked buffer, one event on the current stream) and return a handle for ``pack(..., prefetched=handle)``.
        A training step calls this right behind the optimizer update: the copy and the host-side packing then overlap
        whatever the GPU is given next instead of draining the stream first."""
        items = list(state_dict.items())
        on_dev = [i for i, (_, v) in enumerate(items) if isinstance(v, torch.Tensor) and v.device.type != "cpu"]
        if not on_dev:
            return None
        flat = torch.cat([items[i][1].detach().reshape(-1).to(torch.float32) for i in on_dev])
        buf = getattr(self, "_pinned_params", None)
        if buf is None or buf.numel() < flat.numel():
            buf = self._pinned_params = torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True)
        buf[: flat.numel()].copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flat.device))
        return {"event": ev, "host": buf[: flat.numel()], "numels": [items[i][1].numel() for i in on_dev], "keep": flat}

    def pack(self, state_dict: Mapping[str, object], reuse_pinned: bool = False, prefetched=None) -> torch.Tensor:
        """Fold weight-norm and pack a state dict (either key layout) into the kernel blob.

        Returns a CPU float32 tensor of ``blob_bytes`` bytes (upload or broadcast it).  Device tensors are gathered
        with ONE device-to-host copy (a training step re-packs after every optimizer update: 251 separate copies were
        a third of that).  ``reuse_pinned``: return this plan's page-locked staging buffer (overwritten by the next
        such call) - for callers that upload it right away."""
        items = list(state_dict.items())
        host = [None] * len(items)
        on_dev = [i for i, (_, v) in enumerate(items) if isinstance(v, torch.Tensor) and v.device.type != "cpu"]
        if on_dev:
            if prefetched is not None and prefetched["numels"] == [items[i][1].numel() for i in on_dev]:
                prefetched["event"].synchronize()            # the copy issued by pack_prefetch (same tensors, same order)
                flat = prefetched["host"].numpy()
            else:
                flat = torch.cat([items[i][1].detach().reshape(-1).to(torch.float32) for i in on_dev]).cpu().numpy()
            o = 0
            for i in on_dev:
                n = items[i][1].numel()
                host[i] = flat[o:o + n]
                o += n
        keep = []
        arr = (_Tensor * len(items))()
        for i, (k, v) in enumerate(items):
            if host[i] is not None:
                a = host[i]
            else:
                if isinstance(v, torch.Tensor):
                    v = v.detach().to("cpu", torch.float32).contiguous().numpy()
                a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            name = k.encode("utf-8")
            keep.append((a, name))
            arr[i].name = name
            arr[i].data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            arr[i].numel = a.size
        if reuse_pinned:
            if getattr(self, "_pinned_blob", None) is None:
                self._pinned_blob = torch.empty(self.blob_bytes // 4, dtype=torch.float32,
                                                pin_memory=torch.cuda.is_available())
            blob = self._pinned_blob
        else:
            blob = torch.empty(self.blob_bytes // 4, dtype=torch.float32)
        rc = self.lib.fastsvc_pack_weights(self._h, arr, len(items), ctypes.c_void_p(blob.data_ptr()))
        if rc == -2:
            raise KeyError(self.lib.fastsvc_last_error().decode())
        _check(self.lib, rc, "fastsvc_pack_weights")
        return blob

    def tap_info(self, name: str, B: int, F: int) -> Tuple[int, int, Tuple[int, int, int]]:
        off = ctypes.c_size_t()
        numel = ctypes.c_int64()
        shape = (ctypes.c_int64 * 3)()
        _check(self.lib, self.lib.fastsvc_workspace_tap(self._h, B, F, name.encode(), ctypes.byref(off),
                                                        ctypes.byref(numel), ctypes.byref(shape)),
               "fastsvc_workspace_tap")
        return int(off.value), int(numel.value), (int(shape[0]), int(shape[1]), int(shape[2]))

    def tap(self, name: str, B: int, F: int, workspace: torch.Tensor) -> torch.Tensor:
        """View of a named intermediate inside ``workspace`` (valid after a forward)."""
        off, numel, shape = self.tap_info(name, B, F)
        if name.endswith(".stats"):
            return workspace[off: off + numel * 8].view(torch.float64).view(shape)
        if self.storage == "bfloat16" and name != "sig" and not name.endswith(".spk"):
            return workspace[off: off + numel * 2].view(torch.bfloat16).view(shape)
        return workspace[off: off + numel * 4].view(torch.float32).view(shape)

    _LENS_SLOTS = 32

    def _stage_lengths(self, lens_host: torch.Tensor, B: int, dev) -> torch.Tensor:
        """Frame counts -> device through a RING of page-locked slots owned by the plan.  A fresh pinned tensor per
        call looked harmless, but the host runs many batches ahead of the GPU: the pinned allocator cannot recycle a
        block whose copy has not run yet, so every forward of a pass over ragged batches paid a hipHostMalloc
        (≈ 3 ms each, serialised with the device: 187 ms instead of 145 ms for the 13 batches of cfg4var).  A slot is
        reused only after its copy has completed (event; the host waits only if it is 32 batches ahead)."""
        ring = getattr(self, "_lens_ring", None)
        if ring is None or ring["buf"].shape[1] < B:
            ring = {"buf": torch.empty((self._LENS_SLOTS, max(64, B)), dtype=torch.int32, pin_memory=True),
                    "ev": [None] * self._LENS_SLOTS, "i": 0}
            self._lens_ring = ring
        slot = ring["i"] % self._LENS_SLOTS
        ring["i"] += 1
        if ring["ev"][slot] is not None:
            ring["ev"][slot].synchronize()
        src = ring["buf"][slot, :B]
        src.copy_(lens_host)
        out = src.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ring["ev"][slot] = ev
        return out

    def forward(self, blob: torch.Tensor, ppg: torch.Tensor, sine: torch.Tensor, lft: torch.Tensor,
                spk_emb: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None, profile: Optional[list] = None,
                autotune: bool = False, lengths=None) -> torch.Tensor:
        """Enqueue one forward on the current HIP stream of ``ppg.device``; returns (B, O, T).

        ``lengths`` (B frame counts, 1 <= n <= F; sequence or int tensor) makes the batch ragged:
        inputs stay padded to F, utterance b is computed exactly as if run alone with lengths[b]
        frames and ``out[b, :, lengths[b]*hop:]`` is zero.

        With ``profile`` (a list) the launches are bracketed by hipEvents on that stream, the
        stream is synchronised and one dict per kernel launch is appended to the list.
        With ``autotune`` every convolution first times its candidate launch shapes for this
        (B, F) and the plan remembers the fastest (cf. ``cudnn.benchmark``); synchronises."""
        cfg = self.cfg
        if not ppg.is_cuda:
            raise FastSVCError("FastSVC HIP path needs GPU tensors (no CPU fallback); got " + str(ppg.device))
        dev = ppg.device
        for name, t in (("sine", sine), ("lft", lft), ("spk_emb", spk_emb), ("blob", blob)):
            if t is not None and t.device != dev:
                raise ValueError(f"{name} is on {t.device}, expected {dev}")
        if ppg.dim() != 3 or ppg.shape[1] != cfg.in_channels:
            raise ValueError(f"ppg must be (B, {cfg.in_channels}, F), got {tuple(ppg.shape)}")
        B, _, F = ppg.shape
        T = F * cfg.hop
        for name, t in (("sine", sine), ("lft", lft)):
            if tuple(t.shape) != (B, 1, T):
                raise ValueError(f"{name} must be (B, 1, F*{cfg.hop}) = {(B, 1, T)}, got {tuple(t.shape)}")
        if spk_emb is not None and tuple(spk_emb.shape) != (B, cfg.spk_emb_size):
            raise ValueError(f"spk_emb must be {(B, cfg.spk_emb_size)}, got {tuple(spk_emb.shape)}")
        ppg, sine, lft = (t.to(torch.float32).contiguous() for t in (ppg, sine, lft))
        if spk_emb is not None:
            spk_emb = spk_emb.to(torch.float32).contiguous()
        if F % 4 != 0 and profile is None and (self.storage == "bfloat16" or self.pad_odd_lengths):
            # bfloat16 storage moves 4 time steps per access at the frame rate, so the library wants F % 4 == 0
            # (three of four real utterances are not); float32 storage runs such rows on its slower kernels.  Pad to
            # the next multiple and run the padded batch as a ragged one - `lengths` makes every utterance exactly
            # what it would be alone at its own length.  The caller's workspace is used when it holds the padded
            # batch (size it with `workspace_bytes(B, padded_frames(F))`); the padded inputs live in one reused
            # staging set.  Autotuning tunes the padded shape (launch shapes are keyed by the padded row lengths).
            Fp = self.padded_frames(F)
            hop = cfg.hop
            if lengths is not None and not (isinstance(lengths, torch.Tensor) and lengths.is_cuda):
                lh = torch.as_tensor(lengths, dtype=torch.int64, device="cpu").reshape(-1)
                if lh.numel() != B or int(lh.min()) < 1 or int(lh.max()) > F:        # (against the caller's F, not the padded one)
                    raise ValueError(f"lengths must hold {B} frame counts in [1, {F}]")
            if workspace is not None and workspace.numel() < self.workspace_bytes(B, Fp):
                raise ValueError(f"workspace too small for the padded batch: size it with workspace_bytes({B}, padded_frames({F}) = {Fp})")
            # one staging set per (shape, device, STREAM): forwards of one plan on different streams must not share it
            key = (B, Fp, str(dev), int(torch.cuda.current_stream(dev).cuda_stream))
            sets = self.__dict__.setdefault("_pad_sets", {})
            if key not in sets:
                if len(sets) >= 8:
                    sets.pop(next(iter(sets)))
                sets[key] = [torch.zeros((B, cfg.in_channels, Fp), dtype=torch.float32, device=dev),
                             torch.zeros((B, 1, Fp * hop), dtype=torch.float32, device=dev),
                             torch.zeros((B, 1, Fp * hop), dtype=torch.float32, device=dev),
                             torch.empty((B, cfg.out_channels, Fp * hop), dtype=torch.float32, device=dev), F]
            pp, ps, pl, py, lastF = sets[key]
            pp[..., :F].copy_(ppg); ps[..., :T].copy_(sine); pl[..., :T].copy_(lft)
            if lastF > F:                                    # a longer batch used this set: its tail is not padding
                pp[..., F:lastF].zero_(); ps[..., T:lastF * hop].zero_(); pl[..., T:lastF * hop].zero_()
            sets[key][4] = F
            ok_ws = workspace is not None
            if autotune:
                if lengths is not None:
                    raise ValueError("autotune times full-length batches: call it without lengths")
                self.forward(blob, pp, ps, pl, spk_emb, out=py, workspace=workspace if ok_ws else None, autotune=True)
            self.forward(blob, pp, ps, pl, spk_emb, out=py, workspace=workspace if ok_ws else None,
                         lengths=[F] * B if lengths is None else lengths)
            if out is None:
                return py[..., :T].contiguous()
            out.copy_(py[..., :T])
            return out
        lens_dev = None
        if lengths is not None:
            if autotune:
                raise ValueError("autotune times full-length batches: call it without lengths")
            if isinstance(lengths, torch.Tensor) and lengths.is_cuda:
                # already on the device: used as is (int32, B entries; range checked by the caller - reading it back
                # here would synchronise)
                if lengths.device != dev or lengths.numel() != B:
                    raise ValueError(f"lengths must hold {B} frame counts on {dev}")
                lens_dev = lengths.reshape(-1).to(torch.int32).contiguous()
            else:
                lens_host = torch.as_tensor(lengths, dtype=torch.int64, device="cpu").reshape(-1)
                if lens_host.numel() != B or int(lens_host.min()) < 1 or int(lens_host.max()) > F:
                    raise ValueError(f"lengths must hold {B} frame counts in [1, {F}]")
                # through page-locked memory: a pageable source makes the copy block the host until it is done
                # (the header promises a forward that never synchronises)
                lens_dev = self._stage_lengths(lens_host, B, dev)
        need = self.workspace_bytes(B, F)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        if out is None:
            out = torch.empty((B, cfg.out_channels, T), dtype=torch.float32, device=dev)
        elif out.dtype != torch.float32 or tuple(out.shape) != (B, cfg.out_channels, T) or out.device != dev or \
                not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 {(B, cfg.out_channels, T)} tensor on {dev}")
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            common = (
                self._h, ctypes.c_void_p(blob.data_ptr()),
                ctypes.c_void_p(ppg.data_ptr()), ctypes.c_void_p(sine.data_ptr()),
                ctypes.c_void_p(lft.data_ptr()),
                ctypes.c_void_p(spk_emb.data_ptr()) if spk_emb is not None else None,
                ctypes.c_void_p(out.data_ptr()), B, F,
                ctypes.c_void_p(lens_dev.data_ptr()) if lens_dev is not None else None,
                ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), ctypes.c_void_p(stream))
            if autotune:
                ntr = ctypes.c_int32(0)
                rc = self.lib.fastsvc_autotune(*common[:9], common[10], common[11], common[12], ctypes.byref(ntr))
                self.last_autotune_trials = int(ntr.value)
            elif profile is None:
                rc = self.lib.fastsvc_forward(*common)
            else:
                recs = (_LaunchRecord * 256)()
                n = ctypes.c_int32(0)
                rc = self.lib.fastsvc_forward_profile(*common, recs, 256, ctypes.byref(n))
                if rc == 0:
                    for i in range(n.value):
                        profile.append(dict(layer=recs[i].layer.decode(), kernel=recs[i].kernel.decode(),
                                            flops=recs[i].flops, bytes=recs[i].bytes, ms=recs[i].ms))
        _check(self.lib, rc, "fastsvc_forward")
        self._last_workspace = (workspace, lens_dev)      # keep alive until the stream has consumed them
        return out
