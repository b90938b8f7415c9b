"""MI355X-native FastSVC generator forward (gfx950 HIP kernels behind the reference's
``harana.models.FastSVCGenerator`` surface).  See DESIGN.md / INTEGRATION.md."""
from .synth import GeneratorConfig, FULL_CONFIG, TINY_CONFIG  # noqa: F401
from .engine import FastSVCError, Plan, load_library, library_path, gather_padded  # noqa: F401
from .generator import FastSVCGenerator, install_into_harana  # noqa: F401
from .signal import SignalGenerator  # noqa: F401
from .loudness import loudness_extract  # noqa: F401
from .stft_loss import MultiResolutionSTFTLoss  # noqa: F401

__all__ = ["FastSVCGenerator", "SignalGenerator", "loudness_extract", "MultiResolutionSTFTLoss", "GeneratorConfig", "Plan", "FastSVCError", "install_into_harana",
           "load_library", "library_path", "gather_padded", "FULL_CONFIG", "TINY_CONFIG"]
