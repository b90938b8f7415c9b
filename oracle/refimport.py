"""Import the live reference (``/root/reference``, package ``harana``) in the build container.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` (golden-vector generation)
and by the oracle-vs-live-reference checks that are skipped when ``/root/reference`` is absent
(it never exists on the GPU box).  Nothing under ``svcc23_fastsvc_amd/`` imports this file.

The generator lives in ``harana/models/fastsvc.py:235-383``; importing ``harana.models`` pulls in
four off-path third-party modules that are not installed here (``h5py``, ``librosa`` via
``harana/utils/utils.py:25,29``; ``tkinter.W`` via ``harana/models/hnusfgan.py:17``;
``torchaudio.functional.spectrogram`` via ``hnusfgan.py:28``; ``kaldiio`` via
``harana/datasets/scp_dataset.py:12``).  None of them is touched by the generator, so they are
replaced by empty placeholder modules.  We never write into ``/root/reference`` (no bytecode).
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FASTSVC_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "harana", "models", "fastsvc.py"))


def _placeholder(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    mod.__path__ = []
    sys.modules[name] = mod
    return mod


def import_reference():
    """Return the live ``harana.models`` module of the reference (container only)."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # root could write into the read-only mount: don't
    for name in ("h5py", "librosa", "tkinter", "torchaudio", "torchaudio.functional", "kaldiio"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _placeholder(name)
    if not hasattr(sys.modules["tkinter"], "W"):
        sys.modules["tkinter"].W = "w"
    if not hasattr(sys.modules["torchaudio.functional"], "spectrogram"):
        sys.modules["torchaudio.functional"].spectrogram = None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # drop any stand-in `harana` namespace (svcc23_fastsvc_amd.install_into_harana creates one
    # when the real package is not importable) so that the real package is the one imported
    if "harana" in sys.modules and not getattr(sys.modules["harana"], "__file__", None):
        for name in [m for m in sys.modules if m == "harana" or m.startswith("harana.")]:
            del sys.modules[name]
    import harana.models as ref_models  # noqa: E402

    return ref_models


def import_reference_signal_generator():
    """Return the reference ``SignalGenerator`` class (harana/utils/features.py:111-213)."""
    import_reference()
    from harana.utils.features import SignalGenerator  # noqa: E402

    return SignalGenerator
