"""prompt_tts_amd: MI355X-native hot path of khaidoan25/prompt-tts (denoiser training step + Encodec decode).

Importing this package loads the hand-written HIP library (prompt_tts_amd/libprompt_tts_hip.so) and fails
loudly if it is missing -- there is no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library has not been built)

__all__ = ["_lib"]
