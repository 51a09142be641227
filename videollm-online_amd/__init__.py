"""videollm-online_amd — MI355X-native (gfx950) engine for the videollm-online streaming hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI of
include/vlo.h, built into libvlo.so) and the host-side mirror of the reference's
``LiveInfer`` / ``modeling_live.py`` surface.  Import as ``videollm_online_amd``."""
from .build import build  # noqa: F401

__all__ = ["build"]
