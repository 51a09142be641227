"""The schema of ``LiveInfer.trace`` events — ONE definition that the package's LiveInfer, the restated reference control
flow (tests/test_reference_liveinfer_flow.py::ReferenceFlow) and the long-trace follower (tests/test_gpu_long.py::Follower)
all import, so a field added on one side cannot silently leave the others behind (round 3's red GPU gate was exactly that).

Events are plain tuples (named: they compare equal to bare tuples of the same content and index the same way).

  frame event     one per sampler decision of ``_call_for_streaming`` (demo/inference.py:75-81)
  response event  one per ``_call_for_response`` (demo/inference.py:40-52)
"""
from typing import NamedTuple, Optional

FRAME, RESPONSE = "frame", "response"


class FrameEvent(NamedTuple):
    kind: str            # FRAME
    video_time: float    # time of the frame the step consumed
    token: int           # the token the session went on with (the sampler's, or the schedule's under a forced schedule)
    kv_len: int          # KV length after the step
    sampled: int         # the token the sampler chose (== token unless a schedule overrode it)


class ResponseEvent(NamedTuple):
    kind: str            # RESPONSE
    video_time: float
    query: Optional[str]  # the user query answered, None for a self-triggered response
    output_ids: list      # greedy tokens, EOS included


FRAME_FIELDS = FrameEvent._fields
RESPONSE_FIELDS = ResponseEvent._fields


def frame_event(video_time, token, kv_len, sampled=None) -> FrameEvent:
    """``sampled`` defaults to ``token``: a flow without a schedule (the reference's) uses what it sampled."""
    return FrameEvent(FRAME, video_time, int(token), int(kv_len), int(token if sampled is None else sampled))


def response_event(video_time, query, output_ids) -> ResponseEvent:
    return ResponseEvent(RESPONSE, video_time, query, list(output_ids))


# ---- per-stage trace ranges (the reference's only instrument is the wall clock of demo/cli.py:31-38) ---------------------------
# With VLO_ROCTX=1 LiveInfer brackets its stages — "encode" (ViT + connector launches), "step" (the Llama frame step), "sample" (streaming sampler
# + the host's read of its token), "respond" (greedy response) — with roctx ranges, so a `rocprofv3 --marker-trace --kernel-trace` run cuts the
# kernel timeline by stage without kernel-name heuristics.  Off (the default) the context manager is a no-op object: nothing is loaded.
class _NoRange:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_RANGE = _NoRange()
_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        import ctypes
        import os
        _roctx = False
        if os.environ.get("VLO_ROCTX") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _roctx = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _roctx.roctxRangePushA(self.name)
        return self

    def __exit__(self, *exc):
        _roctx.roctxRangePop()
        return False


def stage(name: str):
    """`with stage("encode"): ...` — a roctx range when VLO_ROCTX=1 and the library loads, otherwise nothing."""
    return _Range(name.encode()) if _load_roctx() else _NO_RANGE
