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
