"""Host logic of the streaming session (demo/inference.py:40-123) on the CPU oracle driver: the three queue rules,
last_ids transitions, scheduled mode bookkeeping.  The product LiveInfer (GPU) is checked against this driver
event by event in tests/test_gpu_liveinfer.py."""
import torch

from oracle import vlo_oracle as O


def _mk(schedule=None, max_new=4, threshold=0.725):
    spec, vspec = O.LLM_SPECS["toy"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=0), O.init_vit_weights(vspec, seed=1)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    li = O.LiveInferOracle(O.LlamaOracle(spec, w, torch.bfloat16), vw, vspec, toks, frame_fps=2, max_new=max_new,
                           schedule=schedule, threshold=threshold)
    li.load_video(O.synthetic_frames(8, vspec.image_size, seed=1234))
    return li, toks


def test_rule2_query_at_frame_time_is_answered_after_that_frame():
    li, toks = _mk(schedule=lambda i: (False, 3))
    li.input_query_stream("Please narrate the video in real time.", video_time=0.0)
    li.input_video_stream(0.0)
    q, r = li()
    assert q == "Please narrate the video in real time." and len(r) == 3 and r[-1] == toks.eos_token_id
    # cache = start prompt + 10 frame tokens + query prompt + (3 - 1) fed-back tokens
    assert len(li.past_key_values) == 19 + 10 + 12 + 2
    assert li.trace[0][0] == "response" and li.trace[0][2] is not None     # no sampler event for that frame


def test_rule1_query_before_next_frame_preempts_it():
    li, toks = _mk(schedule=lambda i: (False, 2))
    li.input_video_stream(0.0); li()                 # frame 0, silent
    li.input_query_stream("Please narrate the video in real time.", video_time=0.7)
    li.input_video_stream(1.0)                       # frames 1 (t=0.5) and 2 (t=1.0) queued
    q, r = li()
    assert q is not None                             # frame 0.5 processed, then the 0.7 s query answered before frame 1.0
    kinds = [(e[0], e[1]) for e in li.trace]
    assert kinds == [("frame", 0.0), ("frame", 0.5), ("response", 0.7)]
    assert len(li.frame_embeds_queue) == 1 and li.frame_embeds_queue[0][0] == 1.0
    q2, r2 = li()                                    # now the remaining frame
    assert q2 is None and r2 is None and li.trace[-1][:2] == ("frame", 1.0)


def test_rule3_non_interval_token_triggers_proactive_response_and_eos_transition():
    li, toks = _mk(schedule=lambda i: (i == 1, 2))   # speak after frame 1
    for t in (0.0, 0.5, 1.0):
        li.input_video_stream(t)
        li()
    ev = li.trace
    assert [e[0] for e in ev] == ["frame", "frame", "response", "frame"]
    assert ev[0][2] == toks.interval_id and ev[1][2] != toks.interval_id
    assert ev[2][2] is None and ev[2][3][-1] == toks.eos_token_id                      # proactive: no query
    # frame steps: 19+10 | 1+10 | response: 4 prompt + 1 fed back | after EOS: [eos]+2 prompt ids + 10
    assert len(li.past_key_values) == 29 + 11 + 5 + 13


def test_free_running_threshold_semantics():
    # threshold 0: the interval token is never zeroed; threshold > 1: always zeroed -> never silent
    li0, toks = _mk(threshold=0.0, max_new=2)
    li1, _ = _mk(threshold=1.1, max_new=2)
    for li in (li0, li1):
        li.input_video_stream(0.0)
        li()
    assert li1.trace[0][2] != toks.interval_id
    p_int = li0.trace[0][3]
    assert 0.0 <= p_int <= 1.0


def test_reset_clears_session():
    li, _ = _mk(schedule=lambda i: (False, 2))
    li.input_video_stream(0.0); li()
    assert li.past_key_values is not None
    li.reset()
    assert li.past_key_values is None and not li.frame_embeds_queue and li.last_frame_idx == -1
