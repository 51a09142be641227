"""The trace-event schema (videollm-online_amd/trace.py) is ONE definition shared by the package's LiveInfer, the restated
reference flow (ReferenceFlow) and the long-trace Follower.  Round 3's GPU gate went red because LiveInfer grew a fifth
field and ReferenceFlow kept emitting 4-tuples; these CPU tests pin the arity and run ReferenceFlow itself (over an
oracle-backed duck-typed model, no GPU) so that mismatch can no longer wait for a GPU box to be seen."""
import inspect
import types

import pytest
import torch

from oracle import vlo_oracle as O
from videollm_online_amd import trace as T


def test_schema_arity_and_fields():
    assert T.FRAME_FIELDS == ("kind", "video_time", "token", "kv_len", "sampled")
    assert T.RESPONSE_FIELDS == ("kind", "video_time", "query", "output_ids")
    f = T.frame_event(0.5, 7, 40)
    assert f == ("frame", 0.5, 7, 40, 7) and len(f) == len(T.FRAME_FIELDS)          # sampled defaults to the token used
    assert T.frame_event(0.5, 7, 40, sampled=9).sampled == 9
    r = T.response_event(1.0, None, (1, 2, 3))
    assert r == ("response", 1.0, None, [1, 2, 3]) and len(r) == len(T.RESPONSE_FIELDS)
    with pytest.raises(TypeError):
        T.FrameEvent(*("frame", 0.5, 7, 40))                                        # a 4-tuple from an out-of-date producer fails loudly


def test_every_producer_and_consumer_goes_through_the_schema():
    """No literal ("frame", ...) / ("response", ...) tuples outside trace.py: LiveInfer, ReferenceFlow, Follower import it."""
    import videollm_online_amd.inference as inf
    import tests.test_reference_liveinfer_flow as rflow
    import tests.test_gpu_long as tlong
    assert inf.frame_event is T.frame_event and inf.response_event is T.response_event
    assert rflow.frame_event is T.frame_event and rflow.response_event is T.response_event
    assert tlong.FrameEvent is T.FrameEvent and tlong.ResponseEvent is T.ResponseEvent
    for mod in (inf, rflow, tlong):
        src = inspect.getsource(mod)
        assert '(("frame"' not in src and '(("response"' not in src, f"{mod.__name__} builds a trace event by hand"


class _OracleBackedModel:
    """The duck-typed surface demo/inference.py uses on `self.model`, backed by the CPU oracle (bf16)."""

    def __init__(self, llm, vit_W, vspec, toks):
        self.llm, self.vit_W, self.vspec = llm, vit_W, vspec
        self.device = torch.device("cpu")
        self.config = types.SimpleNamespace(hidden_size=llm.spec.hidden_size, frame_num_tokens=vspec.frame_num_tokens,
                                            frame_token_interval_id=toks.interval_id, eos_token_id=toks.eos_token_id)

    def visual_embed(self, frames):
        return self.llm.visual_embed(self.vit_W, self.vspec, frames, None)

    def get_input_embeddings(self):
        return lambda ids: self.llm.embed(ids.view(-1)).view(1, -1, self.llm.spec.hidden_size)

    def __call__(self, inputs_embeds, use_cache=True, past_key_values=None):
        logits, cache = self.llm.forward(inputs_embeds[0], past_key_values)
        return types.SimpleNamespace(logits=logits[None], past_key_values=cache)


def _generate(model, inputs_embeds, past_key_values, eos_token_id, inplace_output_ids):
    out, cache = O.fast_greedy_generate(model.llm, inputs_embeds[0], past_key_values, eos_token_id, inplace_output_ids.shape[1])
    return torch.tensor([out]), cache


@pytest.mark.parametrize("query_at", [0.0, 1.2, None])
def test_reference_flow_emits_schema_events_and_matches_the_oracle_driver(query_at):
    """ReferenceFlow (tensor-typed last_ids, torch.cat, logits[:, -1:].softmax — the reference's own statements) over the oracle
    model == the oracle's LiveInferOracle, event by event, in the shared schema."""
    from tests.test_reference_liveinfer_flow import ReferenceFlow
    spec, vspec = O.LLM_SPECS["toy"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    frames = O.synthetic_frames(6, vspec.image_size, seed=1234)
    llm = O.LlamaOracle(spec, w, torch.bfloat16)
    rf = ReferenceFlow(_OracleBackedModel(llm, vw, vspec, toks), _generate, toks, 2, torch.device("cpu"), 5)
    o = O.LiveInferOracle(llm, vw, vspec, toks, frame_fps=2, max_new=5)
    q = "Please narrate the video in real time."
    for drv in (rf, o):
        drv.load_video(frames)
        if query_at is not None:
            drv.input_query_stream(q, video_time=query_at)
        for i in range(6):
            drv.input_video_stream(i / 2)
            drv()
    assert len(rf.events) == len(o.trace) >= 6
    for ev, ref in zip(rf.events, o.trace):
        if ref[0] == "frame":
            assert isinstance(ev, T.FrameEvent) and len(ev) == len(T.FRAME_FIELDS)
            assert ev == T.frame_event(ref[1], ref[2], ref[4]), (ev, ref)          # oracle: (kind, time, tok, p_int, kv_len, margin)
        else:
            assert isinstance(ev, T.ResponseEvent) and len(ev) == len(T.RESPONSE_FIELDS)
            assert ev == T.response_event(ref[1], ref[2], ref[3]), (ev, ref)
