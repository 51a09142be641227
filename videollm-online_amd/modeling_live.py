"""Host-side mirror of the reference's model surface (models/modeling_live.py, models/live_llama/)
on top of the HIP engine, so ``demo/inference.py``'s LiveInfer logic runs unchanged on it:

    model.config.{hidden_size, frame_resolution, frame_num_tokens, v_placeholder, frame_token_interval_id,
                  v_placeholder_id, eos_token_id}                        demo/inference.py:19-32
    model.get_input_embeddings()(LongTensor[1,k]) -> [1,k,H]            :46,66
    model.visual_embed(uint8[B,3,R,R]) -> [B*T,H]                       :106  (models/modeling_live.py:21-27)
    model(inputs_embeds=[1,n,H], use_cache=True, past_key_values=h) -> .logits (indexable [:, -1:]), .past_key_values   :69-76
    fast_greedy_generate(model=, inputs_embeds=, past_key_values=, eos_token_id=, inplace_output_ids=)  models/modeling_live.py:173-182
"""
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from .engine import Engine, EngineConfig, Session


@dataclass
class LiveOutput:
    logits: torch.Tensor            # [1, 1, V]: only the last row is ever read (demo/inference.py:76, modeling_live.py:177)
    past_key_values: Session


class _Embedding:
    def __init__(self, engine: Engine):
        self.engine = engine

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        shape = tuple(ids.shape)
        return self.engine.embed(ids).view(*shape, self.engine.cfg.hidden_size)


class LiveModel:
    """Quacks like LiveLlamaForCausalLM for the streaming-inference path."""

    def __init__(self, engine: Engine, *, eos_token_id: int, frame_token_interval_id: int, frame_resolution: int = 384,
                 v_placeholder: str = "<v>", v_placeholder_id: int | None = None):
        self.engine = engine
        c = engine.cfg
        self.config = SimpleNamespace(hidden_size=c.hidden_size, frame_resolution=frame_resolution,
                                      frame_num_tokens=c.frame_num_tokens, v_placeholder=v_placeholder,
                                      frame_token_interval_id=frame_token_interval_id,
                                      v_placeholder_id=c.vocab_size if v_placeholder_id is None else v_placeholder_id,
                                      eos_token_id=eos_token_id, vocab_size=c.vocab_size)
        self.dtype = torch.bfloat16
        self.device = engine.device
        self._embedding = _Embedding(engine)

    def to(self, *_a, **_k):          # model.to('cuda') (demo/inference.py:16): weights already live in HBM
        return self

    def get_input_embeddings(self):
        return self._embedding

    def visual_embed(self, frames: torch.Tensor) -> torch.Tensor:
        return self.engine.visual_embed(frames.to(self.device))

    def new_cache(self) -> Session:
        return self.engine.new_session()

    def __call__(self, *, inputs_embeds: torch.Tensor, past_key_values: Session | None = None, use_cache: bool = True, **_):
        if inputs_embeds.dim() == 3:
            assert inputs_embeds.shape[0] == 1, "streaming inference is batch 1 (models/modeling_live.py:55)"
        sess = past_key_values if past_key_values is not None else self.new_cache()
        last, _ = self.engine.llm_step(sess, inputs_embeds)
        return LiveOutput(logits=last.view(1, 1, -1), past_key_values=sess)


def fast_greedy_generate(*, model: LiveModel, inputs_embeds: torch.Tensor, past_key_values: Session | None, eos_token_id: int,
                         inplace_output_ids: torch.Tensor, force_len: int = 0):
    """models/modeling_live.py:173-182 — tokens are written into ``inplace_output_ids`` in place; stops after
    writing EOS.  Returns (inplace_output_ids[:, :i+1], past_key_values)."""
    sess = past_key_values if past_key_values is not None else model.new_cache()
    n = model.engine.greedy_generate(sess, inputs_embeds, eos_token_id, inplace_output_ids.view(-1), force_len=force_len)
    return inplace_output_ids[:, :n], sess


def build_engine_config(llm: dict, vit: dict | None = None, **kw) -> EngineConfig:
    return EngineConfig(hidden_size=llm["hidden_size"], intermediate_size=llm["intermediate_size"],
                        num_hidden_layers=llm["num_hidden_layers"], num_attention_heads=llm["num_attention_heads"],
                        num_key_value_heads=llm["num_key_value_heads"], vocab_size=llm["vocab_size"],
                        rope_theta=llm.get("rope_theta", 10000.0), rms_norm_eps=llm.get("rms_norm_eps", 1e-5),
                        vision_hidden_size=(vit or {}).get("hidden_size", llm.get("vision_hidden_size", 1024)), vit=vit, **kw)
