"""Host-side mirror of the reference's model surface (models/modeling_live.py, models/live_llama/)
on top of the HIP engine, so ``demo/inference.py``'s LiveInfer logic runs unchanged on it:

    model.config.{hidden_size, frame_resolution, frame_num_tokens, v_placeholder, frame_token_interval_id,
                  v_placeholder_id, eos_token_id}                        demo/inference.py:19-32
    model.get_input_embeddings()(LongTensor[1,k]) -> [1,k,H]            :46,66
    model.visual_embed(uint8[B,3,R,R]) -> [B*T,H]                       :106  (models/modeling_live.py:21-27)
    model(inputs_embeds=[1,n,H], use_cache=True, past_key_values=h) -> .logits (indexable [:, -1:]), .past_key_values   :69-76
    fast_greedy_generate(model=, inputs_embeds=, past_key_values=, eos_token_id=, inplace_output_ids=)  models/modeling_live.py:173-182

and the teacher-forced evaluation surface:

    model.joint_embed(input_ids, frames)                                models/modeling_live.py:29-42
    model(input_ids=[1,n], frames=..., past_key_values=h) -> .logits [1,n,V]   models/live_llama/modeling_live_llama.py:24-43
    model.trim_past_key_values(past, 0, stop)                           models/modeling_live.py:170-171
    model.stream_evaluate(input_ids, labels, frames, ...) -> [lm_ppl, frame_diff, fluency, lm_correctness]   :44-168
"""
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
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
        """models/modeling_live.py:21-27: uint8 frames go through the vision tower + connector; floating-point input is
        taken as pre-extracted features [T, frame_num_tokens, vision_hidden] (the branch without ``vision_encode``, the
        form data/stream.py:91 feeds to training / evaluation) and goes through the connector only."""
        if frames.dtype == torch.uint8:
            return self.engine.visual_embed(frames.to(self.device))
        return self.engine.connector(frames.to(self.device))

    def new_cache(self) -> Session:
        return self.engine.new_session()

    def joint_embed(self, input_ids: torch.Tensor | None = None, frames: torch.Tensor | None = None) -> torch.Tensor:
        """models/modeling_live.py:29-42."""
        if frames is None:
            return self.get_input_embeddings()(input_ids)
        if input_ids is None:
            return self.visual_embed(frames)
        rows = self.visual_embed(frames) if frames.numel() else None
        x = self.engine.joint_embed(input_ids, rows, self.config.v_placeholder_id)
        return x.view(*input_ids.shape, self.config.hidden_size)

    def __call__(self, *, input_ids: torch.Tensor | None = None, frames: torch.Tensor | None = None,
                 inputs_embeds: torch.Tensor | None = None, past_key_values: Session | None = None, use_cache: bool = True,
                 all_logits: bool | None = None, **_):
        """LiveLlamaForCausalLM.forward (models/live_llama/modeling_live_llama.py:24-43), inference part.  With
        ``inputs_embeds`` (the LiveInfer path) only the last row of the logits is produced — the only row that path reads;
        with ``input_ids`` (teacher-forced path) or ``all_logits=True`` the full [1, n, V] matrix is returned."""
        if inputs_embeds is None:
            inputs_embeds = self.joint_embed(input_ids, frames)
            if all_logits is None:
                all_logits = True
        if inputs_embeds.dim() == 3:
            assert inputs_embeds.shape[0] == 1, "inference is batch 1 (models/modeling_live.py:55)"
        sess = past_key_values if past_key_values is not None else self.new_cache()
        if all_logits:
            _, full = self.engine.llm_step(sess, inputs_embeds, want_last=False, want_all=True)
            return LiveOutput(logits=full.view(1, -1, full.shape[-1]), past_key_values=sess)
        last, _ = self.engine.llm_step(sess, inputs_embeds)
        return LiveOutput(logits=last.view(1, 1, -1), past_key_values=sess)

    forward = __call__

    def trim_past_key_values(self, past_key_values: Session, start: int, stop: int) -> Session:
        """models/modeling_live.py:170-171.  The reference only ever trims from 0 (:118); the result is a new cache and the
        source keeps its length (slices + DynamicCache's concatenating update never write into it)."""
        if start != 0:
            raise NotImplementedError("only prefixes can be kept (start must be 0, as in stream_evaluate)")
        return past_key_values.fork(int(stop))

    # ---- stream_evaluate ------------------------------------------------------------------------------------------
    def _row_stats(self, embeds: torch.Tensor, sess: Session, labels: torch.Tensor | None, sil: int, slab: int = 2048):
        """Forward ``embeds`` [n, H] on ``sess`` in slabs and reduce every logits row to the five numbers the metrics
        need (include/vlo.h::vlo_logit_rows); the [n, V] logits never exist as a whole.  Returns host numpy arrays."""
        n = embeds.shape[0]
        parts = []
        for a in range(0, n, slab):
            b = min(n, a + slab)
            _, lg = self.engine.llm_step(sess, embeds[a:b], want_last=False, want_all=True)
            parts.append(self.engine.logit_rows(lg, None if labels is None else labels[a:b], sil))
        return {k: torch.cat([p[k] for p in parts]).cpu().numpy() for k in parts[0]}

    @torch.no_grad()
    def stream_evaluate(self, input_ids: torch.LongTensor, labels: torch.LongTensor, frames: torch.Tensor,
                        ignore_token_id: int = -100, frame_token_interval_threshold: float = 0.0, **kwargs) -> torch.Tensor:
        """LiveMixin.stream_evaluate (models/modeling_live.py:44-168): [lm_ppl, frame_diff, fluency, lm_correctness].

        One teacher-forced pass over the whole dialogue, every logits row reduced on the GPU to (log-sum-exp, argmax,
        logit at the label, p(interval), argmax of the bf16 softmax); the per-turn bookkeeping below runs on those small
        host arrays.  When a turn's learnt frames all stay silent, the KV prefix up to the last of them is forked and
        continued with the next turn's frames to measure how late the reply would come (:116-148)."""
        assert input_ids.size(0) == labels.size(0) == 1, "evaluation is batch 1 (:55)"
        cfg = self.config
        ids_d, lab_d = input_ids[0].to(self.device), labels[0].to(self.device)
        ids, lab = ids_d.cpu().numpy(), lab_d.cpu().numpy()
        v_id, eos, fnt = cfg.v_placeholder_id, cfg.eos_token_id, cfg.frame_num_tokens
        use_interval = cfg.frame_token_interval_id is not None
        sil = cfg.frame_token_interval_id if use_interval else eos                      # :72-73
        thr = float(frame_token_interval_threshold)
        thr_b = float(torch.tensor(thr, dtype=self.dtype)) if thr > 0 else 0.0         # torch compares bf16 scores in bf16

        frame_rows = self.visual_embed(frames) if frames is not None and frames.numel() else None
        sess = self.new_cache()
        st = self._row_stats(self.engine.joint_embed(ids_d, frame_rows, v_id), sess, lab_d, sil)

        def speaks(rows):
            # :107-112 / :140-144 — a row below the threshold is zeroed as a whole, so its argmax is id 0
            tok = rows["p_argmax"].copy()
            if thr > 0:
                tok[rows["p_interval"] < thr_b] = 0
            return tok != sil

        stops = (np.nonzero(ids == eos)[0] + 1).tolist()                                # :62-63
        starts = [0] + stops[:-1]
        spoke = speaks(st)
        ce = st["lse"] - st["label_logit"]
        ppls, diffs, fluencies, corrects = [], [], [], []
        frames_seen = 0
        for r, (a, b) in enumerate(zip(starts, stops)):
            learnt = lab[a:b] != ignore_token_id
            if not learnt.any():
                continue                                                               # (:83-84: frame counter not advanced)
            at_v = ids[a:b] == v_id
            n_frames = int(at_v.sum()) // fnt
            on_stream = at_v & learnt
            on_text = learnt & ~on_stream
            n_ok = diff = None
            if on_text.any():                                                          # :93-102
                ppls.append(float(np.exp(ce[a:b][on_text].astype(np.float64).mean())))
                wrong = st["argmax"][a:b][on_text] != lab[a:b][on_text]
                n_ok = int(np.argmax(wrong)) if wrong.any() else int(wrong.size)
                corrects.append(n_ok / int(on_text.sum()))
            if on_stream.any():                                                        # :105-149
                hit = spoke[a:b][on_stream]
                n_stream = int(on_stream.sum())
                if hit.any():
                    diff = n_stream - int(np.argmax(hit)) - 1
                elif r == len(starts) - 1:
                    diff = 0
                else:
                    nxt = int((ids[starts[r + 1]:stops[r + 1]] == v_id).sum()) // fnt
                    k = min(nxt, n_frames - 1)
                    if k == 0:
                        diff = 0
                    else:
                        keep = a + int(np.nonzero(on_stream)[0][-1]) + 1
                        f0 = frames_seen + n_frames
                        unit = ([sil] if use_interval else []) + [v_id] * fnt
                        more = torch.tensor(unit * k, dtype=torch.long, device=self.device)
                        prefix = self.trim_past_key_values(sess, 0, keep)
                        try:
                            x2 = self.engine.joint_embed(more, frame_rows[f0 * fnt:(f0 + k) * fnt], v_id)
                            late = speaks(self._row_stats(x2, prefix, None, sil))[len(unit) - 1::len(unit)]
                        finally:
                            prefix.close()
                        diff = -(int(np.argmax(late)) + 1) if late.any() else -k
                diffs.append(abs(diff))
            if on_text.any() and on_stream.any():                                      # :152-161
                n_v, denom = int(on_stream.sum()), int(on_text.sum()) + int(on_stream.sum())
                if diff == 0:
                    fluencies.append((n_v + n_ok) / denom)
                elif diff > 0:
                    fluencies.append((n_v - diff) / denom)
                else:
                    fluencies.append((n_v - 1) / denom)
            frames_seen += n_frames
        sess.close()
        mean = lambda xs, empty: float(np.mean(xs)) if xs else empty
        out = [mean(ppls, 1.0), mean(diffs, 0.0), mean(fluencies, 1.0), mean(corrects, 1.0)]
        return torch.tensor(out, dtype=torch.float32, device=self.device)


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
