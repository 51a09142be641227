"""The "two edits" claim of INTEGRATION.md, executed: the reference's OWN LiveInfer control flow (demo/inference.py:40-123 —
tensor-typed `last_ids`, `torch.cat` step inputs, `outputs.logits[:, -1:].softmax`, the three queue rules) restated here only for
`device` and the tokenizer-derived id tensors (no tokenizer files / torchvision exist offline; `assert last_ids == 933` is
Llama-3-tokenizer specific and dropped, as SURVEY.md §8c explains), run over the package's `LiveModel` +
`fast_greedy_generate` duck-typed surface.  It must produce the same events as the package's own `LiveInfer` (encode stream,
fused sampler, staging buffer) on the same engine.  Test infrastructure, not product code."""
import collections

import pytest
import torch

from oracle import vlo_oracle as O
from test_gpu_liveinfer import _build
from videollm_online_amd.trace import frame_event, response_event      # the ONE event schema (also LiveInfer's and Follower's)

pytestmark = pytest.mark.gpu


class ReferenceFlow:
    """demo/inference.py::LiveInfer with `self.model` = anything that quacks like LiveLlamaForCausalLM."""

    def __init__(self, model, generate, toks, frame_fps, device, max_new):
        self.model, self.generate, self.dev = model, generate, device
        self.hidden_size = model.config.hidden_size                                  # :19-26
        self.frame_fps = frame_fps
        self.frame_num_tokens = model.config.frame_num_tokens
        self.frame_token_interval_id = model.config.frame_token_interval_id
        self.inplace_output_ids = torch.zeros(1, max_new, device=device, dtype=torch.long)   # :30
        self.frame_token_interval_threshold = 0.725                                   # :31
        self.eos_token_id = model.config.eos_token_id
        t = lambda ids: torch.tensor([ids], device=device, dtype=torch.long)
        self._start_ids, self._added_stream_prompt_ids = t(toks.start_ids), t(toks.stream_prompt_ids)   # :33-35
        self._added_stream_generation_ids = t(toks.stream_generation_ids)
        self._query_ids = {q: t(ids) for q, ids in toks.query_ids.items()}
        self.events = []
        self.reset()

    def reset(self):                                                                  # :84-91
        self.query_queue, self.frame_embeds_queue = collections.deque(), collections.deque()
        self.video_time, self.last_frame_idx, self.video_tensor = 0, -1, None
        self.last_ids = torch.tensor([[]], device=self.dev, dtype=torch.long)
        self.past_key_values = None

    def load_video(self, frames):                                                     # :111-115 (read_video replaced by a tensor)
        self.video_tensor = frames.to(self.dev)

    def input_query_stream(self, query, video_time=None):                             # :93-97
        self.query_queue.append((self.video_time if video_time is None else video_time, query))

    def input_video_stream(self, video_time):                                         # :102-109
        frame_idx = int(video_time * self.frame_fps)
        if frame_idx > self.last_frame_idx:
            ranger = range(self.last_frame_idx + 1, frame_idx + 1)
            embeds = self.model.visual_embed(self.video_tensor[ranger]).split(self.frame_num_tokens)
            self.frame_embeds_queue.extend((r / self.frame_fps, e) for r, e in zip(ranger, embeds))
        self.last_frame_idx, self.video_time = frame_idx, video_time

    def _call_for_response(self, video_time, query):                                  # :40-52
        self.last_ids = self._query_ids[query] if query is not None else self._added_stream_generation_ids
        inputs_embeds = self.model.get_input_embeddings()(self.last_ids)
        output_ids, self.past_key_values = self.generate(model=self.model, inputs_embeds=inputs_embeds,
                                                         past_key_values=self.past_key_values, eos_token_id=self.eos_token_id,
                                                         inplace_output_ids=self.inplace_output_ids)
        self.last_ids = output_ids[:, -1:]
        self.events.append(response_event(video_time, query, output_ids[0].tolist()))

    def _call_for_streaming(self):                                                    # :54-82
        while self.frame_embeds_queue:
            if self.query_queue and self.frame_embeds_queue[0][0] > self.query_queue[0][0]:       # rule 1
                return self.query_queue.popleft()
            video_time, frame_embeds = self.frame_embeds_queue.popleft()
            if not self.past_key_values:
                self.last_ids = self._start_ids
            elif self.last_ids.numel() == 1 and int(self.last_ids) == self.eos_token_id:          # `self.last_ids == self.eos_token_id`
                self.last_ids = torch.cat([self.last_ids, self._added_stream_prompt_ids], dim=1)
            inputs_embeds = torch.cat([self.model.get_input_embeddings()(self.last_ids).view(1, -1, self.hidden_size),
                                       frame_embeds.view(1, -1, self.hidden_size)], dim=1)
            outputs = self.model(inputs_embeds=inputs_embeds, use_cache=True, past_key_values=self.past_key_values)
            self.past_key_values = outputs.past_key_values
            if self.query_queue and video_time >= self.query_queue[0][0]:                         # rule 2
                return self.query_queue.popleft()
            next_score = outputs.logits[:, -1:].softmax(dim=-1)                                   # rule 3
            if next_score[:, :, self.frame_token_interval_id] < self.frame_token_interval_threshold:
                next_score[:, :, self.frame_token_interval_id].zero_()
            self.last_ids = next_score.argmax(dim=-1)
            self.events.append(frame_event(video_time, int(self.last_ids), len(self.past_key_values)))   # no schedule: sampled == token
            if int(self.last_ids) != self.frame_token_interval_id:
                return video_time, None
        return None, None

    def __call__(self):                                                               # :117-123
        video_time, query = self._call_for_streaming()
        if video_time is not None:
            self._call_for_response(video_time, query)


@pytest.mark.parametrize("query_at", [0.0, 1.2, None])
def test_reference_control_flow_on_livemodel_equals_package_liveinfer(query_at):
    from videollm_online_amd.modeling_live import fast_greedy_generate
    spec, vspec = O.LLM_SPECS["toy128"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    frames = O.synthetic_frames(8, vspec.image_size, seed=1234).cuda()
    eng, li = _build(spec, vspec, w, vw, toks, prefetch=True, max_new_tokens=5)
    rf = ReferenceFlow(li.model, fast_greedy_generate, toks, 2, li.model.device, 5)
    q = "Please narrate the video in real time."
    for drv in (li, rf):
        drv.load_video(frames)
        if query_at is not None:
            drv.input_query_stream(q, video_time=query_at)
        for i in range(8):
            drv.input_video_stream(i / 2)
            drv()
    got, ref = list(li.trace), rf.events
    assert len(ref) >= 8 and any(e[0] == "response" for e in ref) == any(e[0] == "response" for e in got)
    assert got == ref, f"package LiveInfer and the reference control flow over LiveModel disagree:\n{got}\n{ref}"
    assert len(li.past_key_values) == len(rf.past_key_values)
    li.reset()
    rf.past_key_values.close()
    eng.close()
