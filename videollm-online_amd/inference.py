"""LiveInfer — the streaming session state machine of demo/inference.py:12-123 on the HIP engine.

Same public methods and the same three queue rules (:56-59, :71-74, :75-81); what changes is
where the device work runs:
  * frame encode (ViT + connector) goes to a dedicated HIP stream and, when ``prefetch`` is on,
    frame t+1 is encoded while the LLM step of frame t runs (README.md:25 promises this
    asynchrony; the reference's implementation is synchronous — SURVEY.md §2.3);
  * softmax / threshold / argmax run in one fused sampler kernel and the host reads ONE token per
    frame (the reference syncs twice, :77 and :80);
  * the KV handle is the engine's paged session, not a re-concatenated DynamicCache.
"""
import collections
from dataclasses import dataclass, field

import torch

from .modeling_live import LiveModel, fast_greedy_generate
from .trace import frame_event, response_event, stage


@dataclass
class StreamTokens:
    """Ids the reference derives from the tokenizer chat template (demo/inference.py:33-35,42;
    models/tokenization_live.py:27-65).  Pass a tokenizer to LiveInfer to derive them instead."""
    start_ids: list
    stream_prompt_ids: list
    stream_generation_ids: list
    eos_token_id: int
    interval_id: int
    query_ids: dict = field(default_factory=dict)


class LiveInfer:
    def __init__(self, model: LiveModel, tokens: StreamTokens | None = None, tokenizer=None, frame_fps: float = 2,
                 system_prompt: str = "", prefetch: bool = True, prefetch_frames: int = 2, schedule=None,
                 max_new_tokens: int = 100, record: int = 65536, encode_stream=None):
        self.model = model
        self.engine = model.engine
        self.tokenizer = tokenizer
        dev = model.device
        # visual (demo/inference.py:19-26)
        self.hidden_size = model.config.hidden_size
        self.frame_fps = frame_fps
        self.frame_interval = 1 / frame_fps
        self.frame_resolution = model.config.frame_resolution
        self.frame_num_tokens = model.config.frame_num_tokens
        self.frame_token_interval_id = model.config.frame_token_interval_id
        # generation (:29-35)
        self.system_prompt = system_prompt
        self.inplace_output_ids = torch.zeros(1, max_new_tokens, device=dev, dtype=torch.long)
        self.frame_token_interval_threshold = 0.725
        self.eos_token_id = model.config.eos_token_id
        if tokens is None:
            if tokenizer is None:
                raise ValueError("LiveInfer needs either a tokenizer or explicit StreamTokens")
            t = tokenizer
            ids = lambda *a, **k: _flat_ids(t.apply_chat_template(*a, **k))
            tokens = StreamTokens(
                start_ids=ids([{"role": "system", "content": system_prompt}], add_stream_prompt=True),
                stream_prompt_ids=ids([{}], add_stream_prompt=True),
                stream_generation_ids=ids([{}], add_stream_generation_prompt=True),
                eos_token_id=self.eos_token_id, interval_id=self.frame_token_interval_id)
        self.tokens = tokens
        self._start_ids = list(tokens.start_ids)
        self._added_stream_prompt_ids = list(tokens.stream_prompt_ids)
        self._added_stream_generation_ids = list(tokens.stream_generation_ids)
        # device plumbing
        self.prefetch = prefetch
        self.real_greedy = False                         # scheduled responses through the token-reading greedy loop instead of the forced-length one (bench.py)
        self.frame_wait_s = 10.0           # how long input_video_stream waits for a frame a FrameRing's feeder has not pushed yet
        self.prefetch_frames = max(1, prefetch_frames)   # frames encoded ahead in ONE batched ViT call (the reference
        # batches all pending frames the same way, demo/inference.py:105-106); the video is fully loaded up front
        self.schedule = schedule           # frame_idx -> None | (speak: bool, num_tokens: int)  (throughput runs)
        self._main = torch.cuda.current_stream(dev)
        # the encode stream: the caller's (e.g. the main stream itself: encodes then run BETWEEN Llama steps instead of beside
        # them), or a dedicated one
        self._enc = encode_stream if encode_stream is not None else torch.cuda.Stream(dev)
        self._tok_dev = torch.zeros(1, dtype=torch.long, device=dev)
        self._p_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._tok_host = torch.zeros(1, dtype=torch.long).pin_memory()
        # step input staging (demo/inference.py:61-68): text rows + this frame's rows, written by ONE launch (vlo_step_input);
        # sized for the longest text prefix a frame step can carry; longer ones (a long system prompt) grow it
        self._stage = torch.empty(64 + self.frame_num_tokens, self.hidden_size, dtype=torch.bfloat16, device=dev)
        # event log for tests / the bench: bounded (a long-running session must not grow host memory), cleared by reset()
        self._record = max(0, int(record))
        self.past_key_values = None
        self.reset()

    # ---- queries -> ids ----------------------------------------------------------------------
    def _query_ids(self, query):
        if query in self.tokens.query_ids:
            return list(self.tokens.query_ids[query])
        if self.tokenizer is None:
            raise KeyError(f"no ids for query {query!r} and no tokenizer")
        return _flat_ids(self.tokenizer.apply_chat_template([{"role": "user", "content": query}], add_stream_query_prompt=True,
                                                            add_generation_prompt=True))

    # ---- reference API -----------------------------------------------------------------------
    def reset(self):                                                   # :84-91
        self.query_queue = collections.deque()
        self.frame_embeds_queue = collections.deque()
        self.video_time = 0
        self.last_frame_idx = -1
        self.video_tensor = None
        self._ring = None
        self.last_ids = []
        if self.past_key_values is not None:
            self.past_key_values.close()
        self.past_key_values = None
        self._encoded = {}                 # frame idx -> (embeds [T,H], ready event)
        self._frames_done = 0
        self.trace = collections.deque(maxlen=self._record or 1)       # trace.FrameEvent / trace.ResponseEvent, newest last
        self.step_log = collections.deque(maxlen=self._record or 1)    # (cache length before, new tokens) of the Llama steps
        self.steps_total = 0               # Llama steps executed since reset() (step_log keeps the newest `record` of them)

    def _log_step(self, Lc, n):
        self.steps_total += 1
        if self._record:
            self.step_log.append((Lc, n))

    def drop_prefetched(self):
        """Forget frames encoded ahead and not yet queued: with a resident video they are simply encoded again when their time
        comes.  A FrameRing has already been told it may overwrite them (their encode has read them), so there the embeddings are
        kept — dropping them would leave frames that can no longer be encoded."""
        if self._ring is None:
            self._encoded.clear()

    def load_video(self, video):                                       # :111-115
        """``video``: uint8 tensor [T,3,R,R] (what read_video(..., output_format='TCHW') yields for the ffmpeg-prepared file),
        decoded frames of ANY size as uint8 [T,H,W,3] / [T,3,H,W] (prepared on the device by vlo_frame_ingest exactly as
        data/utils.py:51-66 prepares the file: longer side to R, bicubic, black padding), a ``FrameRing`` that is being
        filled while the stream runs (the video never has to be resident as a whole), or a path (needs torchvision, exactly
        like the reference)."""
        from .ingest import FrameRing
        self._ring = None
        if isinstance(video, FrameRing):
            self._ring, self.video_tensor = video, None
            self._video_ready = None
            return
        if isinstance(video, str):
            from torchvision.io import read_video
            video = read_video(video, pts_unit="sec", output_format="TCHW")[0]
        R = self.frame_resolution
        if video.dtype != torch.uint8 or video.dim() != 4 or 3 not in (video.shape[1], video.shape[3]):
            raise ValueError(f"video must be uint8 [T,3,H,W] or [T,H,W,3], got {video.dtype} {tuple(video.shape)}")
        video = video.to(self.model.device)
        if tuple(video.shape[1:]) != (3, R, R):
            video = self.engine.frame_ingest(video, resolution=R)     # the ffmpeg_once preparation, on the device
        self.video_tensor = video
        self._video_ready = torch.cuda.Event()
        self._video_ready.record(self._main)
        self.video_duration = self.video_tensor.size(0) / self.frame_fps

    @property
    def num_video_frames(self):
        return len(self._ring) if self._ring is not None else (0 if self.video_tensor is None else self.video_tensor.size(0))

    def input_query_stream(self, query, history=None, video_time=None):   # :93-100
        self.query_queue.append((self.video_time if video_time is None else video_time, query))
        if not self.past_key_values:
            return f'(NOTE: No video stream here. Please select or upload a video. Then the assistant will answer "{query} (at {self.video_time}s)" in the video stream)'
        return f'(NOTE: Received "{query}" (at {self.video_time}s). Please wait until previous frames have been processed)'

    def _encode_async(self, lo, hi):
        """Launch ViT+connector for frames [lo, hi) on the encode stream; one batched call like :106."""
        todo = [i for i in range(lo, hi) if i not in self._encoded and 0 <= i < self.num_video_frames]
        if not todo:
            return
        lo2, hi2 = todo[0], todo[-1] + 1
        if self._ring is not None:
            frames, ready = self._ring.window(lo2, hi2)
            frames.record_stream(self._enc)
        else:
            frames, ready = self.video_tensor[lo2:hi2], self._video_ready
        self._enc.wait_event(ready)                  # the video upload / ingest; NOT the main stream's Llama work
        with stage("encode"), torch.cuda.stream(self._enc):
            emb = self.model.engine.visual_embed(frames, stream=self._enc)
            emb.record_stream(self._main)
            ev = torch.cuda.Event()
            ev.record(self._enc)
        if self._ring is not None:
            self._ring.release(hi2, after=ev)        # the ring may overwrite these frames once this encode has read them
        for j, i in enumerate(range(lo2, hi2)):
            self._encoded[i] = (emb[j * self.frame_num_tokens:(j + 1) * self.frame_num_tokens], ev)

    def input_video_stream(self, video_time):                          # :102-109
        frame_idx = int(video_time * self.frame_fps)
        if frame_idx > self.last_frame_idx:
            ranger = range(self.last_frame_idx + 1, frame_idx + 1)
            if self._ring is not None and not self._ring.closed:
                if ranger.stop - self._ring.tail > self._ring.capacity:
                    raise RuntimeError(f"input_video_stream({video_time}) needs frames [{ranger.start}, {ranger.stop}) at once but the FrameRing holds "
                                       f"{self._ring.capacity}: advance the stream in smaller steps (or build a larger ring)")
                self._ring.wait_for(ranger.stop, self.frame_wait_s)   # a live feed: the frames of this instant may still be on their way
            self._encode_async(ranger.start, ranger.stop)
            for r in ranger:
                if r not in self._encoded:
                    # a decoder failure matters only when a frame this instant needs never arrived (a late error — e.g. at
                    # process teardown after every frame was pushed — is reported by the caller at the end, cli.py)
                    feed_error = getattr(getattr(self._ring, "feed", None), "error", None)
                    if feed_error is not None:
                        raise RuntimeError(f"the video decoder feeding the FrameRing failed before frame {r} arrived: {feed_error}") from feed_error
                    raise RuntimeError(f"frame {r} is not available: the video has {self.num_video_frames} frames"
                                       + (" so far (push it into the FrameRing first)" if self._ring is not None else ""))
                self.frame_embeds_queue.append((r / self.frame_fps, self._encoded.pop(r)))
        self.last_frame_idx = frame_idx
        self.video_time = video_time

    def _call_for_response(self, video_time, query):                   # :40-52
        if query is not None:
            self.last_ids = self._query_ids(query)
        else:
            # the reference asserts last_ids == 933 here (Llama-3-tokenizer specific, :44); any non-interval
            # token is the trigger (rule 3)
            self.last_ids = list(self._added_stream_generation_ids)
        inputs_embeds = self.model.get_input_embeddings()(torch.tensor([self.last_ids], device=self.model.device))
        forced = self.schedule(self._frames_done - 1) if self.schedule is not None else None
        if self.past_key_values is None:
            self.past_key_values = self.model.new_cache()
        L0 = len(self.past_key_values)
        if forced is not None and self.real_greedy:
            # a scheduled response through the loop a real stream takes (models/modeling_live.py:173-182: every token read on the host, here with the
            # next step enqueued speculatively): an EOS id the model cannot emit and a buffer of exactly the scheduled length — same tokens fed,
            # same KV, same number of steps as the forced form, which never looks at a token (bench.py times both)
            output_ids, self.past_key_values = fast_greedy_generate(
                model=self.model, inputs_embeds=inputs_embeds, past_key_values=self.past_key_values,
                eos_token_id=-1, inplace_output_ids=self.inplace_output_ids[:, :forced[1]], force_len=0)
        else:
            output_ids, self.past_key_values = fast_greedy_generate(
                model=self.model, inputs_embeds=inputs_embeds, past_key_values=self.past_key_values,
                eos_token_id=self.eos_token_id, inplace_output_ids=self.inplace_output_ids,
                force_len=forced[1] if forced is not None else 0)
        out = output_ids[0].tolist()
        if forced is not None and self.real_greedy:
            out[-1] = self.eos_token_id          # the forced form ends every scheduled response with EOS (never fed to the model): the same token flow
        self._log_step(L0, len(self.last_ids))
        for j in range(len(out) - 1):
            self._log_step(L0 + len(self.last_ids) + j, 1)
        self.last_ids = out[-1:]
        if self._record:
            self.trace.append(response_event(video_time, query, out))
        if query:
            query = f"(Video Time = {video_time}s) User: {query}"
        if self.tokenizer is not None:
            text = self.tokenizer.decode(output_ids[0], skip_special_tokens=True, clean_up_tokenization_spaces=True)
        else:
            text = " ".join(map(str, out))
        response = f"(Video Time = {video_time}s) Assistant:{text}"
        return query, response

    def _call_for_streaming(self):                                     # :54-82
        eng = self.engine
        while self.frame_embeds_queue:
            # 1. if query is before next frame, response
            if self.query_queue and self.frame_embeds_queue[0][0] > self.query_queue[0][0]:
                video_time, query = self.query_queue.popleft()
                return video_time, query
            video_time, (frame_embeds, ready) = self.frame_embeds_queue.popleft()
            if not self.past_key_values:
                self.last_ids = list(self._start_ids)
            elif self.last_ids == [self.eos_token_id]:
                self.last_ids = self.last_ids + self._added_stream_prompt_ids
            if self.past_key_values is None:
                self.past_key_values = self.model.new_cache()
            self._main.wait_event(ready)
            if len(self.last_ids) + self.frame_num_tokens > self._stage.shape[0]:
                self._stage = torch.empty(len(self.last_ids) + self.frame_num_tokens, self.hidden_size, dtype=torch.bfloat16,
                                          device=self.model.device)
            with stage("step"):
                inputs_embeds = eng.step_input(self.last_ids, frame_embeds, self._stage)      # :61-68 without torch.tensor / torch.cat
                self._log_step(len(self.past_key_values), inputs_embeds.shape[0])
                eng.llm_step(self.past_key_values, inputs_embeds, want_last=False)
            self._frames_done += 1
            nxt = self.last_frame_idx + 1
            if self.prefetch and not self.frame_embeds_queue and nxt not in self._encoded:
                # the next frame(s) are encoded on the encode stream while this Llama step runs; the encode is
                # enqueued here, AFTER the step's own launches, so the host never delays the step
                self._encode_async(nxt, nxt + self.prefetch_frames)
            # 2. if the same time, response after frame at that time
            if self.query_queue and video_time >= self.query_queue[0][0]:
                video_time, query = self.query_queue.popleft()
                return video_time, query
            # 3. if the next is frame but next is not interval, then response
            with stage("sample"):
                eng.stream_sample(self.past_key_values, self.frame_token_interval_threshold, self.frame_token_interval_id,
                                  tok_out=self._tok_dev, p_out=self._p_dev)
                self._tok_host.copy_(self._tok_dev, non_blocking=True)
                self._main.synchronize()
            tok = sampled = int(self._tok_host[0])
            forced = self.schedule(self._frames_done - 1) if self.schedule is not None else None
            if forced is not None:
                tok = self._added_stream_generation_ids[0] if forced[0] else self.frame_token_interval_id
            self.last_ids = [tok]
            if self._record:
                # trace.FrameEvent: the sampler's own choice differs from the token used only under a schedule
                self.trace.append(frame_event(video_time, tok, len(self.past_key_values), sampled))
            if tok != self.frame_token_interval_id:
                return video_time, None
        return None, None

    def __call__(self):                                                # :117-123
        if not self.frame_embeds_queue:
            raise RuntimeError("no frame queued: call input_video_stream first (the reference busy-waits here, :118)")
        video_time, query = self._call_for_streaming()
        response = None
        if video_time is not None:
            with stage("respond"):
                query, response = self._call_for_response(video_time, query)
        return query, response


def _flat_ids(x):
    """apply_chat_template returns a tensor (transformers 4.4x) or a BatchEncoding (5.x)."""
    if hasattr(x, "input_ids"):
        x = x.input_ids
    if hasattr(x, "tolist"):
        x = x.tolist()
    while x and isinstance(x[0], (list, tuple)):
        x = x[0]
    return list(x)
