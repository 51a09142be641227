"""Thin torch-facing wrapper of the C ABI (include/vlo.h).  torch is plumbing only: it owns
the caller-side device buffers and the HIP streams whose handles are passed down; every
computation happens in libvlo.so's HIP kernels."""
import ctypes as C
import weakref
from dataclasses import dataclass

import torch

from . import _C

_DT = {torch.float32: _C.DT_F32, torch.bfloat16: _C.DT_BF16, torch.float16: _C.DT_F16}


@dataclass
class EngineConfig:
    """Field names follow LlamaConfig / LiveConfigMixin (models/configuration_live.py:4-21)."""
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    rope_theta: float = 10000.0
    rms_norm_eps: float = 1e-5
    vision_hidden_size: int = 1024
    frame_num_tokens: int = 10
    frame_token_pooled: tuple = (3, 3)
    # vision tower (None = LLM-only engine)
    vit: dict | None = None
    kv_pool_tokens: int = 16384
    # tensor parallel shard held by this engine (see TpGroup)
    tp_rank: int = 0
    tp_size: int = 1
    # storage of the streamed Llama projections: "bf16", or "fp8" = e4m3 + one fp32 scale per output channel
    # (BASELINE.json configs[4]; include/vlo.h vlo_config.weight_dtype).  bf16 weights handed to an fp8 engine are quantised
    # on the way in (checkpoint.quantize_fp8_per_channel)
    weight_dtype: str = "bf16"
    # fp8 engines only: "fp8" = the long-input (prefill) projections quantise their X rows to e4m3 (one fp32 scale per row) and run on the native
    # fp8 MFMA (include/vlo.h vlo_config.prefill_act_dtype); "bf16" = expand the weight image per GEMM, bf16 MFMA.  The live step is bf16 either way.
    prefill_act_dtype: str = "bf16"

    def to_c(self) -> _C.VloConfig:
        c = _C.VloConfig()
        c.abi_version = _C.VLO_ABI_VERSION
        c.hidden_size, c.intermediate_size, c.num_layers = self.hidden_size, self.intermediate_size, self.num_hidden_layers
        c.num_heads, c.num_kv_heads, c.vocab_size = self.num_attention_heads, self.num_key_value_heads, self.vocab_size
        c.rope_theta, c.rms_eps = self.rope_theta, self.rms_norm_eps
        c.vision_hidden_size, c.frame_num_tokens = self.vision_hidden_size, self.frame_num_tokens
        c.pool_h, c.pool_w = self.frame_token_pooled
        c.kv_pool_tokens = self.kv_pool_tokens
        c.tp_rank, c.tp_size = self.tp_rank, self.tp_size
        if self.weight_dtype not in ("bf16", "fp8"):
            raise ValueError("weight_dtype must be 'bf16' or 'fp8'")
        c.weight_dtype = 1 if self.weight_dtype == "fp8" else 0
        if self.prefill_act_dtype not in ("bf16", "fp8") or (self.prefill_act_dtype == "fp8" and self.weight_dtype != "fp8"):
            raise ValueError("prefill_act_dtype must be 'bf16', or 'fp8' on an engine with weight_dtype='fp8'")
        c.prefill_act_dtype = 1 if self.prefill_act_dtype == "fp8" else 0
        if self.vit:
            v = self.vit
            c.has_vit = 1
            c.vit_hidden_size, c.vit_intermediate_size = v["hidden_size"], v["intermediate_size"]
            c.vit_num_layers, c.vit_num_heads = v["num_layers"], v["num_heads"]
            c.vit_image_size, c.vit_patch_size = v["image_size"], v["patch_size"]
            c.vit_ln_eps = v.get("ln_eps", 1e-6)
        return c


def _stream_handle(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


class Session:
    """KV handle: what the reference passes around as `past_key_values` (a DynamicCache).
    Truthiness and get_seq_length() match the uses in demo/inference.py:61,98."""

    def __init__(self, engine: "Engine", max_tokens_hint: int = 0):
        self.engine = engine
        h = C.c_void_p()
        _C.check(_C.lib().vlo_session_create(engine._h, max_tokens_hint, C.byref(h)))
        self._h = h
        engine._sessions.add(self)

    def __bool__(self):
        return True

    def get_seq_length(self) -> int:
        return int(_C.lib().vlo_session_len(self._h))

    __len__ = get_seq_length

    def reset(self):
        _C.check(_C.lib().vlo_session_reset(self._h))

    def close(self):
        if self._h:
            _C.lib().vlo_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fork(self, n_tokens: int, stream=None) -> "Session":
        """A new session holding a copy of the first ``n_tokens`` positions (trim_past_key_values(past, 0, n),
        models/modeling_live.py:170-171); this session is left untouched."""
        h = C.c_void_p()
        _C.check(_C.lib().vlo_session_fork(self._h, n_tokens, C.byref(h), _stream_handle(stream)))
        out = Session.__new__(Session)
        out.engine, out._h = self.engine, h
        self.engine._sessions.add(out)
        return out

    def crop(self, n_tokens: int):
        """Forget every position >= n_tokens in place."""
        _C.check(_C.lib().vlo_session_crop(self._h, n_tokens))

    def read_kv(self, layer, which, kv_head, t0, t1):
        out = torch.empty(t1 - t0, self.engine.head_dim, dtype=torch.bfloat16, device=self.engine.device)
        _C.check(_C.lib().vlo_session_read_kv(self._h, layer, which, kv_head, t0, t1, _ptr(out), _stream_handle()))
        return out


class Engine:
    def __init__(self, cfg: EngineConfig, device: int | str | torch.device = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("videollm-online_amd needs a ROCm GPU (MI355X); there is no CPU path")
        self.cfg = cfg
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.head_dim = cfg.hidden_size // cfg.num_attention_heads
        h = C.c_void_p()
        cc = cfg.to_c()
        _C.check(_C.lib().vlo_engine_create(C.byref(cc), self.device.index, C.byref(h)))
        self._h = h
        self._finalized = False
        self._sessions = weakref.WeakSet()      # sessions borrow the engine's KV pool: closed before the engine is

    # ---- weights -------------------------------------------------------------------------
    _STREAMED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "gate_proj.weight", "up_proj.weight",
                 "down_proj.weight")

    def load_weight(self, name: str, t: torch.Tensor):
        t = t.detach().contiguous()
        streamed = (name.startswith("model.layers.") and name.endswith(self._STREAMED)) or name == "lm_head.weight"
        if self.cfg.weight_dtype == "fp8" and streamed and t.dtype != torch.float8_e4m3fn:
            from .checkpoint import quantize_fp8_per_channel
            q, scale = quantize_fp8_per_channel(t.to(self.device))
            self.load_weight(name, q)
            self.load_weight(name + "_scale", scale)
            return
        if t.dtype == torch.float8_e4m3fn:
            dt, t = _C.DT_FP8_E4M3, t.view(torch.uint8)
        elif t.dtype in _DT:
            dt = _DT[t.dtype]
        else:
            raise TypeError(f"{name}: unsupported dtype {t.dtype}")
        shape = (C.c_int64 * t.dim())(*t.shape)
        _C.check(_C.lib().vlo_engine_load_weight(self._h, name.encode(), _ptr(t), dt, shape, t.dim()))

    def load_weights(self, weights: dict):
        for k, v in weights.items():
            self.load_weight(k, v)

    def finalize(self):
        _C.check(_C.lib().vlo_engine_finalize(self._h))
        self._finalized = True
        return self

    @property
    def weight_bytes(self) -> int:
        return int(_C.lib().vlo_engine_weight_bytes(self._h))

    def step_algorithmic_bytes(self, Lc: int, n: int) -> float:
        return float(_C.lib().vlo_step_algorithmic_bytes(self._h, Lc, n))

    def profile_enable(self, stride: int = 1):
        _C.check(_C.lib().vlo_profile_enable(self._h, stride))

    def profile_read(self):
        """(timed launches, total ms, algorithmic bytes per launch) of the dominant kernel (gate/up GEMV)."""
        n, ms, b = C.c_int64(0), C.c_double(0), C.c_double(0)
        _C.check(_C.lib().vlo_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value

    def profile_calibrate(self, stream=None) -> float:
        """microseconds an empty HIP-event bracket reads on this stream"""
        us = C.c_double(0)
        _C.check(_C.lib().vlo_profile_calibrate(self._h, _stream_handle(stream), C.byref(us)))
        return us.value

    def new_session(self, max_tokens_hint: int = 0) -> Session:
        return Session(self, max_tokens_hint)

    def close(self):
        if self._h:
            for sess in list(self._sessions):   # a session handle must not outlive its engine (include/vlo.h)
                sess.close()
            _C.lib().vlo_engine_destroy(self._h)
            self._h = None

    # ---- ops (each mirrors one reference call; see include/vlo.h) --------------------------
    def embed(self, ids: torch.Tensor, stream=None) -> torch.Tensor:
        ids = ids.to(device=self.device, dtype=torch.long).contiguous().view(-1)
        out = torch.empty(ids.numel(), self.cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        _C.check(_C.lib().vlo_embed(self._h, _ptr(ids), ids.numel(), _ptr(out), _stream_handle(stream)))
        return out

    def step_input(self, ids: list, frame_rows: torch.Tensor | None, out: torch.Tensor, stream=None) -> torch.Tensor:
        """`torch.cat([embed(ids), frame_rows])` (demo/inference.py:61-68) written by one launch into the caller's staging
        buffer ``out`` (bf16 [>= k + rows, H]); ``ids`` are host integers and travel as kernel arguments.  Returns the view
        ``out[:k + rows]``."""
        k = len(ids)
        rows = 0 if frame_rows is None else frame_rows.shape[0]
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape[0] >= k + rows and out.shape[1] == self.cfg.hidden_size
        if rows:
            assert frame_rows.dtype == torch.bfloat16 and frame_rows.is_contiguous() and frame_rows.shape[1] == self.cfg.hidden_size
        arr = (C.c_int64 * max(k, 1))(*ids)
        _C.check(_C.lib().vlo_step_input(self._h, arr, k, _ptr(frame_rows) if rows else None, rows, _ptr(out), _stream_handle(stream)))
        return out[:k + rows]

    def connector(self, feats: torch.Tensor, stream=None) -> torch.Tensor:
        feats = feats.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.vision_hidden_size)
        out = torch.empty(feats.shape[0], self.cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        _C.check(_C.lib().vlo_connector(self._h, _ptr(feats), feats.shape[0], _ptr(out), _stream_handle(stream)))
        return out

    def _check_frames(self, frames_u8: torch.Tensor):
        """The C side reads B*3*R*R bytes from the pointer: anything but uint8 [B,3,R,R] at the tower's resolution would be
        read out of bounds or silently misread (the reference fails on the position-embedding shape in the same case)."""
        if self.cfg.vit is None:
            raise RuntimeError("engine built without a vision tower")
        R = self.cfg.vit["image_size"]
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or tuple(frames_u8.shape[1:]) != (3, R, R):
            raise ValueError(f"frames must be uint8 [B,3,{R},{R}], got {frames_u8.dtype} {tuple(frames_u8.shape)}")
        if not frames_u8.is_cuda:
            raise ValueError("frames must live on the engine's device")

    def visual_embed(self, frames_u8: torch.Tensor, stream=None, out: torch.Tensor | None = None) -> torch.Tensor:
        self._check_frames(frames_u8)
        frames_u8 = frames_u8.contiguous()
        B = frames_u8.shape[0]
        if out is None:
            out = torch.empty(B * self.cfg.frame_num_tokens, self.cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        _C.check(_C.lib().vlo_visual_embed(self._h, _ptr(frames_u8), B, _ptr(out), _stream_handle(stream)))
        return out

    def frame_ingest(self, frames: torch.Tensor, layout: str | None = None, resolution: int = 0, cubic_a: float = -0.6,
                     out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """Decoded uint8 RGB frames of any size -> uint8 [T,3,R,R] as the reference's ffmpeg preparation + read_video yield them
        (data/utils.py:51-66, demo/inference.py:112; include/vlo.h vlo_frame_ingest).  ``layout``: "THWC" (decoder output) or
        "TCHW"; default = by the position of the size-3 axis."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or not frames.is_cuda:
            raise ValueError("frames must be a uint8 [T,H,W,3] or [T,3,H,W] tensor on the engine's device")
        if layout is None:
            layout = "THWC" if frames.shape[-1] == 3 and frames.shape[1] != 3 else "TCHW"
        if layout == "THWC":
            T, H, W, c = frames.shape
        else:
            T, c, H, W = frames.shape
        if c != 3:
            raise ValueError(f"expected 3 colour channels, got {tuple(frames.shape)} as {layout}")
        R = resolution or (self.cfg.vit or {}).get("image_size", 0)
        if R <= 0:
            raise ValueError("resolution must be given for an engine without a vision tower")
        frames = frames.contiguous()
        if out is None:
            out = torch.empty(T, 3, R, R, dtype=torch.uint8, device=self.device)
        assert out.dtype == torch.uint8 and out.is_contiguous() and tuple(out.shape) == (T, 3, R, R)
        _C.check(_C.lib().vlo_frame_ingest(self._h, _ptr(frames), T, H, W, 0 if layout == "THWC" else 1, R, cubic_a, _ptr(out),
                                           _stream_handle(stream)))
        return out

    def vision_tokens(self, frames_u8: torch.Tensor, stream=None) -> torch.Tensor:
        """CLS + pooled tokens before the connector: bf16 [B, frame_num_tokens, vision_hidden_size]."""
        self._check_frames(frames_u8)
        frames_u8 = frames_u8.contiguous()
        B = frames_u8.shape[0]
        out = torch.empty(B, self.cfg.frame_num_tokens, self.cfg.vision_hidden_size, dtype=torch.bfloat16, device=self.device)
        _C.check(_C.lib().vlo_vision_tokens(self._h, _ptr(frames_u8), B, _ptr(out), _stream_handle(stream)))
        return out

    def encode_video(self, frames_u8: torch.Tensor, batch_size: int = 256) -> torch.Tensor:
        """Offline feature extraction of one video, as data/utils.py:97-101 does it: split into batches of
        ``batch_size`` frames, encode, concatenate, keep bf16 -> [T, frame_num_tokens, vision_hidden_size]."""
        outs = [self.vision_tokens(frames_u8[i:i + batch_size].to(self.device)) for i in range(0, frames_u8.shape[0], batch_size)]
        return torch.cat(outs)

    def joint_embed(self, ids: torch.Tensor, frame_rows: torch.Tensor | None, v_placeholder_id: int, stream=None) -> torch.Tensor:
        """models/modeling_live.py:29-42: token embeddings with the placeholder rows replaced, in order, by ``frame_rows``
        (bf16 [num_frames * frame_num_tokens, H], from visual_embed / connector).  Raises on a count mismatch."""
        ids = ids.to(device=self.device, dtype=torch.long).contiguous().view(-1)
        n_rows = 0
        if frame_rows is not None and frame_rows.numel():
            frame_rows = frame_rows.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.hidden_size)
            n_rows = frame_rows.shape[0]
        out = torch.empty(ids.numel(), self.cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        _C.check(_C.lib().vlo_joint_embed(self._h, _ptr(ids), ids.numel(), v_placeholder_id, _ptr(frame_rows) if n_rows else None,
                                          n_rows, _ptr(out), _stream_handle(stream)))
        return out

    def logit_rows(self, logits: torch.Tensor, labels: torch.Tensor | None = None, interval_id: int = -1, stream=None) -> dict:
        """Per-row statistics of bf16 logits [n, V] (see include/vlo.h::vlo_logit_rows): lse, argmax, label_logit,
        p_interval, p_argmax — device tensors of n elements."""
        logits = logits.view(-1, self.cfg.vocab_size)
        assert logits.dtype == torch.bfloat16 and logits.is_cuda and logits.is_contiguous()
        n = logits.shape[0]
        if labels is not None:
            labels = labels.to(device=self.device, dtype=torch.long).contiguous().view(-1)
            assert labels.numel() == n
        f = lambda dt: torch.empty(n, dtype=dt, device=self.device)
        out = dict(lse=f(torch.float32), argmax=f(torch.long), label_logit=f(torch.float32), p_interval=f(torch.float32),
                   p_argmax=f(torch.long))
        _C.check(_C.lib().vlo_logit_rows(self._h, _ptr(logits), n, _ptr(labels) if labels is not None else None, interval_id,
                                         _ptr(out["lse"]), _ptr(out["argmax"]), _ptr(out["label_logit"]), _ptr(out["p_interval"]),
                                         _ptr(out["p_argmax"]), _stream_handle(stream)))
        return out

    def llm_step(self, session: Session, embeds: torch.Tensor, want_last=True, want_all=False, stream=None):
        """Returns (last_logits [V] bf16 | None, all_logits [n,V] bf16 | None)."""
        embeds = embeds.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.hidden_size)
        n = embeds.shape[0]
        last = torch.empty(self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device) if want_last else None
        allr = torch.empty(n, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device) if want_all else None
        _C.check(_C.lib().vlo_llm_step(session._h, _ptr(embeds), n, _ptr(last) if want_last else None,
                                       _ptr(allr) if want_all else None, _stream_handle(stream)))
        return last, allr

    def stream_sample(self, session: Session, threshold: float, interval_id: int, stream=None, tok_out=None, p_out=None):
        tok = tok_out if tok_out is not None else torch.empty(1, dtype=torch.long, device=self.device)
        p = p_out if p_out is not None else torch.empty(1, dtype=torch.float32, device=self.device)
        _C.check(_C.lib().vlo_stream_sample(session._h, threshold, interval_id, _ptr(tok), _ptr(p), _stream_handle(stream)))
        return tok, p

    def greedy_generate(self, session: Session, embeds: torch.Tensor, eos_token_id: int, inplace_output_ids: torch.Tensor,
                        force_len: int = 0, stream=None) -> int:
        embeds = embeds.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.hidden_size)
        assert inplace_output_ids.dtype == torch.long and inplace_output_ids.is_cuda and inplace_output_ids.is_contiguous()
        n = C.c_int(0)
        _C.check(_C.lib().vlo_greedy_generate(session._h, _ptr(embeds), embeds.shape[0], eos_token_id,
                                              _ptr(inplace_output_ids), inplace_output_ids.numel(), force_len,
                                              C.byref(n), _stream_handle(stream)))
        return n.value


def test_gemv(x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """y[n,N] f32 = x[n,K] @ W[N,K]^T through the packed MFMA GEMV (unit tests)."""
    x, W = x.contiguous(), W.contiguous()
    y = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=x.device)
    _C.check(_C.lib().vlo_test_gemv(_ptr(x), _ptr(W), _ptr(y), x.shape[0], W.shape[0], W.shape[1], _stream_handle()))
    return y


def test_gemv_fp8(x: torch.Tensor, Wq: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """y[n,N] f32 = (x[n,K] @ Wq[N,K]^T) * scale[N] through the fp8 e4m3 weight image (unit tests)."""
    x, Wq, scale = x.contiguous(), Wq.contiguous().view(torch.uint8), scale.float().contiguous()
    y = torch.empty(x.shape[0], Wq.shape[0], dtype=torch.float32, device=x.device)
    _C.check(_C.lib().vlo_test_gemv_fp8(_ptr(x), _ptr(Wq), _ptr(scale), _ptr(y), x.shape[0], Wq.shape[0], Wq.shape[1], _stream_handle()))
    return y


def fp8_row_order(K: int) -> torch.Tensor:
    """pos[k] = byte of a quantised activation row that holds column k (csrc/prefill.h::vlo_fp8_row_pos: inside every 64 k's the order of
    the fp8 weight image's registers)"""
    k = torch.arange(K)
    r = k & 63
    return (k >> 6) * 64 + ((r & 31) >> 3) * 16 + (r >> 5) * 8 + (r & 7)


def test_gemm_fp8(x: torch.Tensor, Wq: torch.Tensor, scale: torch.Tensor, iters: int = 0):
    """The long-input W8A8 GEMM (prefill_act_dtype='fp8'): x bf16 [M,K] -> (y f32 [M,N], codes e4m3 [M,K] in column order, row scales f32 [M]
    [, microseconds per GEMM when iters > 0]) — unit tests and tools/probe_prefill.py."""
    x, Wq, scale = x.contiguous(), Wq.contiguous().view(torch.uint8), scale.float().contiguous()
    M, K, N = x.shape[0], x.shape[1], Wq.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    xq = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    xs = torch.empty(M, dtype=torch.float32, device=x.device)
    us = C.c_double(0.0)
    _C.check(_C.lib().vlo_test_gemm_fp8(_ptr(x), _ptr(Wq), _ptr(scale), _ptr(y), _ptr(xq), _ptr(xs), M, N, K, iters, C.byref(us), _stream_handle()))
    codes = xq[:, fp8_row_order(K).to(x.device)].view(torch.float8_e4m3fn)
    return (y, codes, xs, us.value) if iters > 0 else (y, codes, xs)


class TpSession:
    """KV handle of a tensor-parallel group: one KV shard (this rank's kv heads) per local rank."""

    def __init__(self, group: "TpGroup", max_tokens_hint: int = 0):
        self.group = group
        h = C.c_void_p()
        _C.check(_C.lib().vlo_tp_session_create(group._g, max_tokens_hint, C.byref(h)))
        self._h = h

    def __bool__(self):
        return True

    def get_seq_length(self) -> int:
        return int(_C.lib().vlo_tp_session_len(self._h))

    __len__ = get_seq_length

    def reset(self):
        _C.check(_C.lib().vlo_tp_session_reset(self._h))

    def close(self):
        if self._h:
            _C.lib().vlo_tp_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fork(self, n_tokens: int, stream=None) -> "TpSession":
        """A new tensor-parallel session holding a copy of the first ``n_tokens`` positions of every local KV shard
        (trim_past_key_values(past, 0, n), models/modeling_live.py:170-171); one process per GPU: every rank calls it."""
        h = C.c_void_p()
        _C.check(_C.lib().vlo_tp_session_fork(self._h, n_tokens, C.byref(h), _stream_handle(stream)))
        out = TpSession.__new__(TpSession)
        out.group, out._h = self.group, h
        return out

    def crop(self, n_tokens: int):
        """Forget every position >= n_tokens in place, on every local KV shard."""
        _C.check(_C.lib().vlo_tp_session_crop(self._h, n_tokens))


class TpGroup:
    """Tensor-parallel Llama (north_star TP; include/vlo.h `vlo_tp_*`) behind the same surface as Engine for the STREAMING
    path: LiveInfer, LiveModel(inputs_embeds=...), joint_embed, the samplers and fast_greedy_generate run on it unchanged.
    Not supported under TP: KV fork / crop (`TpSession.fork/crop` raise NotImplementedError), hence `stream_evaluate` and
    `trim_past_key_values` — run teacher-forced evaluation on a TP=1 engine.

    * ``TpGroup(cfg, tp_size, device=0)``: single process, ``tp_size`` logical ranks on one device (exchanges are
      device kernels) — validates the sharding arithmetic without a multi-GPU box.
    * ``TpGroup(cfg, tp_size, device=local_rank, rank=r, unique_id=bytes)``: one process per GPU; exchanges are RCCL
      all-reduce / all-gather.  ``unique_id`` comes from ``TpGroup.unique_id()`` on rank 0, broadcast by the caller
      (e.g. ``torch.distributed.broadcast_object_list``).
    * ``allreduce="p2p"`` (either mode): the exchanges are the one-shot peer-to-peer all-reduce over xGMI fused with the
      residual add + RMSNorm (include/vlo.h `vlo_tp_p2p_*`) instead of RCCL calls / sum kernels.  One process per GPU:
      pass ``handle_allgather`` — a callable taking this rank's 64-byte mailbox handle and returning every rank's, in
      rank order (e.g. built on ``torch.distributed.all_gather_object``); no RCCL unique id is needed then (give one anyway,
      with ``frame_parallel=True``, to encode frame-parallel: RCCL then carries only the all-gather of the frame embeddings).
    Weights are given in FULL; every rank slices its shard.  ViT, connector and embeddings are replicated."""

    def __init__(self, cfg: EngineConfig, tp_size: int, device: int = 0, rank: int | None = None, unique_id: bytes | None = None,
                 allreduce: str = "default", handle_allgather=None, frame_parallel: bool = False):
        from dataclasses import replace
        self.cfg = cfg
        self.tp_size = tp_size
        ranks = list(range(tp_size)) if rank is None else [rank]
        self._uid = unique_id
        if allreduce not in ("default", "rccl", "p2p"):
            raise ValueError(f"allreduce must be 'default', 'rccl' or 'p2p', not {allreduce!r}")
        self.allreduce = "p2p" if allreduce == "p2p" else ("rccl" if rank is not None else "kernel")
        self.frame_parallel = frame_parallel
        self._handle_allgather = handle_allgather
        if rank is not None and tp_size > 1:
            if self.allreduce == "p2p" and handle_allgather is None:
                raise ValueError("one-process-per-GPU p2p TP needs handle_allgather (mailbox handles of all ranks)")
            if self.allreduce == "rccl" and unique_id is None:
                raise ValueError("one-process-per-GPU TP needs the RCCL unique id")
        self.engines = [Engine(replace(cfg, tp_rank=r, tp_size=tp_size, vit=cfg.vit if i == 0 else None), device)
                        for i, r in enumerate(ranks)]
        self.device = self.engines[0].device
        self.head_dim = self.engines[0].head_dim
        self._g = None

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _C.check(_C.lib().vlo_tp_unique_id(buf))
        return buf.raw

    def load_weight(self, name: str, t: torch.Tensor):
        e0 = self.engines[0]
        streamed = (name.startswith("model.layers.") and name.endswith(e0._STREAMED)) or name == "lm_head.weight"
        if self.cfg.weight_dtype == "fp8" and streamed and t.dtype != torch.float8_e4m3fn:
            from .checkpoint import quantize_fp8_per_channel      # once for all local ranks (scales are per FULL output row)
            q, scale = quantize_fp8_per_channel(t.to(self.device))
            self.load_weight(name, q)
            self.load_weight(name + "_scale", scale)
            return
        for i, e in enumerate(self.engines):
            if name.startswith("vision.") and i > 0:
                continue                      # the vision tower lives on the first local engine only
            e.load_weight(name, t)

    def load_weights(self, weights: dict):
        for k, v in weights.items():
            self.load_weight(k, v)

    def finalize(self):
        for e in self.engines:
            e.finalize()
        arr = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        g = C.c_void_p()
        uid = C.create_string_buffer(self._uid, 128) if self._uid is not None else None
        _C.check(_C.lib().vlo_tp_group_create(arr, len(self.engines), uid, C.byref(g)))
        self._g = g
        if self.allreduce == "p2p" and self.tp_size > 1:
            if len(self.engines) == self.tp_size:          # logical ranks of one process: mailboxes are local allocations
                _C.check(_C.lib().vlo_tp_p2p_enable(g, None))
            else:
                mine = C.create_string_buffer(64)
                _C.check(_C.lib().vlo_tp_p2p_export(g, mine))
                handles = list(self._handle_allgather(mine.raw))
                if len(handles) != self.tp_size or any(len(h) != 64 for h in handles):
                    raise ValueError("handle_allgather must return tp_size handles of 64 bytes, in rank order")
                _C.check(_C.lib().vlo_tp_p2p_enable(g, C.create_string_buffer(b"".join(handles), 64 * self.tp_size)))
        return self

    def comm_info(self) -> dict:
        """{nranks, rank} as RCCL reports them for the group's communicator (0 / -1 without one)."""
        n, r = C.c_int(0), C.c_int(-1)
        _C.check(_C.lib().vlo_tp_comm_info(self._g, C.byref(n), C.byref(r)))
        return dict(nranks=n.value, rank=r.value)

    def p2p_status(self) -> dict:
        """{enabled, timed_out, uncached_mailbox} of the peer-to-peer exchange (all 0 when it is not in use)."""
        en, to, uc = C.c_int(0), C.c_int(0), C.c_int(0)
        _C.check(_C.lib().vlo_tp_p2p_status(self._g, C.byref(en), C.byref(to), C.byref(uc)))
        return dict(enabled=en.value, timed_out=to.value, uncached_mailbox=uc.value)

    @property
    def weight_bytes(self) -> int:
        return sum(e.weight_bytes for e in self.engines)

    def step_algorithmic_bytes(self, Lc: int, n: int) -> float:
        return self.engines[0].step_algorithmic_bytes(Lc, n)

    def profile_enable(self, stride: int = 1):
        pass

    def new_session(self, max_tokens_hint: int = 0) -> TpSession:
        return TpSession(self, max_tokens_hint)

    def embed(self, ids, stream=None):
        return self.engines[0].embed(ids, stream)

    def connector(self, feats, stream=None):
        return self.engines[0].connector(feats, stream)

    # embeddings are replicated and the logits arrive gathered, so these run on the first local engine
    def step_input(self, ids, frame_rows, out, stream=None):
        return self.engines[0].step_input(ids, frame_rows, out, stream)

    def joint_embed(self, ids, frame_rows, v_placeholder_id, stream=None):
        return self.engines[0].joint_embed(ids, frame_rows, v_placeholder_id, stream)

    def logit_rows(self, logits, labels=None, interval_id=-1, stream=None):
        return self.engines[0].logit_rows(logits, labels, interval_id, stream)

    def vision_tokens(self, frames_u8, stream=None):
        return self.engines[0].vision_tokens(frames_u8, stream)

    def frame_ingest(self, frames, layout=None, resolution=0, cubic_a=-0.6, out=None, stream=None):
        return self.engines[0].frame_ingest(frames, layout, resolution, cubic_a, out, stream)

    def visual_embed(self, frames_u8, stream=None, out=None):
        """Replicated tower (every rank encodes every frame) unless this is a one-process-per-GPU group WITH an RCCL communicator
        (``unique_id`` given — also next to ``allreduce="p2p"``, where RCCL then carries only this all-gather) and
        ``frame_parallel`` set: then rank r encodes frames r, r + T, ... and ONE all-gather hands every rank all the
        [frame_num_tokens, H] embeddings (north_star's frame-embedding broadcast; 81 920 B per frame for Llama-3-8B).
        Every rank must call it with the same number of frames."""
        if not (self.frame_parallel and self._uid is not None and len(self.engines) == 1 and self.tp_size > 1):
            return self.engines[0].visual_embed(frames_u8, stream, out)
        e, T, r = self.engines[0], self.tp_size, self.engines[0].cfg.tp_rank
        B, rows, H = frames_u8.shape[0], self.cfg.frame_num_tokens, self.cfg.hidden_size
        k = (B + T - 1) // T                                   # frames per rank, the last ranks' shares padded
        # the staging tensors below are filled / read by torch kernels: they must run on the stream the encode and the all-gather
        # are enqueued on, whatever torch's current stream is
        ts = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(ts):
            mine = frames_u8[r::T].contiguous()
            send = torch.zeros(k * rows, H, dtype=torch.bfloat16, device=self.device)
            if mine.shape[0]:
                e.visual_embed(mine, ts, out=send[:mine.shape[0] * rows])
            recv = torch.empty(T, k * rows, H, dtype=torch.bfloat16, device=self.device)
            _C.check(_C.lib().vlo_tp_allgather(self._g, _ptr(send), _ptr(recv), send.numel() * 2, _stream_handle(ts)))
            if out is None:
                out = torch.empty(B * rows, H, dtype=torch.bfloat16, device=self.device)
            # frame i was encoded by rank i % T as its (i // T)-th frame
            out.view(B, rows, H).copy_(recv.view(T, k, rows, H).transpose(0, 1).reshape(T * k, rows, H)[:B])
        return out

    def llm_step(self, session: TpSession, embeds: torch.Tensor, want_last=True, want_all=False, stream=None):
        embeds = embeds.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.hidden_size)
        n = embeds.shape[0]
        last = torch.empty(self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device) if want_last else None
        allr = torch.empty(n, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device) if want_all else None
        _C.check(_C.lib().vlo_tp_llm_step(session._h, _ptr(embeds), n, _ptr(last) if want_last else None,
                                          _ptr(allr) if want_all else None, _stream_handle(stream)))
        return last, allr

    def stream_sample(self, session: TpSession, threshold: float, interval_id: int, stream=None, tok_out=None, p_out=None):
        tok = tok_out if tok_out is not None else torch.empty(1, dtype=torch.long, device=self.device)
        p = p_out if p_out is not None else torch.empty(1, dtype=torch.float32, device=self.device)
        _C.check(_C.lib().vlo_tp_stream_sample(session._h, threshold, interval_id, _ptr(tok), _ptr(p), _stream_handle(stream)))
        return tok, p

    def greedy_generate(self, session: TpSession, embeds: torch.Tensor, eos_token_id: int, inplace_output_ids: torch.Tensor,
                        force_len: int = 0, stream=None) -> int:
        embeds = embeds.to(device=self.device, dtype=torch.bfloat16).contiguous().view(-1, self.cfg.hidden_size)
        n = C.c_int(0)
        _C.check(_C.lib().vlo_tp_greedy_generate(session._h, _ptr(embeds), embeds.shape[0], eos_token_id,
                                                 _ptr(inplace_output_ids), inplace_output_ids.numel(), force_len,
                                                 C.byref(n), _stream_handle(stream)))
        return n.value

    def bench_exchange(self, session: TpSession, m: int, iters: int = 200, stream=None) -> float:
        """microseconds of ONE exchange of m rows (all-reduce + residual add + RMSNorm), timed over ``iters`` back-to-back ones;
        every rank must call it; the session is scratch afterwards (reset it)."""
        us = C.c_double(0)
        _C.check(_C.lib().vlo_tp_bench_exchange(session._h, m, iters, C.byref(us), _stream_handle(stream)))
        return us.value

    def close(self):
        if self._g:
            _C.lib().vlo_tp_group_destroy(self._g)
            self._g = None
        for e in self.engines:
            e.close()
