"""Drives tests/hip_emul/_build/libvlo_emul.so — the engine's SOURCES compiled for the CPU (build_emul.py) — through the
same C ABI as the product (include/vlo.h, argument types from videollm_online_amd/_C.py::bind).  "Device" pointers are
host pointers of torch CPU tensors.  Test infrastructure only: the product's Engine refuses to run without a GPU."""
import ctypes as C
import os

import torch

from videollm_online_amd import _C

from . import build_emul

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = build_emul.build()
        if path is None:
            return None
        # built with -Bsymbolic: never binds to libvlo.so (same symbol names) when both are loaded in one process
        _LIB = _C.bind(C.CDLL(path, mode=os.RTLD_LOCAL))
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libvlo_emul error {rc}: {lib().vlo_last_error().decode()}")


def _ptr(t):
    return C.c_void_p(t.data_ptr())


_DT = {torch.float32: _C.DT_F32, torch.bfloat16: _C.DT_BF16, torch.float16: _C.DT_F16}


def test_gemv(x, W):
    """y[n,N] f32 = x[n,K] @ W[N,K]^T through the packed MFMA GEMV of the emulated library (include/vlo.h vlo_test_gemv)."""
    x, W = x.to(torch.bfloat16).contiguous(), W.to(torch.bfloat16).contiguous()
    y = torch.zeros(x.shape[0], W.shape[0], dtype=torch.float32)
    check(lib().vlo_test_gemv(_ptr(x), _ptr(W), _ptr(y), x.shape[0], W.shape[0], W.shape[1], None))
    return y


def test_gemv_fp8(x, Wq, scale):
    """y[n,N] f32 = (x @ Wq^T) * scale through the fp8 e4m3 weight image of the emulated library (vlo_test_gemv_fp8)."""
    x, Wq, scale = x.to(torch.bfloat16).contiguous(), Wq.contiguous().view(torch.uint8), scale.float().contiguous()
    y = torch.zeros(x.shape[0], Wq.shape[0], dtype=torch.float32)
    check(lib().vlo_test_gemv_fp8(_ptr(x), _ptr(Wq), _ptr(scale), _ptr(y), x.shape[0], Wq.shape[0], Wq.shape[1], None))
    return y


def test_gemm_fp8(x, Wq, scale):
    """The W8A8 prefill GEMM of the emulated library (vlo_test_gemm_fp8): (y f32 [M,N], e4m3 codes [M,K] in column order, row scales f32 [M])."""
    from videollm_online_amd.engine import fp8_row_order
    x, Wq, scale = x.to(torch.bfloat16).contiguous(), Wq.contiguous().view(torch.uint8), scale.float().contiguous()
    M, K, N = x.shape[0], x.shape[1], Wq.shape[0]
    y = torch.zeros(M, N, dtype=torch.float32)
    xq = torch.zeros(M, K, dtype=torch.uint8)
    xs = torch.zeros(M, dtype=torch.float32)
    check(lib().vlo_test_gemm_fp8(_ptr(x), _ptr(Wq), _ptr(scale), _ptr(y), _ptr(xq), _ptr(xs), M, N, K, 0, None, None))
    return y, xq[:, fp8_row_order(K)].view(torch.float8_e4m3fn), xs


def gemv_plan(K, allow_ksplit):
    out = (C.c_int * 4)()
    check(lib().vlo_debug_gemv_plan(K, int(allow_ksplit), out))
    return tuple(out)


class EmulEngine:
    def __init__(self, spec, kv_pool_tokens=1024, tp_rank=0, tp_size=1, vit=None, weight_dtype=0, prefill_act_dtype=0):
        self.spec, self.vit = spec, vit
        c = _C.VloConfig()
        c.weight_dtype = weight_dtype                    # 1: fp8 e4m3 image of the streamed projections (+ "<name>_scale")
        c.prefill_act_dtype = prefill_act_dtype          # 1 (fp8 engines): W8A8 prefill GEMMs on the fp8 MFMA
        if vit is not None:
            c.has_vit = 1
            c.vit_hidden_size, c.vit_intermediate_size = vit.hidden_size, vit.intermediate_size
            c.vit_num_layers, c.vit_num_heads = vit.num_layers, vit.num_heads
            c.vit_image_size, c.vit_patch_size, c.vit_ln_eps = vit.image_size, vit.patch_size, vit.ln_eps
        c.abi_version = _C.VLO_ABI_VERSION
        c.hidden_size, c.intermediate_size, c.num_layers = spec.hidden_size, spec.intermediate_size, spec.num_layers
        c.num_heads, c.num_kv_heads, c.vocab_size = spec.num_heads, spec.num_kv_heads, spec.vocab_size
        c.rope_theta, c.rms_eps = spec.rope_theta, spec.rms_eps
        c.vision_hidden_size, c.frame_num_tokens, c.pool_h, c.pool_w = spec.vision_hidden_size, 10, 3, 3
        if vit is not None:
            c.frame_num_tokens, (c.pool_h, c.pool_w) = vit.frame_num_tokens, vit.pooled
        c.kv_pool_tokens, c.tp_rank, c.tp_size = kv_pool_tokens, tp_rank, tp_size
        h = C.c_void_p()
        check(lib().vlo_engine_create(C.byref(c), 0, C.byref(h)))
        self._h = h
        self.sessions = []

    def load_weights(self, weights, inv_freq=None):
        for name, t in weights.items():
            if name.startswith("vision.") and self.vit is None:
                continue
            t = t.detach().contiguous()
            dt = _C.DT_FP8_E4M3 if t.dtype == torch.float8_e4m3fn else _DT[t.dtype]
            if t.dtype == torch.float8_e4m3fn:
                t = t.view(torch.uint8)
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(lib().vlo_engine_load_weight(self._h, name.encode(), _ptr(t), dt, shape, t.dim()))
        if inv_freq is not None:
            t = inv_freq.float().contiguous()
            check(lib().vlo_engine_load_weight(self._h, b"rope.inv_freq", _ptr(t), _C.DT_F32, (C.c_int64 * 1)(t.numel()), 1))
        check(lib().vlo_engine_finalize(self._h))
        return self

    def new_session(self):
        h = C.c_void_p()
        check(lib().vlo_session_create(self._h, 0, C.byref(h)))
        self.sessions.append(h)
        return h

    def session_len(self, s):
        return int(lib().vlo_session_len(s))

    def llm_step(self, s, embeds, want_all=True):
        x = embeds.to(torch.bfloat16).contiguous().view(-1, self.spec.hidden_size)
        n, V = x.shape[0], self.spec.vocab_size
        last = torch.zeros(V, dtype=torch.bfloat16)
        allr = torch.zeros(n, V, dtype=torch.bfloat16) if want_all else None
        check(lib().vlo_llm_step(s, _ptr(x), n, _ptr(last), _ptr(allr) if want_all else None, None))
        return last, allr

    def embed(self, ids):
        ids = ids.to(torch.long).contiguous().view(-1)
        out = torch.zeros(ids.numel(), self.spec.hidden_size, dtype=torch.bfloat16)
        check(lib().vlo_embed(self._h, _ptr(ids), ids.numel(), _ptr(out), None))
        return out

    def stream_sample(self, s, threshold, interval_id):
        tok, p = torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.float32)
        check(lib().vlo_stream_sample(s, threshold, interval_id, _ptr(tok), _ptr(p), None))
        return int(tok), float(p)

    def greedy_generate(self, s, embeds, eos, max_new, force_len=0):
        x = embeds.to(torch.bfloat16).contiguous().view(-1, self.spec.hidden_size)
        ids = torch.zeros(max_new, dtype=torch.long)
        n = C.c_int(0)
        check(lib().vlo_greedy_generate(s, _ptr(x), x.shape[0], eos, _ptr(ids), max_new, force_len, C.byref(n), None))
        return ids[:n.value].tolist()

    def visual_embed(self, frames_u8, stream=None):
        """uint8 [B,3,R,R] -> bf16 [B * frame_num_tokens, H]; stream = a non-null fake handle takes the captured-graph path"""
        f = frames_u8.contiguous()
        out = torch.zeros(f.shape[0] * self.vit.frame_num_tokens, self.spec.hidden_size, dtype=torch.bfloat16)
        check(lib().vlo_visual_embed(self._h, _ptr(f), f.shape[0], _ptr(out), stream))
        return out

    def frame_ingest(self, frames_u8, layout, resolution, cubic_a=-0.6):
        """decoded uint8 frames [T,H,W,3] (layout 0) or [T,3,H,W] (layout 1) -> uint8 [T,3,R,R] (include/vlo.h vlo_frame_ingest)"""
        f = frames_u8.contiguous()
        T = f.shape[0]
        H, W = (f.shape[1], f.shape[2]) if layout == 0 else (f.shape[2], f.shape[3])
        out = torch.zeros(T, 3, resolution, resolution, dtype=torch.uint8)
        check(lib().vlo_frame_ingest(self._h, _ptr(f), T, H, W, layout, resolution, cubic_a, _ptr(out), None))
        return out

    def vision_tokens(self, frames_u8):
        f = frames_u8.contiguous()
        out = torch.zeros(f.shape[0], self.vit.frame_num_tokens, self.vit.hidden_size, dtype=torch.bfloat16)
        check(lib().vlo_vision_tokens(self._h, _ptr(f), f.shape[0], _ptr(out), None))
        return out

    def fork(self, s, n_tokens):
        h = C.c_void_p()
        check(lib().vlo_session_fork(s, n_tokens, C.byref(h), None))
        self.sessions.append(h)
        return h

    def crop(self, s, n_tokens):
        check(lib().vlo_session_crop(s, n_tokens))

    def joint_embed(self, ids, frame_rows, v_id):
        ids = ids.to(torch.long).contiguous().view(-1)
        rows = frame_rows.to(torch.bfloat16).contiguous().view(-1, self.spec.hidden_size)
        out = torch.zeros(ids.numel(), self.spec.hidden_size, dtype=torch.bfloat16)
        check(lib().vlo_joint_embed(self._h, _ptr(ids), ids.numel(), v_id, _ptr(rows), rows.shape[0], _ptr(out), None))
        return out

    def logit_rows(self, logits, labels, interval_id):
        logits = logits.contiguous()
        n = logits.shape[0]
        labels = labels.to(torch.long).contiguous()
        o = dict(lse=torch.zeros(n), argmax=torch.zeros(n, dtype=torch.long), label_logit=torch.zeros(n), p_interval=torch.zeros(n),
                 p_argmax=torch.zeros(n, dtype=torch.long))
        check(lib().vlo_logit_rows(self._h, _ptr(logits), n, _ptr(labels), interval_id, _ptr(o["lse"]), _ptr(o["argmax"]),
                                   _ptr(o["label_logit"]), _ptr(o["p_interval"]), _ptr(o["p_argmax"]), None))
        return o

    def close(self):
        for s in self.sessions:
            lib().vlo_session_destroy(s)
        self.sessions = []
        if self._h:
            lib().vlo_engine_destroy(self._h)
            self._h = None


class EmulTpGroup:
    """T logical ranks in one process (the single-process mode of include/vlo.h vlo_tp_*)."""

    def __init__(self, spec, T, weights, inv_freq, p2p=False, kv_pool_tokens=1024):
        self.spec, self.T = spec, T
        self.engines = [EmulEngine(spec, kv_pool_tokens, r, T).load_weights(weights, inv_freq) for r in range(T)]
        arr = (C.c_void_p * T)(*[e._h for e in self.engines])
        g = C.c_void_p()
        check(lib().vlo_tp_group_create(arr, T, None, C.byref(g)))
        self._g = g
        if p2p:
            check(lib().vlo_tp_p2p_enable(g, None))
        self.sessions = []

    def p2p_status(self):
        en, to, uc = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().vlo_tp_p2p_status(self._g, C.byref(en), C.byref(to), C.byref(uc)))
        return dict(enabled=en.value, timed_out=to.value, uncached_mailbox=uc.value)

    def new_session(self):
        h = C.c_void_p()
        check(lib().vlo_tp_session_create(self._g, 0, C.byref(h)))
        self.sessions.append(h)
        return h

    def session_len(self, s):
        return int(lib().vlo_tp_session_len(s))

    def bench_exchange(self, s, m, iters):
        us = C.c_double(-1)
        check(lib().vlo_tp_bench_exchange(s, m, iters, C.byref(us), None))
        return us.value

    def llm_step(self, s, embeds, want_all=True):
        x = embeds.to(torch.bfloat16).contiguous().view(-1, self.spec.hidden_size)
        n, V = x.shape[0], self.spec.vocab_size
        last = torch.zeros(V, dtype=torch.bfloat16)
        allr = torch.zeros(n, V, dtype=torch.bfloat16) if want_all else None
        check(lib().vlo_tp_llm_step(s, _ptr(x), n, _ptr(last), _ptr(allr) if want_all else None, None))
        return last, allr

    def close(self):
        for s in self.sessions:
            lib().vlo_tp_session_destroy(s)
        self.sessions = []
        if self._g:
            lib().vlo_tp_group_destroy(self._g)
            self._g = None
        for e in self.engines:
            e.close()


class EmulTpRankRccl:
    """ONE rank of a one-process-per-GPU group exchanging through the RCCL entry points (include/vlo.h: n_local = 1 and a
    unique id), which VLO_RCCL_LIBRARY points at the shared-memory stand-in tests/hip_emul/rccl_shim.cpp."""

    def __init__(self, spec, T, rank, weights, inv_freq, unique_id, kv_pool_tokens=1024, vit=None):
        self.spec, self.T, self.rank = spec, T, rank
        self.engine = EmulEngine(spec, kv_pool_tokens, rank, T, vit=vit).load_weights(weights, inv_freq)
        arr = (C.c_void_p * 1)(self.engine._h)
        g = C.c_void_p()
        check(lib().vlo_tp_group_create(arr, 1, C.create_string_buffer(unique_id, 128), C.byref(g)))
        self._g = g
        h = C.c_void_p()
        check(lib().vlo_tp_session_create(g, 0, C.byref(h)))
        self._s = h

    def comm_info(self):
        n, r = C.c_int(0), C.c_int(-1)
        check(lib().vlo_tp_comm_info(self._g, C.byref(n), C.byref(r)))
        return n.value, r.value

    def visual_embed_frame_parallel(self, frames_u8):
        """engine.TpGroup.visual_embed(frame_parallel=True) on the emulated library: this rank encodes frames rank, rank + T, ...,
        one vlo_tp_allgather, frame i comes back from rank i % T as its (i // T)-th frame"""
        T, r = self.T, self.rank
        B, rows, H = frames_u8.shape[0], self.engine.vit.frame_num_tokens, self.spec.hidden_size
        k = (B + T - 1) // T
        mine = frames_u8[r::T].contiguous()
        send = torch.zeros(k * rows, H, dtype=torch.bfloat16)
        if mine.shape[0]:
            send[:mine.shape[0] * rows] = self.engine.visual_embed(mine)
        recv = torch.zeros(T, k * rows, H, dtype=torch.bfloat16)
        check(lib().vlo_tp_allgather(self._g, _ptr(send), _ptr(recv), send.numel() * 2, None))
        return recv.view(T, k, rows, H).transpose(0, 1).reshape(T * k, rows, H)[:B].reshape(B * rows, H)

    def llm_step(self, embeds, want_all=True):
        return EmulTpGroup.llm_step(self, self._s, embeds, want_all)

    def bench_exchange(self, m, iters):
        return EmulTpGroup.bench_exchange(self, self._s, m, iters)

    def close(self):
        lib().vlo_tp_session_destroy(self._s)
        lib().vlo_tp_group_destroy(self._g)
        self.engine.close()


def unique_id():
    buf = C.create_string_buffer(128)
    check(lib().vlo_tp_unique_id(buf))
    return buf.raw


class EmulTpRank:
    """ONE rank of a one-process-per-GPU group (include/vlo.h: n_local = 1) with the peer-to-peer exchange and no RCCL.
    ``exchange`` takes this rank's 64-byte mailbox handle and returns every rank's, in rank order."""

    def __init__(self, spec, T, rank, weights, inv_freq, exchange, kv_pool_tokens=1024):
        self.spec, self.T, self.rank = spec, T, rank
        self.engine = EmulEngine(spec, kv_pool_tokens, rank, T).load_weights(weights, inv_freq)
        arr = (C.c_void_p * 1)(self.engine._h)
        g = C.c_void_p()
        check(lib().vlo_tp_group_create(arr, 1, None, C.byref(g)))
        self._g = g
        mine = C.create_string_buffer(64)
        check(lib().vlo_tp_p2p_export(g, mine))
        handles = exchange(mine.raw)
        check(lib().vlo_tp_p2p_enable(g, C.create_string_buffer(b"".join(handles), 64 * T)))
        h = C.c_void_p()
        check(lib().vlo_tp_session_create(g, 0, C.byref(h)))
        self._s = h

    p2p_status = EmulTpGroup.p2p_status

    def llm_step(self, embeds, want_all=True):
        return EmulTpGroup.llm_step(self, self._s, embeds, want_all)

    def close(self):
        lib().vlo_tp_session_destroy(self._s)
        lib().vlo_tp_group_destroy(self._g)
        self.engine.close()
