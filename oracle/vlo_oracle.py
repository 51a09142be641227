"""CPU oracle for the videollm-online streaming hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / CPU baseline.  The product path
(``videollm-online_amd``) never imports this module and fails loudly when its
HIP library is missing.

What this restates (reference = /root/reference, HF = transformers 5.15.0, the
unpinned third-party dependency where the arithmetic actually lives —
/root/reference/README.md:54 installs ``transformers`` with no version):

  * ``_siglip_vision_encode``            models/vision_live.py:10-30
  * ``SiglipVisionModel.forward``        HF:models/siglip/modeling_siglip.py:576-619
      embeddings  :175-186, encoder layer :335-357, attention :273-307,
      MLP :318-322, MAP head :622-644
  * ``LiveMixin.visual_embed``           models/modeling_live.py:21-27
  * connector                            models/live_llama/modeling_live_llama.py:18-22
      (``GELUActivation(config.hidden_size)`` => use_gelu_python=True => exact
      erf GELU written as ``x * 0.5 * (1 + erf(x / sqrt(2)))``, HF:activations.py)
  * ``LlamaForCausalLM.forward`` with ``inputs_embeds`` + growing cache
      HF:models/llama/modeling_llama.py: RMSNorm :62-67, RoPE :113-160,
      attention :217-281, MLP :174-176, decoder layer :300-325, lm_head :477-480
      cache append = torch.cat            HF:cache_utils.py:127-151
      mask / sdpa glue                    HF:integrations/sdpa_attention.py:79-166
  * streaming sampler                    demo/inference.py:76-81
  * ``fast_greedy_generate``             models/modeling_live.py:173-182
  * ``LiveInfer`` state machine          demo/inference.py:40-123

Parity pinning: the reference ships NO golden vectors or known-answer tests for
this path.  This oracle is therefore pinned against
(1) the reference's own classes imported from /root/reference in the build
container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``, committed), and
(2) the HF modules it restates, instantiated with the same weights
(``tests/test_oracle_vs_hf.py``; transformers is present on the GPU box too).

Two arithmetic modes for the LLM:
  * ``dtype=torch.bfloat16``: same torch CPU ops, in the same order and with the
    same rounding points as the reference's CPU/sdpa bf16 path (the named parity
    target, BASELINE.json north_star).
  * ``dtype=torch.float32``: the "gold" path — the same bf16-valued weights upcast
    to fp32, all arithmetic fp32.  Used for the 3-way check
    ``err(engine, gold) <= err(reference-bf16, gold) * k + tol``.
"""
from __future__ import annotations

import collections
import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# model specs (SURVEY.md §8 dimension table)
# --------------------------------------------------------------------------------------
@dataclass
class LlmSpec:
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    vocab_size: int
    rope_theta: float = 10000.0
    rms_eps: float = 1e-5
    vision_hidden_size: int = 1024

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


@dataclass
class VitSpec:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_layers: int = 24
    num_heads: int = 16
    image_size: int = 384
    patch_size: int = 16
    ln_eps: float = 1e-6
    pooled: tuple = (3, 3)

    @property
    def grid(self):
        return self.image_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid * self.grid

    @property
    def frame_num_tokens(self):
        return 1 + self.pooled[0] * self.pooled[1]


LLM_SPECS = {
    "llama-3-8b": LlmSpec(4096, 14336, 32, 32, 8, 128256, 500000.0, 1e-5),
    "tinyllama-1.1b": LlmSpec(2048, 5632, 22, 32, 4, 32000, 10000.0, 1e-5),
    # reduced-depth true-width variants and toy shapes for per-commit tests
    "llama-3-8b-2l": LlmSpec(4096, 14336, 2, 32, 8, 128256, 500000.0, 1e-5),
    # one decoder layer at the Llama-3-70B width (BASELINE.json configs[4]); reduced vocabulary keeps the CPU oracle small
    "llama-3-70b-1l": LlmSpec(8192, 28672, 1, 64, 8, 8192, 500000.0, 1e-5),
    "tinyllama-2l": LlmSpec(2048, 5632, 2, 32, 4, 32000, 10000.0, 1e-5),
    "toy": LlmSpec(256, 704, 2, 4, 2, 1024, 10000.0, 1e-5, vision_hidden_size=128),
    "toy128": LlmSpec(512, 1408, 3, 4, 2, 2048, 500000.0, 1e-5, vision_hidden_size=128),
}

VIT_SPECS = {
    "siglip-l16-384": VitSpec(),
    "siglip-l16-384-2l": VitSpec(num_layers=2),
    "toy": VitSpec(hidden_size=128, intermediate_size=512, num_layers=2, num_heads=2,
                   image_size=96, patch_size=16, pooled=(3, 3)),
    # BASELINE.json configs[4]'s tower (SURVEY.md §8 dimension table; HF google/siglip-so400m-patch14-384): head dim 72, MLP 4304,
    # 27 x 27 patches of 14 pixels (the 384-pixel image is cropped to 378 by the strided conv).  The reference's build_live_vision
    # accepts SigLIP-L only; the oracle's tower code is shape-generic and pinned to HF's SiglipVisionModel by tests/test_oracle_vs_hf.py
    "siglip-so400m14-384": VitSpec(hidden_size=1152, intermediate_size=4304, num_layers=27, num_heads=16, image_size=384, patch_size=14),
    "siglip-so400m14-384-2l": VitSpec(hidden_size=1152, intermediate_size=4304, num_layers=2, num_heads=16, image_size=384, patch_size=14),
    # the same irregularities at toy size: head dim 72 (16 heads: the connector's GEMV has no plan for K = 576), MLP width 336 (not a
    # multiple of 64), 3 x 3 patches of 14 pixels (K = 588)
    "toy-hd72": VitSpec(hidden_size=1152, intermediate_size=336, num_layers=1, num_heads=16, image_size=42, patch_size=14, pooled=(3, 3)),
}


# --------------------------------------------------------------------------------------
# seeded weight init (no checkpoints exist on disk)
# --------------------------------------------------------------------------------------
def _randn(gen, *shape, std=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float32) * std


def init_llm_weights(spec: LlmSpec, seed: int = 0, dtype=torch.bfloat16) -> dict:
    """Seeded random weights under HF state-dict names.  Scales are chosen so that
    activations stay O(1) and logits have O(1) spread (HF's std=0.02 default gives
    near-uniform logits whose argmax is decided by rounding noise)."""
    g = torch.Generator().manual_seed(seed)
    H, I, V = spec.hidden_size, spec.intermediate_size, spec.vocab_size
    hd, nh, nkv = spec.head_dim, spec.num_heads, spec.num_kv_heads
    w = {}
    w["model.embed_tokens.weight"] = _randn(g, V, H, std=1.0)
    for i in range(spec.num_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = 1.0 + _randn(g, H, std=0.1)
        w[p + "self_attn.q_proj.weight"] = _randn(g, nh * hd, H, std=H ** -0.5)
        w[p + "self_attn.k_proj.weight"] = _randn(g, nkv * hd, H, std=H ** -0.5)
        w[p + "self_attn.v_proj.weight"] = _randn(g, nkv * hd, H, std=H ** -0.5)
        w[p + "self_attn.o_proj.weight"] = _randn(g, H, nh * hd, std=(nh * hd) ** -0.5)
        w[p + "post_attention_layernorm.weight"] = 1.0 + _randn(g, H, std=0.1)
        w[p + "mlp.gate_proj.weight"] = _randn(g, I, H, std=H ** -0.5)
        w[p + "mlp.up_proj.weight"] = _randn(g, I, H, std=H ** -0.5)
        w[p + "mlp.down_proj.weight"] = _randn(g, H, I, std=I ** -0.5)
    w["model.norm.weight"] = 1.0 + _randn(g, H, std=0.1)
    w["lm_head.weight"] = _randn(g, V, H, std=2.0 * H ** -0.5)
    Hv = spec.vision_hidden_size
    w["connector.0.weight"] = _randn(g, H, Hv, std=Hv ** -0.5)
    w["connector.0.bias"] = _randn(g, H, std=0.1)
    w["connector.2.weight"] = _randn(g, H, H, std=H ** -0.5)
    w["connector.2.bias"] = _randn(g, H, std=0.1)
    return {k: v.to(dtype) for k, v in w.items()}


def init_vit_weights(spec: VitSpec, seed: int = 1) -> dict:
    """Seeded fp32 weights under SiglipVisionModel (transformers 5.x) state-dict names,
    prefixed ``vision.``."""
    g = torch.Generator().manual_seed(seed)
    D, I, P = spec.hidden_size, spec.intermediate_size, spec.patch_size
    w = {}
    w["vision.embeddings.patch_embedding.weight"] = _randn(g, D, 3, P, P, std=(3 * P * P) ** -0.5)
    w["vision.embeddings.patch_embedding.bias"] = _randn(g, D, std=0.1)
    w["vision.embeddings.position_embedding.weight"] = _randn(g, spec.num_patches, D, std=0.5)

    def ln(prefix):
        w[prefix + ".weight"] = 1.0 + _randn(g, D, std=0.1)
        w[prefix + ".bias"] = _randn(g, D, std=0.1)

    def lin(prefix, out, inp):
        w[prefix + ".weight"] = _randn(g, out, inp, std=inp ** -0.5)
        w[prefix + ".bias"] = _randn(g, out, std=0.1)

    for i in range(spec.num_layers):
        p = f"vision.encoder.layers.{i}."
        ln(p + "layer_norm1")
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, D, D)
        ln(p + "layer_norm2")
        lin(p + "mlp.fc1", I, D)
        lin(p + "mlp.fc2", D, I)
    ln("vision.post_layernorm")
    w["vision.head.probe"] = _randn(g, 1, 1, D, std=1.0)
    w["vision.head.attention.in_proj_weight"] = _randn(g, 3 * D, D, std=D ** -0.5)
    w["vision.head.attention.in_proj_bias"] = _randn(g, 3 * D, std=0.1)
    lin("vision.head.attention.out_proj", D, D)
    ln("vision.head.layernorm")
    lin("vision.head.mlp.fc1", I, D)
    lin("vision.head.mlp.fc2", D, I)
    return w


def synthetic_frames(num_frames: int, resolution: int = 384, seed: int = 1234) -> torch.Tensor:
    """uint8 [T,3,R,R] NCHW: i.i.d. noise plus a low-frequency moving gradient so
    consecutive frames differ smoothly (format = data/utils.py:51-66)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(0, 256, (1, 3, resolution, resolution), generator=g, dtype=torch.int32)
    yy, xx = torch.meshgrid(torch.arange(resolution), torch.arange(resolution), indexing="ij")
    frames = []
    for t in range(num_frames):
        grad = ((xx + 3 * t) % resolution + (yy + 2 * t) % resolution) * 255 // (2 * resolution)
        noise = torch.randint(0, 64, (3, resolution, resolution), generator=g, dtype=torch.int32)
        f = (base[0] // 4 + grad[None].to(torch.int32) // 2 + noise).clamp(0, 255)
        frames.append(f.to(torch.uint8))
    return torch.stack(frames)


# --------------------------------------------------------------------------------------
# SigLIP vision tower + token selection
# --------------------------------------------------------------------------------------
def _mm_round(x, mm_dtype):
    return x if mm_dtype is None else x.to(mm_dtype).to(torch.float32)


def _linear(x, w, b, mm_dtype=None):
    """fp32 linear; with ``mm_dtype`` the inputs and the output are rounded to that
    dtype first, which is what CUDA autocast does around ``F.linear``
    (models/vision_live.py:13 — a no-op on CPU, active on the reference's GPU path)."""
    y = F.linear(_mm_round(x, mm_dtype), _mm_round(w, mm_dtype), None if b is None else _mm_round(b, mm_dtype))
    return _mm_round(y, mm_dtype)


def gelu_tanh(x):
    """HF:activations.py gelu_pytorch_tanh == F.gelu(approximate='tanh')."""
    return F.gelu(x, approximate="tanh")


def vit_forward(W: dict, spec: VitSpec, pixel_values: torch.Tensor, mm_dtype=None, taps: dict | None = None):
    """SiglipVisionModel.forward (HF:models/siglip/modeling_siglip.py:576-619).
    Returns (last_hidden_state [B,S,D] after post_layernorm, pooler_output [B,D])."""
    D, nh = spec.hidden_size, spec.num_heads
    hd = D // nh
    B = pixel_values.shape[0]
    # embeddings :175-186  (Conv2d k=s=patch, 'valid') + learned position embedding
    pe = F.conv2d(_mm_round(pixel_values, mm_dtype),
                  _mm_round(W["vision.embeddings.patch_embedding.weight"], mm_dtype),
                  _mm_round(W["vision.embeddings.patch_embedding.bias"], mm_dtype), stride=spec.patch_size)
    pe = _mm_round(pe, mm_dtype)
    h = pe.flatten(2).transpose(1, 2) + W["vision.embeddings.position_embedding.weight"][None]
    if taps is not None:
        taps["embed"] = h.clone()
    for i in range(spec.num_layers):
        p = f"vision.encoder.layers.{i}."
        # layer :335-357
        x = F.layer_norm(h, (D,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], spec.ln_eps)
        q = _linear(x, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"], mm_dtype)
        k = _linear(x, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"], mm_dtype)
        v = _linear(x, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"], mm_dtype)
        q = q.view(B, -1, nh, hd).transpose(1, 2)
        k = k.view(B, -1, nh, hd).transpose(1, 2)
        v = v.view(B, -1, nh, hd).transpose(1, 2)
        # attention :273-307 — non-causal, scale hd^-0.5, fp32 softmax
        s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        pr = _mm_round(torch.softmax(s, dim=-1, dtype=torch.float32), mm_dtype)
        a = _mm_round(torch.matmul(pr, v), mm_dtype).transpose(1, 2).reshape(B, -1, D)
        a = _linear(a, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], mm_dtype)
        h = h + a
        x = F.layer_norm(h, (D,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], spec.ln_eps)
        x = _linear(x, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], mm_dtype)
        x = _mm_round(gelu_tanh(x), mm_dtype)
        x = _linear(x, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], mm_dtype)
        h = h + x
        if taps is not None:
            taps[f"layer{i}"] = h.clone()
    last = F.layer_norm(h, (D,), W["vision.post_layernorm.weight"], W["vision.post_layernorm.bias"], spec.ln_eps)
    if taps is not None:
        taps["post_ln"] = last.clone()
    # MAP head :622-644 — nn.MultiheadAttention(probe as Q, tokens as K/V), packed in_proj
    wi, bi = W["vision.head.attention.in_proj_weight"], W["vision.head.attention.in_proj_bias"]
    probe = W["vision.head.probe"].expand(B, -1, -1)
    q = _linear(probe, wi[:D], bi[:D], mm_dtype).view(B, 1, nh, hd).transpose(1, 2)
    k = _linear(last, wi[D:2 * D], bi[D:2 * D], mm_dtype).view(B, -1, nh, hd).transpose(1, 2)
    v = _linear(last, wi[2 * D:], bi[2 * D:], mm_dtype).view(B, -1, nh, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    pr = _mm_round(torch.softmax(s, dim=-1, dtype=torch.float32), mm_dtype)
    a = _mm_round(torch.matmul(pr, v), mm_dtype).transpose(1, 2).reshape(B, 1, D)
    a = _linear(a, W["vision.head.attention.out_proj.weight"], W["vision.head.attention.out_proj.bias"], mm_dtype)
    r = a
    x = F.layer_norm(a, (D,), W["vision.head.layernorm.weight"], W["vision.head.layernorm.bias"], spec.ln_eps)
    x = _linear(x, W["vision.head.mlp.fc1.weight"], W["vision.head.mlp.fc1.bias"], mm_dtype)
    x = _mm_round(gelu_tanh(x), mm_dtype)
    x = _linear(x, W["vision.head.mlp.fc2.weight"], W["vision.head.mlp.fc2.bias"], mm_dtype)
    pooled = (r + x)[:, 0]
    if taps is not None:
        taps["pooler"] = pooled.clone()
    return last, pooled


def siglip_vision_encode(W: dict, spec: VitSpec, frames_u8: torch.Tensor, mm_dtype=None, taps=None):
    """_siglip_vision_encode (models/vision_live.py:10-30): uint8 [B,3,R,R] -> [B,1+9,D] fp32."""
    x = frames_u8 * 0.00392156862745098                       # :12  frames * rescale_factor (-> fp32)
    mean = torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    std = torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    x = (x - mean) / std                                      # torchvision normalize
    last, pooled = vit_forward(W, spec, x, mm_dtype, taps)
    s = int(math.sqrt(last.shape[1]))
    spatial = F.adaptive_avg_pool2d(last.reshape(last.shape[0], s, s, last.shape[-1]).permute(0, 3, 1, 2),
                                    spec.pooled).flatten(2, 3).permute(0, 2, 1)   # :17-23
    return torch.cat([pooled[:, None], spatial], dim=1)       # :30


def gelu_python(x):
    """HF GELUActivation with use_gelu_python=True (each op rounds in x's dtype)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def connector(W: dict, x: torch.Tensor):
    """models/live_llama/modeling_live_llama.py:18-22, in x's dtype."""
    x = F.linear(x, W["connector.0.weight"], W["connector.0.bias"])
    x = gelu_python(x)
    return F.linear(x, W["connector.2.weight"], W["connector.2.bias"])


# --------------------------------------------------------------------------------------
# Llama with growing cache
# --------------------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    """LlamaRMSNorm.forward HF:models/llama/modeling_llama.py:62-67."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_inv_freq(hd: int, theta: float):
    return 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))   # :98


def rope_cos_sin(positions: torch.Tensor, hd: int, theta: float, dtype):
    """LlamaRotaryEmbedding.forward :113-127 (fp32 angles, cast to model dtype)."""
    inv = rope_inv_freq(hd, theta).to(positions.device)
    freqs = (inv[None, :, None] @ positions[None, None, :].float()).transpose(1, 2)[0]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class KVCacheOracle:
    """DynamicCache restatement (HF:cache_utils.py:127-151): per-layer torch.cat growth."""

    def __init__(self, num_layers):
        self.k = [None] * num_layers
        self.v = [None] * num_layers

    def __len__(self):
        return 0 if self.k[0] is None else self.k[0].shape[1]

    def __bool__(self):       # LiveInfer only uses truthiness (demo/inference.py:61,98)
        return True           # a DynamicCache object is always truthy once it exists

    def update(self, i, k, v):
        self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=1)
        self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=1)
        return self.k[i], self.v[i]


FP8_STREAMED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "gate_proj.weight", "up_proj.weight",
                "down_proj.weight", "lm_head.weight")


def e4m3_rne(x: torch.Tensor) -> torch.Tensor:
    """fp32 values (|x| <= 448) rounded to the nearest OCP e4m3 value, ties to even: 8 steps per binade, 2^-9 below 2^-6"""
    _, ex = torch.frexp(x)
    step_exp = (ex - 4).clamp_min(-9)
    q = torch.ldexp(torch.round(torch.ldexp(x, -step_exp)), step_exp).clamp(-448.0, 448.0)
    return q


def fp8_quantize_rows(x: torch.Tensor):
    """The W8A8 prefill rule of an fp8 engine with prefill_act_dtype = 1 (include/vlo.h; csrc/prefill.h) — the weights' rule applied to an
    activation row: scale[m] = max|x[m]| * (1/448) (1 for a zero row), q[m] = e4m3_rne(clamp(x[m] / scale[m], -448, 448)).
    Returns (q as fp32 e4m3 values, scale fp32 [m, 1]).  The reference has no fp8 path (config 5 exceeds it)."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    return e4m3_rne((xf / s).clamp(-448.0, 448.0)), s


def fp8_dequantized_weights(weights: dict) -> dict:
    """BASELINE.json configs[4] ("fp8 MFMA weights"): the streamed Llama projections replaced by what an fp8 e4m3 store with one
    scale per output channel holds — W' = e4m3_rne(W / s) * s, s[n] = max|W[n]| * (1/448) — as fp32 tensors; everything else
    untouched.  The reference has no fp8 path (config 5 exceeds the reference); the parity target of the fp8
    engine is the reference arithmetic run on these weights: LlamaOracle keeps fp32 weights in fp32 and rounds the Linear's
    OUTPUT to the activation dtype, like a bf16 Linear does."""
    out = {}
    for k, v in weights.items():
        if k.endswith(FP8_STREAMED) and not k.startswith(("vision.", "connector.")):
            Wf = v.float()
            amax = Wf.abs().amax(dim=1)
            s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
            q = e4m3_rne((Wf / s[:, None]).clamp(-448.0, 448.0))
            assert torch.equal(q, q.to(torch.float8_e4m3fn).float())     # every q is an e4m3 value
            out[k] = q * s[:, None]
        else:
            out[k] = v
    return out


def _llm_linear(x, W, act_fp8=False, k_groups=1):
    """F.linear; a weight kept in fp32 next to lower-precision activations (fp8_dequantized_weights) is applied in fp32 and
    the output rounded to the activation dtype.  ``act_fp8``: the rows of x go through fp8_quantize_rows first (W8A8 prefill):
    y = (q @ W^T) * scale[m], accumulated in fp32, rounded to the activation dtype.  ``k_groups`` > 1: a ROW-sharded projection of a
    tensor-parallel group (o_proj, down_proj: rank r holds columns [r K/T, (r+1) K/T) of x) — every rank quantises ITS slice of the row
    with its own scale, the fp32 partial products are summed across the ranks."""
    if act_fp8:
        Kg = x.shape[-1] // k_groups
        y = None
        for gi in range(k_groups):
            q, s = fp8_quantize_rows(x[..., gi * Kg:(gi + 1) * Kg])
            part = F.linear(q, W[:, gi * Kg:(gi + 1) * Kg].float()) * s
            y = part if y is None else y + part
        return y.to(x.dtype)
    if W.dtype == x.dtype:
        return F.linear(x, W)
    return F.linear(x.float(), W.float()).to(x.dtype)


class LlamaOracle:
    def __init__(self, spec: LlmSpec, weights: dict, dtype=torch.bfloat16, keep_fp32=()):
        """``keep_fp32``: names kept in fp32 whatever ``dtype`` is (the dequantised fp8 projections)"""
        self.spec, self.dtype = spec, dtype
        done = {}                                   # tensors aliased under several names are converted (and held) once
        self.W = {}
        for k, v in weights.items():
            if id(v) not in done:
                done[id(v)] = v.float() if k in keep_fp32 else v.to(dtype)
            self.W[k] = done[id(v)]

    def new_cache(self):
        return KVCacheOracle(self.spec.num_layers)

    def embed(self, ids: torch.Tensor):
        return F.embedding(ids, self.W["model.embed_tokens.weight"])

    @torch.no_grad()
    def forward(self, inputs_embeds: torch.Tensor, cache: KVCacheOracle | None, taps: dict | None = None, logits_from: int = 0, act_fp8: bool = False,
                act_fp8_tp: int = 1):
        """inputs_embeds [n,H] -> (logits [n,V], cache).  LlamaModel.forward :367-417.
        ``act_fp8``: every decoder-layer projection of THIS call (q/k/v/o, gate/up/down; not the lm_head) quantises its input rows to e4m3
        (fp8_quantize_rows) — what an fp8 engine with prefill_act_dtype = 1 does on a long input; not a reference feature.  ``act_fp8_tp`` = T: the
        same under T-way tensor parallelism (the row-sharded o_proj / down_proj quantise per rank: _llm_linear's k_groups).
        ``logits_from`` (test bookkeeping for long cache fills): lm_head only on rows [logits_from, n) — the rows are
        independent, so the rows that are produced equal HF's; n means "no logits" (returns an empty [0,V])."""
        s, W = self.spec, self.W
        if cache is None:
            cache = self.new_cache()
        n, Lc = inputs_embeds.shape[0], len(cache)
        nh, nkv, hd = s.num_heads, s.num_kv_heads, s.head_dim
        dev = inputs_embeds.device              # CPU everywhere except the long-context fills of tests/test_gpu_long.py (same code, torch on the GPU)
        pos = torch.arange(Lc, Lc + n, device=dev)                       # :386-389
        cos, sin = rope_cos_sin(pos, hd, s.rope_theta, self.dtype)
        h = inputs_embeds.to(self.dtype)
        # mask (HF:masking_utils.py): none for n==1; is_causal for empty cache; bool mask otherwise
        mask = None
        if n > 1 and Lc > 0:
            mask = (torch.arange(Lc + n, device=dev)[None, :] <= (Lc + torch.arange(n, device=dev))[:, None])
        for i in range(s.num_layers):
            p = f"model.layers.{i}."
            x = rmsnorm(h, W[p + "input_layernorm.weight"], s.rms_eps)
            q = _llm_linear(x, W[p + "self_attn.q_proj.weight"], act_fp8).view(n, nh, hd).transpose(0, 1)
            k = _llm_linear(x, W[p + "self_attn.k_proj.weight"], act_fp8).view(n, nkv, hd).transpose(0, 1)
            v = _llm_linear(x, W[p + "self_attn.v_proj.weight"], act_fp8).view(n, nkv, hd).transpose(0, 1)
            q = (q * cos) + (rotate_half(q) * sin)                       # :157-158
            k = (k * cos) + (rotate_half(k) * sin)
            K, V = cache.update(i, k, v)
            if taps is not None and i in taps.get("_layers", ()):
                taps[f"k{i}"], taps[f"v{i}"], taps[f"q{i}"] = k.clone(), v.clone(), q.clone()
            rep = nh // nkv
            Kr = K[:, None].expand(nkv, rep, Lc + n, hd).reshape(nh, Lc + n, hd)
            Vr = V[:, None].expand(nkv, rep, Lc + n, hd).reshape(nh, Lc + n, hd)
            a = F.scaled_dot_product_attention(q[None], Kr[None], Vr[None], attn_mask=mask,
                                               is_causal=(n > 1 and Lc == 0), scale=hd ** -0.5)[0]
            a = a.transpose(0, 1).reshape(n, nh * hd)
            if taps is not None and i in taps.get("_layers", ()):
                taps[f"attn{i}"] = a.clone()
            h = h + _llm_linear(a, W[p + "self_attn.o_proj.weight"], act_fp8, act_fp8_tp)        # :317
            x = rmsnorm(h, W[p + "post_attention_layernorm.weight"], s.rms_eps)
            x = _llm_linear(F.silu(_llm_linear(x, W[p + "mlp.gate_proj.weight"], act_fp8)) * _llm_linear(x, W[p + "mlp.up_proj.weight"], act_fp8),
                         W[p + "mlp.down_proj.weight"], act_fp8, act_fp8_tp)         # :174-176
            h = h + x                                                    # :323
            if taps is not None and i in taps.get("_layers", ()):
                taps[f"h{i}"] = h.clone()
        h = rmsnorm(h[logits_from:], W["model.norm.weight"], s.rms_eps)
        logits = _llm_linear(h, W["lm_head.weight"])                        # :477-480 (all rows, as HF, unless logits_from > 0); never act_fp8
        return logits, cache

    @torch.no_grad()
    def visual_embed(self, vit_W: dict, vit_spec: VitSpec, frames_u8: torch.Tensor, mm_dtype=None):
        """LiveMixin.visual_embed models/modeling_live.py:21-27 (autocast is a no-op on CPU)."""
        f = siglip_vision_encode(vit_W, vit_spec, frames_u8, mm_dtype)
        f = connector(self.W, f.to(self.dtype))
        return f.view(-1, f.shape[-1])


def stream_sample(logits_last: torch.Tensor, interval_id: int, threshold: float):
    """demo/inference.py:76-81 on the last-row logits [V] (model dtype).
    Returns (token id, p_interval before zeroing)."""
    score = logits_last.softmax(dim=-1)
    p_int = float(score[interval_id])
    if score[interval_id] < threshold:
        score[interval_id] = 0
    return int(score.argmax(dim=-1)), p_int


def top2_margin(logits_last: torch.Tensor, exclude: int | None = None, chosen: int | None = None):
    """(top-1 minus top-2 logit, runner-up id) — test bookkeeping for near-tie detection.  With an exact
    bf16 tie torch.argmax keeps the lower index while topk may order the pair either way, so the
    runner-up is reported as "the top-2 entry that is not the chosen token"."""
    v = logits_last.float().clone()
    if exclude is not None:
        v[exclude] = -float("inf")
    t = v.topk(2)
    i0, i1 = int(t.indices[0]), int(t.indices[1])
    if chosen is None:
        chosen = int(v.argmax())
    return float(t.values[0] - t.values[1]), (i1 if chosen == i0 else i0)


@torch.no_grad()
def fast_greedy_generate(model: LlamaOracle, inputs_embeds, cache, eos_token_id: int, max_new: int = 100, margins=None):
    """models/modeling_live.py:173-182.  Returns (list of ids incl. a terminating EOS, cache)."""
    out = []
    for _ in range(max_new):
        logits, cache = model.forward(inputs_embeds, cache)
        tok = int(logits[-1].argmax(dim=-1))
        if margins is not None:
            margins.append(top2_margin(logits[-1], None, tok))
        out.append(tok)
        if tok == eos_token_id:
            break
        inputs_embeds = model.embed(torch.tensor([tok]))
    return out, cache


# --------------------------------------------------------------------------------------
# LiveInfer state machine (demo/inference.py:12-123), device- and tokenizer-free
# --------------------------------------------------------------------------------------
@dataclass
class StreamTokens:
    """Token-id lists the reference derives from the tokenizer's chat template
    (demo/inference.py:33-35,42; models/tokenization_live.py:27-65); injected here
    because no tokenizer files exist offline."""
    start_ids: list
    stream_prompt_ids: list          # "\n[" after a response
    stream_generation_ids: list      # "]\nAssistant:"
    eos_token_id: int
    interval_id: int
    query_ids: dict = field(default_factory=dict)   # query string -> ids of "]\nUser: {q}\nAssistant:"


class LiveInferOracle:
    def __init__(self, llm: LlamaOracle, vit_W, vit_spec: VitSpec, tokens: StreamTokens, frame_fps=2,
                 threshold=0.725, max_new=100, mm_dtype=None, schedule=None):
        self.llm, self.vit_W, self.vit_spec, self.tok = llm, vit_W, vit_spec, tokens
        self.frame_fps, self.threshold, self.max_new, self.mm_dtype = frame_fps, threshold, max_new, mm_dtype
        self.frame_num_tokens = vit_spec.frame_num_tokens
        # optional deterministic speech schedule for throughput runs:
        # schedule(frame_idx) -> None (free-running) | (speak: bool, num_tokens: int)
        self.schedule = schedule
        self.trace = []
        self.reset()

    def reset(self):                                                   # :84-91
        self.query_queue = collections.deque()
        self.frame_embeds_queue = collections.deque()
        self.video_time = 0
        self.last_frame_idx = -1
        self.video_tensor = None
        self.last_ids = []
        self.past_key_values = None
        self._frames_done = 0

    def load_video(self, frames_u8):                                   # :111-115
        self.video_tensor = frames_u8
        self.num_video_frames = frames_u8.shape[0]

    def input_query_stream(self, query, video_time=None):              # :93-100
        self.query_queue.append((self.video_time if video_time is None else video_time, query))

    def input_video_stream(self, video_time):                          # :102-109
        frame_idx = int(video_time * self.frame_fps)
        if frame_idx > self.last_frame_idx:
            ranger = range(self.last_frame_idx + 1, frame_idx + 1)
            emb = self.llm.visual_embed(self.vit_W, self.vit_spec, self.video_tensor[ranger.start:ranger.stop],
                                        self.mm_dtype).split(self.frame_num_tokens)
            self.frame_embeds_queue.extend([(r / self.frame_fps, e) for r, e in zip(ranger, emb)])
        self.last_frame_idx = frame_idx
        self.video_time = video_time

    def _call_for_response(self, video_time, query):                   # :40-52
        if query is not None:
            self.last_ids = list(self.tok.query_ids[query])
        else:
            # reference asserts last_ids == 933 (Llama-3 tokenizer specific hack, :44); any
            # non-interval token is the trigger (rule 3, :80-81) — SURVEY.md §8c
            self.last_ids = list(self.tok.stream_generation_ids)
        emb = self.llm.embed(torch.tensor(self.last_ids))
        max_new = self.max_new
        forced = None
        if self.schedule is not None:
            forced = self.schedule(self._frames_done - 1)
        if forced is not None:
            out, self.past_key_values = forced_generate(self.llm, emb, self.past_key_values, forced[1],
                                                        self.tok.eos_token_id)
            margins = None
        else:
            margins = []
            out, self.past_key_values = fast_greedy_generate(self.llm, emb, self.past_key_values,
                                                             self.tok.eos_token_id, max_new, margins)
        self.last_ids = out[-1:]
        self.trace.append(("response", video_time, query, list(out), margins))
        return query, out

    def _call_for_streaming(self):                                     # :54-82
        while self.frame_embeds_queue:
            if self.query_queue and self.frame_embeds_queue[0][0] > self.query_queue[0][0]:      # rule 1
                return self.query_queue.popleft()
            video_time, frame_embeds = self.frame_embeds_queue.popleft()
            if not self.past_key_values:
                self.last_ids = list(self.tok.start_ids)
            elif self.last_ids == [self.tok.eos_token_id]:
                self.last_ids = self.last_ids + list(self.tok.stream_prompt_ids)
            inputs = torch.cat([self.llm.embed(torch.tensor(self.last_ids, dtype=torch.long)).view(-1, self.llm.spec.hidden_size),
                                frame_embeds.view(-1, self.llm.spec.hidden_size)], dim=0)
            logits, self.past_key_values = self.llm.forward(inputs, self.past_key_values)
            self._frames_done += 1
            if self.query_queue and video_time >= self.query_queue[0][0]:                        # rule 2
                return self.query_queue.popleft()
            zeroed = float(logits[-1].softmax(dim=-1)[self.tok.interval_id]) < self.threshold
            tok, p_int = stream_sample(logits[-1], self.tok.interval_id, self.threshold)        # rule 3
            margin = top2_margin(logits[-1], self.tok.interval_id if zeroed else None, tok)
            forced = self.schedule(self._frames_done - 1) if self.schedule is not None else None
            if forced is not None:
                tok = self.tok.stream_generation_ids[0] if forced[0] else self.tok.interval_id
                if tok == self.tok.interval_id and forced[0]:
                    raise ValueError("schedule needs stream_generation_ids[0] != interval_id")
            self.last_ids = [tok]
            self.trace.append(("frame", video_time, tok, p_int, len(self.past_key_values), margin))
            if tok != self.tok.interval_id:
                return video_time, None
        return None, None

    def __call__(self):                                                # :117-123
        if not self.frame_embeds_queue:
            raise RuntimeError("no frame queued (the reference would busy-wait here, :118)")
        video_time, query = self._call_for_streaming()
        response = None
        if video_time is not None:
            query, response = self._call_for_response(video_time, query)
        return query, response


@torch.no_grad()
def forced_generate(model: LlamaOracle, inputs_embeds, cache, num_tokens: int, eos_token_id: int):
    """Scheduled-mode response for deterministic throughput runs: the greedy argmax is
    computed every step (same work as fast_greedy_generate) but EOS is suppressed until
    exactly ``num_tokens`` tokens have been produced; the last one is forced to EOS."""
    out = []
    for i in range(num_tokens):
        logits, cache = model.forward(inputs_embeds, cache)
        tok = int(logits[-1].argmax(dim=-1))
        if i == num_tokens - 1:
            tok = eos_token_id
        elif tok == eos_token_id:
            tok = (eos_token_id + 1) % model.spec.vocab_size
        out.append(tok)
        if i < num_tokens - 1:
            inputs_embeds = model.embed(torch.tensor([tok]))
    return out, cache


def default_tokens(spec: LlmSpec, seed: int = 7, n_start: int = 35) -> StreamTokens:
    """Fixed synthetic id lists with the reference's lengths."""
    g = torch.Generator().manual_seed(seed)
    V = spec.vocab_size
    eos = min(128009, V - 2) if V > 2000 else V - 2
    interval = 11 if V != 32000 else 29892

    def rnd(k):
        ids = torch.randint(12, V - 4, (k,), generator=g).tolist()
        return [i if i not in (eos, interval) else i + 1 for i in ids]

    start = [min(128000, V - 3)] + rnd(n_start - 1)
    return StreamTokens(start_ids=start, stream_prompt_ids=rnd(2), stream_generation_ids=rnd(4),
                        eos_token_id=eos, interval_id=interval,
                        query_ids={"Please narrate the video in real time.": rnd(12)})


# --------------------------------------------------------------------------------------
# Teacher-forced evaluation: joint_embed / stream_evaluate / trim_past_key_values
# (models/modeling_live.py:29-42, 44-168, 170-171) — SURVEY.md §8(f) rank 4
# --------------------------------------------------------------------------------------
def joint_embed(model: LlamaOracle, input_ids: torch.Tensor, frame_embeds: torch.Tensor | None, v_placeholder_id: int):
    """models/modeling_live.py:29-42 with ``frame_embeds`` = visual_embed(frames) already computed
    ([num_frames * frame_num_tokens, H]).  Ids are clamped to the vocabulary before the lookup (:38);
    rows holding the placeholder are then overwritten in order (:39-41)."""
    ids = input_ids.view(-1)
    x = model.embed(ids.clamp(max=model.spec.vocab_size - 1)).clone()
    at_v = ids == v_placeholder_id
    if at_v.any():
        if frame_embeds is None or int(at_v.sum()) != frame_embeds.shape[0]:
            raise ValueError(f"{int(at_v.sum())} placeholder positions but "
                             f"{0 if frame_embeds is None else frame_embeds.shape[0]} frame-token embeddings")
        x[at_v] = frame_embeds.to(x.dtype)
    return x


def cache_prefix(cache: KVCacheOracle, stop: int) -> KVCacheOracle:
    """trim_past_key_values(past, 0, stop) (models/modeling_live.py:170-171): a NEW cache holding the
    first ``stop`` positions of every layer; the source cache is left untouched (the reference slices
    views and DynamicCache.update concatenates into fresh tensors)."""
    out = KVCacheOracle(len(cache.k))
    for i in range(len(cache.k)):
        out.k[i], out.v[i] = cache.k[i][:, :stop], cache.v[i][:, :stop]
    return out


def synthetic_eval_sample(spec: LlmSpec, toks: StreamTokens, turns, frame_num_tokens: int = 10, seed: int = 11,
                          v_placeholder_id: int | None = None):
    """A multi-turn teacher-forced sample with the token layout of the reference's chat template
    (models/tokenization_live.py:27-65, learn ranges :84-105, label shift :137-146):

        start_ids   { <v>*10 (, <v>*10)*  ]\\nAssistant: <response> EOS  \\n[ }*

    ``turns`` = [(num_frames, response_len, learn_frames)] per turn.  Labels are next-token targets on
    the last <v> of each learnt frame (the interval, or ``]`` for the frame the reply follows), on the
    ``]\\nAssistant:`` tokens and on the response through EOS; everything else is -100.  Placeholder
    targets are replaced by EOS as the reference's collator does (:148-150).
    Returns (input_ids [n], labels [n], total_frames)."""
    g = torch.Generator().manual_seed(seed)
    V = spec.vocab_size
    v_id = V if v_placeholder_id is None else v_placeholder_id
    ids, learn = list(toks.start_ids), [False] * len(toks.start_ids)
    total = 0
    for t, (nf, resp_len, learn_frames) in enumerate(turns):
        if t > 0:
            ids += list(toks.stream_prompt_ids); learn += [False] * len(toks.stream_prompt_ids)
        for f in range(nf):
            if f > 0:
                ids.append(toks.interval_id); learn.append(False)
            ids += [v_id] * frame_num_tokens
            learn += [False] * (frame_num_tokens - 1) + [f < learn_frames or f == nf - 1]
        total += nf
        gen = list(toks.stream_generation_ids)
        resp = torch.randint(12, V - 4, (resp_len,), generator=g).tolist()
        resp = [i if i not in (toks.eos_token_id, toks.interval_id) else i + 1 for i in resp]
        ids += gen + resp + [toks.eos_token_id]
        learn += [True] * (len(gen) + resp_len) + [False]
    input_ids = torch.tensor(ids, dtype=torch.long)
    labels = torch.full_like(input_ids, -100)
    m = torch.tensor(learn)
    labels[m] = torch.roll(input_ids, -1)[m]
    labels[labels >= V] = toks.eos_token_id
    return input_ids, labels, total


@torch.no_grad()
def stream_evaluate(model: LlamaOracle, input_ids: torch.Tensor, labels: torch.Tensor, frame_embeds: torch.Tensor, *,
                    v_placeholder_id: int, interval_id: int | None, eos_token_id: int, frame_num_tokens: int = 10,
                    threshold: float = 0.0, ignore_token_id: int = -100, detail: dict | None = None):
    """models/modeling_live.py:44-168.  Returns float32 [lm_ppl, frame_diff, fluency, lm_correctness].

    One full-sequence forward (:67), then per dialogue turn (EOS-delimited, :62-63):
      * LM perplexity / leading-correct fraction over learnt non-placeholder positions (:93-102);
      * time-to-reply error in frames over learnt placeholder positions (:105-149): first position
        whose softmax argmax is not the interval token; if there is none, the KV prefix up to the
        last streamed frame is continued with the next turn's frames to see how LATE the reply comes;
      * fluency (:152-161).
    Reference quirks kept on purpose: thresholding zeroes the whole score row (:110-111, so its argmax
    becomes id 0); a turn without learnt tokens does not advance the frame counter (:83-84 vs :163)."""
    ids, lab = input_ids.view(-1), labels.view(-1)
    fnt = frame_num_tokens
    sil = interval_id if interval_id is not None else eos_token_id           # :72-73
    logits, cache = model.forward(joint_embed(model, ids, frame_embeds, v_placeholder_id), None)
    stops = ((ids == eos_token_id).nonzero().view(-1) + 1).tolist()
    starts = [0] + stops[:-1]

    margins = []                            # test bookkeeping: how far each decision is from flipping

    def replies(rows, used=None):           # rows of logits -> bool per row: "the model speaks here"
        sc = rows.softmax(dim=-1)
        p_sil = sc[:, sil].float().clone()
        if threshold > 0:
            sc[sc[:, sil] < threshold] = 0
        out = sc.argmax(dim=-1) != sil
        if detail is not None:
            z = rows.float()
            others = z.clone()
            others[:, sil] = -float("inf")
            gap = (z[:, sil] - others.max(dim=-1).values).abs()
            if threshold > 0:
                # a row speaks if p_sil < threshold OR another token beats the interval
                gap = torch.where(p_sil < threshold, (threshold - p_sil) * 8, torch.minimum(gap, (p_sil - threshold) * 8))
            sel = torch.ones_like(out) if used is None else used
            n_dec = int(out[sel].nonzero()[0, 0]) + 1 if out[sel].any() else int(sel.sum())
            margins.extend(gap[sel][:n_dec].tolist())
        return out

    ppls, diffs, fluencies, corrects = [], [], [], []
    frames_seen = 0
    turn_log = []
    for r, (a, b) in enumerate(zip(starts, stops)):
        L = lab[a:b]
        learnt = L != ignore_token_id
        if not learnt.any():
            continue
        Z, I = logits[a:b], ids[a:b]
        at_v = I == v_placeholder_id
        n_frames = int(at_v.sum()) // fnt
        on_stream = at_v & learnt
        on_text = learnt & ~on_stream
        n_ok = diff = branch = None
        if on_text.any():
            zt, lt = Z[on_text], L[on_text]
            ppls.append(F.cross_entropy(zt, lt).exp())
            wrong = zt.argmax(dim=-1) != lt
            n_ok = int(wrong.nonzero()[0, 0]) if wrong.any() else int((~wrong).sum())
            if detail is not None:
                t2 = zt.float().topk(2).values
                at_label = zt.float().gather(1, lt[:, None])[:, 0]
                m = torch.where(wrong, t2[:, 0] - at_label, t2[:, 0] - t2[:, 1])
                margins.extend(m[:n_ok + 1].tolist())
            corrects.append(torch.tensor(n_ok) / lt.numel())
        if on_stream.any():
            speak = replies(Z, on_stream)[on_stream]
            branch = "hit"
            n_stream = int(on_stream.sum())
            if speak.any():
                diff = n_stream - int(speak.nonzero()[0, 0]) - 1
            else:
                keep = a + int(on_stream.nonzero()[-1, 0]) + 1
                if r == len(starts) - 1:
                    diff, branch = 0, "late-last-turn"
                else:
                    nxt = int((ids[starts[r + 1]:stops[r + 1]] == v_placeholder_id).sum()) // fnt
                    k = min(nxt, n_frames - 1)
                    if k == 0:
                        diff, branch = 0, "late-no-room"
                    else:
                        f0 = frames_seen + n_frames
                        unit = ([interval_id] if interval_id is not None else []) + [v_placeholder_id] * fnt
                        more = torch.tensor(unit * k, dtype=torch.long)
                        z2, _ = model.forward(joint_embed(model, more, frame_embeds[f0 * fnt:(f0 + k) * fnt], v_placeholder_id),
                                              cache_prefix(cache, keep))
                        late = replies(z2[len(unit) - 1::len(unit)])
                        diff = -(int(late.nonzero()[0, 0]) + 1) if late.any() else -k
                        branch = "late-hit" if late.any() else "late-none"
            diffs.append(abs(diff))
        if on_text.any() and on_stream.any():
            n_v = int(on_stream.sum())
            denom = int(on_text.sum()) + n_v
            if diff == 0:
                fluencies.append((n_v + n_ok) / denom)
            elif diff > 0:
                fluencies.append((n_v - diff) / denom)
            else:
                fluencies.append((n_v - 1) / denom)
        turn_log.append((r, n_ok, diff, branch))
        frames_seen += n_frames
    if detail is not None:
        detail["turns"], detail["logits"], detail["cache"], detail["margins"] = turn_log, logits, cache, margins
    lm_ppl = torch.stack(ppls).mean().float() if ppls else torch.tensor(1.0)
    frame_diff = torch.tensor(diffs, dtype=torch.float32).mean() if diffs else torch.tensor(0.0)
    fluency = torch.tensor(fluencies, dtype=torch.float32).mean() if fluencies else torch.tensor(1.0)
    correct = torch.stack(corrects).float().mean() if corrects else torch.tensor(1.0)
    return torch.stack([lm_ppl, frame_diff, fluency, correct])


def eval_case_from_golden(g, c: int, spec: LlmSpec, base_seed: int = 3):
    """Rebuild case ``c`` of tests/golden/eval_toy128.npz: (weights with the patched interval row, input_ids, labels,
    vision features [T, 10, vision_hidden], threshold).  The recipe lives in oracle/make_golden.py::eval_case."""
    w = init_llm_weights(spec, seed=base_seed, dtype=torch.bfloat16)
    toks = default_tokens(spec, seed=7, n_start=19)
    w["lm_head.weight"] = w["lm_head.weight"].clone()
    w["lm_head.weight"][toks.interval_id] = torch.from_numpy(g[f"c{c}_interval_row"]).to(torch.bfloat16)
    ids, labels = torch.from_numpy(g[f"c{c}_ids"]), torch.from_numpy(g[f"c{c}_labels"])
    T = int((ids == spec.vocab_size).sum()) // 10
    gen = torch.Generator().manual_seed(int(g[f"c{c}_feat_seed"]))
    feats = torch.randn(T, 10, spec.vision_hidden_size, generator=gen)
    return w, toks, ids, labels, feats, float(g[f"c{c}_threshold"])
