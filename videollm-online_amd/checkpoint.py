"""Checkpoint loader for the engine: base Llama safetensors + the reference's PEFT adapter + SigLIP.

What the reference does at load time (models/modeling_live.py:184-222): ``from_pretrained`` the base
Llama, then ``PeftModel.from_pretrained(model, resume_from_checkpoint)`` which leaves LoRA *un-merged*
(every target Linear computes ``W x + (alpha/r) B (A x)`` at run time, r=128, alpha=256 —
models/arguments_live.py:16-19) and restores the fully-trained ``connector`` from ``modules_to_save``;
``set_vision_inside`` then loads ``AutoModel.from_pretrained(siglip).vision_model``
(models/vision_live.py:54-57).

Here the adapter is merged once, in fp32 (``W' = W + (alpha/r) B A``), rounded to bf16 and handed to
``Engine.load_weight`` under the HF names the engine expects, so the streamed weight image needs no
run-time LoRA work.  Merged != bit-identical to the reference's un-merged bf16 arithmetic; parity of a
real checkpoint is therefore judged against the fp32 gold path.
SURVEY.md §8(f) rank 1."""
from __future__ import annotations

import json
import os
import re

import torch

LORA_TARGETS = re.compile(r"model.*(q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj)|lm_head$")  # arguments_live.py:16


class _TensorSource:
    """name -> tensor over one or many .safetensors files (lazy, one tensor in memory at a time)."""

    def __init__(self, path: str):
        from safetensors import safe_open
        self._open = safe_open
        self.files = {}
        if os.path.isdir(path):
            idx = [f for f in os.listdir(path) if f.endswith(".safetensors.index.json")]
            if idx:
                wm = json.load(open(os.path.join(path, idx[0])))["weight_map"]
                self.files = {k: os.path.join(path, v) for k, v in wm.items()}
            else:
                for f in sorted(os.listdir(path)):
                    if f.endswith(".safetensors"):
                        with safe_open(os.path.join(path, f), "pt") as h:
                            for k in h.keys():
                                self.files[k] = os.path.join(path, f)
        else:
            with safe_open(path, "pt") as h:
                self.files = {k: path for k in h.keys()}
        if not self.files:
            raise FileNotFoundError(f"no safetensors tensors under {path}")

    def keys(self):
        return self.files.keys()

    def __contains__(self, k):
        return k in self.files

    def get(self, k) -> torch.Tensor:
        with self._open(self.files[k], "pt") as h:
            return h.get_tensor(k)


def _adapter_maps(adapter_dir: str):
    """PEFT adapter_model.safetensors -> ({module: (A, B)}, {connector key: tensor}, scale)."""
    cfg = json.load(open(os.path.join(adapter_dir, "adapter_config.json")))
    scale = cfg.get("lora_alpha", 256) / cfg.get("r", 128)
    src = _TensorSource(os.path.join(adapter_dir, "adapter_model.safetensors"))
    lora, extra = {}, {}
    for k in src.keys():
        name = k
        for pre in ("base_model.model.",):
            if name.startswith(pre):
                name = name[len(pre):]
        name = name.replace(".modules_to_save.default", "").replace(".default", "")
        m = re.match(r"(.*)\.lora_([AB])\.weight$", name)
        if m:
            lora.setdefault(m.group(1), {})[m.group(2)] = k
        elif name.startswith("connector."):
            extra[name] = k
    return src, lora, extra, scale


FP8_E4M3_MAX = 448.0


def round_to_e4m3(x: torch.Tensor) -> torch.Tensor:
    """fp32 values in [-448, 448] rounded to the nearest OCP e4m3 value, ties to even, returned as fp32.  Written out with
    frexp / ldexp / round so that every device rounds alike (the first hardware run showed torch's own float -> float8
    conversion on the GPU disagreeing with its CPU conversion on ~6e-4 of the values, sub-normals included): a binade
    [2^e, 2^(e+1)) holds 8 steps of 2^(e-3); below 2^-6 the step is 2^-9."""
    _, ex = torch.frexp(x)                              # |x| = m * 2^ex, m in [0.5, 1)  ->  e = ex - 1
    step_exp = (ex - 4).clamp_min(-9)
    return torch.ldexp(torch.round(torch.ldexp(x, -step_exp)), step_exp).clamp_(-FP8_E4M3_MAX, FP8_E4M3_MAX)


def quantize_fp8_per_channel(W: torch.Tensor):
    """Symmetric per-output-channel quantisation of a Linear weight [out, in] to OCP fp8 e4m3 (BASELINE.json configs[4],
    SURVEY.md §8(f)-1 "fp8 + per-channel scales"): scale[n] = max|W[n]| * (1/448) (1 for an all-zero row),
    q[n] = e4m3_rne(W[n] / scale[n]).  Returns (q as torch.float8_e4m3fn [out, in], scale f32 [out]); W ~= q * scale[:, None].
    The engine streams q (one byte per weight) and applies scale[n] to the fp32 dot product."""
    Wf = W.float()
    amax = Wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax * (1.0 / FP8_E4M3_MAX), torch.ones_like(amax))
    q = round_to_e4m3((Wf / scale[:, None]).clamp_(-FP8_E4M3_MAX, FP8_E4M3_MAX))
    return q.to(torch.float8_e4m3fn), scale                # exact: q already holds e4m3 values


def merge_lora(W: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scale: float) -> torch.Tensor:
    """W [out,in] + scale * B [out,r] @ A [r,in], accumulated in fp32, rounded once to bf16."""
    return (W.float() + scale * (B.float() @ A.float())).to(torch.bfloat16)


def iter_llm_weights(base_dir: str, adapter_dir: str | None = None, device: str = "cpu"):
    """Yields (hf_name, bf16 tensor) for every Llama + connector weight, LoRA merged."""
    base = _TensorSource(base_dir)
    src = lora = extra = None
    scale = 1.0
    if adapter_dir:
        src, lora, extra, scale = _adapter_maps(adapter_dir)
        bad = [m for m in lora if not LORA_TARGETS.search(m)]
        if bad:
            raise ValueError(f"adapter targets modules outside the reference's lora_modules pattern: {bad[:4]}")
    seen = set()
    for k in base.keys():
        if "rotary_emb.inv_freq" in k:
            continue
        W = base.get(k).to(device)
        mod = k[:-len(".weight")] if k.endswith(".weight") else None
        if lora and mod in lora:
            ab = lora[mod]
            if set(ab) != {"A", "B"}:
                raise ValueError(f"incomplete LoRA pair for {mod}")
            W = merge_lora(W, src.get(ab["A"]).to(device), src.get(ab["B"]).to(device), scale)
            seen.add(mod)
        yield k, W.to(torch.bfloat16)
    if lora:
        missing = set(lora) - seen
        if missing:
            raise KeyError(f"adapter has LoRA weights for modules absent from the base checkpoint: {sorted(missing)[:4]}")
    if extra:
        for name, k in extra.items():
            yield name, src.get(k).to(device).to(torch.bfloat16)
    if "lm_head.weight" not in base and "model.embed_tokens.weight" in base:   # tied embeddings
        yield "lm_head.weight", base.get("model.embed_tokens.weight").to(device).to(torch.bfloat16)


def iter_vision_weights(siglip_dir: str, device: str = "cpu"):
    """Yields ('vision.<SiglipVisionModel key>', fp32 tensor) from a google/siglip-* checkpoint
    (keys 'vision_model.*'; the text tower and logit scale/bias are skipped)."""
    srcs = _TensorSource(siglip_dir)
    n = 0
    for k in srcs.keys():
        if not k.startswith("vision_model."):
            continue
        n += 1
        yield "vision." + k[len("vision_model."):], srcs.get(k).to(device).float()
    if n == 0:
        raise KeyError(f"no 'vision_model.*' tensors under {siglip_dir}")


def load_engine_weights(engine, base_dir: str, adapter_dir: str | None = None, siglip_dir: str | None = None,
                        device: str | None = None):
    """Stream a real checkpoint into an (un-finalized) Engine; merges LoRA on `device` (default: the GPU)."""
    dev = device or str(engine.device)
    for name, t in iter_llm_weights(base_dir, adapter_dir, dev):
        engine.load_weight(name, t)
    if siglip_dir:
        for name, t in iter_vision_weights(siglip_dir, dev):
            engine.load_weight(name, t)
    return engine
