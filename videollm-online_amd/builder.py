"""``build_model_and_tokenizer`` — the reference's model factory (models/__init__.py:4 -> models/live_llama/
modeling_live_llama.py::build_live_llama -> models/modeling_live.py:184-222 ``build_live``) on the HIP engine, for the
inference branch (``is_training=False``): same keyword arguments, returns ``(model, tokenizer)`` with ``model`` a
``LiveModel`` whose weights already live in HBM.  INTEGRATION.md section 3 shows the two-line edit of demo/inference.py.

What the reference does there and what happens here:
  * ``from_pretrained(llm_pretrained)`` + ``PeftModel.from_pretrained(model, resume_from_checkpoint)`` (LoRA left un-merged)
    -> the base safetensors are streamed into the engine with the adapter MERGED (checkpoint.py);
  * ``set_vision_inside()`` (models/vision_live.py:54-57) -> the SigLIP vision tower is loaded into the same engine;
  * ``build_live_tokenizer_and_update_config`` (models/tokenization_live.py:104-117) stays the reference's own function (the
    tokenizer is HF and out of scope): it is imported from the reference tree when that is importable, or
    passed in as ``tokenizer_builder``; it fills v_placeholder_id / frame_token_interval_id / eos_token_id exactly as it does
    for the reference's config object.
Paths must be local directories (no hub access is assumed)."""
import json
import os

from .checkpoint import load_engine_weights
from .engine import Engine, EngineConfig
from .modeling_live import LiveModel


class LiveConfig:
    """The slice of LiveConfigMixin (models/configuration_live.py:4-21) the tokenizer builder reads and updates."""

    def __init__(self, **kw):
        self.v_placeholder = "<v>"
        self.frame_token_interval = ","
        self.frame_num_tokens = 10
        self.frame_token_cls = True
        self.frame_token_pooled = [3, 3]
        self.v_placeholder_id = None
        self.frame_token_interval_id = None
        self.eos_token_id = None
        self.update(kw)

    def update(self, d):
        for k, v in d.items():
            setattr(self, k, v)


def _vit_config(vision_pretrained: str) -> dict:
    cfg = json.load(open(os.path.join(vision_pretrained, "config.json")))
    v = cfg.get("vision_config", cfg)
    D, nh = v.get("hidden_size", 1024), v.get("num_attention_heads", 16)
    hd = D // nh
    if hd * nh != D or D % 64 or D > 2048 or not (hd == 64 or (68 <= hd <= 80 and hd % 4 == 0)):
        # the reference itself only accepts SigLIP-L and two CLIPs (models/vision_live.py:56-60); the engine also runs SigLIP-so400m/14
        # (head dim 72, BASELINE.json configs[4]) — the shape rule is csrc/vit.hip::vit_finalize's
        raise ValueError(f"unsupported vision tower {vision_pretrained!r}: head_dim {hd} (64 = SigLIP-L/16-384, or 68..80 = SigLIP-so400m/14), "
                         f"hidden {D} (a multiple of 64, at most 2048)")
    return dict(hidden_size=v.get("hidden_size", 1024), intermediate_size=v.get("intermediate_size", 4096),
                num_layers=v.get("num_hidden_layers", 24), num_heads=v.get("num_attention_heads", 16),
                image_size=v.get("image_size", 384), patch_size=v.get("patch_size", 16), ln_eps=v.get("layer_norm_eps", 1e-6))


def build_model_and_tokenizer(*, is_training: bool = False, llm_pretrained: str = None, vision_pretrained: str = None,
                              set_vision_inside: bool = False, resume_from_checkpoint: str = "", frame_token_cls: bool = True,
                              frame_token_pooled=(3, 3), frame_num_tokens: int = 10, frame_token_interval: str = ",",
                              frame_resolution: int = 384, tokenizer_builder=None, kv_pool_tokens: int = 32768, device: int = 0,
                              **_ignored):
    """Keyword-compatible with ``build_model_and_tokenizer(is_training=False, set_vision_inside=True, **asdict(args))``
    (demo/inference.py:15): arguments that only matter for training (lora_*, finetune_modules, attn_implementation,
    torch_dtype, stream_loss_weight, ...) are accepted and ignored."""
    if is_training:
        raise NotImplementedError("the HIP engine is inference-only (SURVEY.md §8: training is out of scope)")
    if not llm_pretrained or not os.path.isdir(llm_pretrained):
        raise FileNotFoundError(f"llm_pretrained must be a local checkpoint directory, got {llm_pretrained!r}")
    hf = json.load(open(os.path.join(llm_pretrained, "config.json")))
    rope = hf.get("rope_parameters") or {}
    scaling = hf.get("rope_scaling") or ({} if rope.get("rope_type", "default") in ("default", None) else rope)
    if scaling and scaling.get("rope_type", scaling.get("type", "default")) != "default":
        raise NotImplementedError(f"rope scaling {scaling!r} is not implemented (the engine builds the plain rotary table of HF's "
                                  f"LlamaRotaryEmbedding; Llama-3.1-style 'llama3' scaling would give wrong logits silently)")
    for what, path in (("vision_pretrained", vision_pretrained if set_vision_inside else None), ("resume_from_checkpoint", resume_from_checkpoint or None)):
        if path is not None and not os.path.isdir(path):
            raise FileNotFoundError(f"{what} must be a local directory (hub ids are not resolved: there is no network path in the engine), got {path!r}")
    vit = _vit_config(vision_pretrained) if set_vision_inside else None
    cfg = EngineConfig(hidden_size=hf["hidden_size"], intermediate_size=hf["intermediate_size"],
                       num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=hf["num_attention_heads"],
                       num_key_value_heads=hf.get("num_key_value_heads", hf["num_attention_heads"]), vocab_size=hf["vocab_size"],
                       rope_theta=float(hf.get("rope_theta", rope.get("rope_theta", 10000.0))), rms_norm_eps=hf.get("rms_norm_eps", 1e-5),
                       vision_hidden_size=vit["hidden_size"] if vit else 1024, frame_num_tokens=frame_num_tokens,
                       frame_token_pooled=tuple(frame_token_pooled or (3, 3)), vit=vit, kv_pool_tokens=kv_pool_tokens)
    lcfg = LiveConfig(frame_token_interval=frame_token_interval or "", frame_num_tokens=frame_num_tokens,
                      frame_token_cls=frame_token_cls, frame_token_pooled=list(frame_token_pooled or ()))
    if tokenizer_builder is None:
        try:
            from models.tokenization_live import build_live_tokenizer_and_update_config as tokenizer_builder
        except Exception as ex:
            raise ImportError("build_model_and_tokenizer needs the reference's models/tokenization_live.py on sys.path "
                              "(run inside the videollm-online checkout) or an explicit tokenizer_builder(llm_pretrained, config)") from ex
    tokenizer = tokenizer_builder(llm_pretrained, lcfg)
    if len(tokenizer) - 1 != lcfg.v_placeholder_id:
        raise RuntimeError("tokenizer builder did not register the <v> placeholder as the last token")
    eng = Engine(cfg, device)
    if not resume_from_checkpoint:
        import logging
        logging.getLogger(__name__).warning(f"!!! Fail to load checkpoint: {resume_from_checkpoint}. Return a new initialized model.")
    load_engine_weights(eng, llm_pretrained, resume_from_checkpoint or None, vision_pretrained if set_vision_inside else None)
    eng.finalize()
    model = LiveModel(eng, eos_token_id=lcfg.eos_token_id, frame_token_interval_id=lcfg.frame_token_interval_id,
                      frame_resolution=frame_resolution, v_placeholder=lcfg.v_placeholder, v_placeholder_id=lcfg.v_placeholder_id)
    return model, tokenizer
