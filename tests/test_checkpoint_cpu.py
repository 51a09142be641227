"""Host logic of the checkpoint loader on synthetic safetensors: LoRA merge == the
reference's un-merged run-time LoRA up to rounding, connector restored from modules_to_save, SigLIP keys mapped."""
import json

import pytest
import torch
from safetensors.torch import save_file

from oracle import vlo_oracle as O
from videollm_online_amd import checkpoint as CK


@pytest.fixture()
def ckpt(tmp_path):
    spec = O.LLM_SPECS["toy"]
    w = O.init_llm_weights(spec, seed=0)
    base = {k: v for k, v in w.items() if not k.startswith("connector.")}
    # two shards + index, like a real HF checkpoint
    keys = sorted(base)
    a, b = keys[: len(keys) // 2], keys[len(keys) // 2:]
    bdir = tmp_path / "base"; bdir.mkdir()
    save_file({k: base[k].contiguous() for k in a}, str(bdir / "model-00001-of-00002.safetensors"))
    save_file({k: base[k].contiguous() for k in b}, str(bdir / "model-00002-of-00002.safetensors"))
    json.dump({"weight_map": {**{k: "model-00001-of-00002.safetensors" for k in a},
                              **{k: "model-00002-of-00002.safetensors" for k in b}}},
              open(bdir / "model.safetensors.index.json", "w"))
    # PEFT adapter: LoRA on every linear + lm_head, connector under modules_to_save
    g = torch.Generator().manual_seed(1)
    r, alpha = 8, 16
    ad = {}
    lora = {}
    for k, v in base.items():
        mod = k[:-len(".weight")]
        if CK.LORA_TARGETS.search(mod) and v.dim() == 2 and "embed" not in k and "norm" not in k:
            A = (torch.randn(r, v.shape[1], generator=g) * 0.05).bfloat16()
            B = (torch.randn(v.shape[0], r, generator=g) * 0.05).bfloat16()
            ad[f"base_model.model.{mod}.lora_A.weight"] = A
            ad[f"base_model.model.{mod}.lora_B.weight"] = B
            lora[mod] = (A, B)
    for k, v in w.items():
        if k.startswith("connector."):
            ad[f"base_model.model.{k}"] = v.contiguous()
    adir = tmp_path / "adapter"; adir.mkdir()
    save_file(ad, str(adir / "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": "x", "modules_to_save": ["connector"]},
              open(adir / "adapter_config.json", "w"))
    return spec, w, lora, alpha / r, str(bdir), str(adir)


def test_merge_matches_runtime_lora(ckpt):
    spec, w, lora, scale, bdir, adir = ckpt
    merged = dict(CK.iter_llm_weights(bdir, adir))
    assert set(merged) == set(w)                                   # every HF name incl. connector, nothing extra
    assert len(lora) == 7 * spec.num_layers + 1                    # q,k,v,o,gate,up,down per layer + lm_head
    x = torch.randn(5, spec.hidden_size, generator=torch.Generator().manual_seed(2))
    for mod in ("model.layers.0.self_attn.q_proj", "model.layers.1.mlp.up_proj", "lm_head"):
        A, B = lora[mod]
        W = w[mod + ".weight"]
        ref = x @ W.float().T + scale * ((x @ A.float().T) @ B.float().T)      # PEFT's un-merged forward
        got = x @ merged[mod + ".weight"].float().T
        assert (got - ref).abs().max() <= 2 ** -8 * ref.abs().max() + 1e-3      # one bf16 rounding of W'
    for k in ("connector.0.weight", "connector.2.bias", "model.norm.weight", "model.embed_tokens.weight"):
        assert torch.equal(merged[k], w[k])                        # untouched tensors pass through bit-exact


def test_no_adapter_is_identity_and_bad_adapter_fails(ckpt, tmp_path):
    spec, w, lora, scale, bdir, adir = ckpt
    plain = dict(CK.iter_llm_weights(bdir, None))
    assert all(torch.equal(plain[k], w[k]) for k in plain)
    bad = tmp_path / "bad"; bad.mkdir()
    save_file({"base_model.model.model.layers.9.self_attn.q_proj.lora_A.weight": torch.zeros(2, 4),
               "base_model.model.model.layers.9.self_attn.q_proj.lora_B.weight": torch.zeros(4, 2)},
              str(bad / "adapter_model.safetensors"))
    json.dump({"r": 2, "lora_alpha": 4}, open(bad / "adapter_config.json", "w"))
    with pytest.raises(KeyError, match="absent from the base"):
        dict(CK.iter_llm_weights(bdir, str(bad)))


def test_siglip_key_mapping(tmp_path):
    vspec = O.VIT_SPECS["toy"]
    vw = O.init_vit_weights(vspec, seed=1)
    sd = {"vision_model." + k[len("vision."):]: v.contiguous() for k, v in vw.items()}
    sd["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)
    sd["logit_scale"] = torch.zeros(1)
    d = tmp_path / "siglip"; d.mkdir()
    save_file(sd, str(d / "model.safetensors"))
    got = dict(CK.iter_vision_weights(str(d)))
    assert set(got) == set(vw)
    assert all(torch.equal(got[k], vw[k]) for k in vw)


def test_builder_vision_config_accepts_the_towers_the_engine_runs(tmp_path):
    """builder._vit_config reads the vision tower's config.json: SigLIP-L/16-384 (what the reference accepts, models/vision_live.py:56-60)
    and SigLIP-so400m/14-384 (BASELINE.json configs[4]; head dim 72) map to the engine's vit dict, a head dim the kernels do not cover
    is refused up front (the same rule as csrc/vit.hip::vit_finalize)."""
    import json
    from videollm_online_amd.builder import _vit_config

    def cfg_dir(name, **vision):
        d = tmp_path / name
        d.mkdir()
        (d / "config.json").write_text(json.dumps({"vision_config": vision}))
        return str(d)

    l = _vit_config(cfg_dir("l", hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=384, patch_size=16))
    assert (l["hidden_size"], l["num_heads"], l["patch_size"], l["num_layers"]) == (1024, 16, 16, 24)
    so = _vit_config(cfg_dir("so", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=384, patch_size=14))
    assert (so["hidden_size"], so["intermediate_size"], so["num_heads"], so["patch_size"], so["num_layers"]) == (1152, 4304, 16, 14, 27)
    with pytest.raises(ValueError, match="unsupported vision tower"):
        _vit_config(cfg_dir("big", hidden_size=1536, intermediate_size=6144, num_hidden_layers=40, num_attention_heads=16, image_size=224, patch_size=14))

