"""Pin the oracle against the third-party modules it restates (HF transformers, the dependency where the
reference's arithmetic lives) instantiated with the same weights.  Runs anywhere transformers is installed
(build container and GPU box); needs neither /root/reference nor a GPU."""
import pytest
import torch

from oracle import vlo_oracle as O

transformers = pytest.importorskip("transformers")


def _hf_llama(spec, w, dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                      num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                      rms_norm_eps=spec.rms_eps, rope_parameters={"rope_type": "default", "rope_theta": spec.rope_theta},
                      attn_implementation="sdpa", tie_word_embeddings=False, max_position_embeddings=8192)
    m = LlamaForCausalLM(cfg)
    sd = {k: v.float() for k, v in w.items() if not k.startswith("connector.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in x for x in missing)
    m = m.to(dtype).eval()
    inv, _ = m.model.rotary_emb.compute_default_rope_parameters(m.config)      # keep inv_freq fp32 as from_pretrained does
    m.model.rotary_emb.inv_freq = inv.float()
    m.model.rotary_emb.original_inv_freq = inv.float().clone()
    return m


# one decoder layer at the Llama-3-8B width and head geometry (H 4096, I 14336, 32 query heads on 8 kv heads of 128, theta 5e5) with a
# reduced vocabulary: the width the oracle is used at as the checker of the GPU suite, not only the toy one
WIDE_1L = O.LlmSpec(4096, 14336, 1, 32, 8, 4096, 500000.0, 1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("which", ["toy128", "llama-3-8b-width-1l"])
def test_llama_step_sequence_matches_hf(dtype, which):
    spec = O.LLM_SPECS["toy128"] if which == "toy128" else WIDE_1L
    w = O.init_llm_weights(spec, seed=3)
    hf = _hf_llama(spec, w, dtype)
    me = O.LlamaOracle(spec, w, dtype)
    g = torch.Generator().manual_seed(0)
    past, cache = None, None
    for n in (29, 11, 1, 13, 1):
        x = torch.randn(n, spec.hidden_size, generator=g).to(dtype)
        with torch.no_grad():
            o = hf(inputs_embeds=x[None], use_cache=True, past_key_values=past)
        past = o.past_key_values
        lg, cache = me.forward(x, cache)
        if dtype == torch.bfloat16:
            assert torch.equal(o.logits[0], lg)          # same ops, same order, same rounding points
        else:
            assert (o.logits[0] - lg).abs().max().item() < 2e-4 * max(1.0, float(lg.abs().max()) / 8)
        assert past.get_seq_length() == len(cache)


# toy-hd72: SigLIP-so400m/14's irregularities (head dim 72, MLP % 64 != 0, 14-pixel patches); so400m-1l: one layer at its true geometry
# (hidden 1152, MLP 4304, 384 / 14 -> 27 x 27 patches: the strided conv drops the last 6 pixels)
@pytest.mark.parametrize("which", ["toy", "toy-hd72", "so400m-1l"])
def test_siglip_tower_matches_hf(which):
    import dataclasses
    from transformers import SiglipVisionConfig, SiglipVisionModel
    vspec = dataclasses.replace(O.VIT_SPECS["siglip-so400m14-384-2l"], num_layers=1) if which == "so400m-1l" else O.VIT_SPECS[which]
    vw = O.init_vit_weights(vspec, seed=1)
    cfg = SiglipVisionConfig(hidden_size=vspec.hidden_size, intermediate_size=vspec.intermediate_size,
                             num_hidden_layers=vspec.num_layers, num_attention_heads=vspec.num_heads,
                             image_size=vspec.image_size, patch_size=vspec.patch_size, layer_norm_eps=vspec.ln_eps)
    vit = SiglipVisionModel(cfg).eval()
    vit.load_state_dict({k[len("vision."):]: v for k, v in vw.items()}, strict=True)
    frames = O.synthetic_frames(2, vspec.image_size, seed=3)
    x = (frames * 0.00392156862745098 - 0.5) / 0.5
    with torch.no_grad():
        out = vit(x)
    last, pooled = O.vit_forward(vw, vspec, x)
    assert (out.last_hidden_state - last).abs().max().item() < 2e-5
    assert (out.pooler_output - pooled).abs().max().item() < 2e-5


def test_connector_gelu_is_python_erf_gelu():
    """GELUActivation(config.hidden_size) => use_gelu_python=True (models/live_llama/modeling_live_llama.py:20)."""
    from transformers.activations import GELUActivation
    act = GELUActivation(4096)
    x = torch.randn(64, 33).bfloat16()
    assert torch.equal(act(x), O.gelu_python(x))
    assert not torch.equal(act(x), torch.nn.functional.gelu(x, approximate="tanh"))
