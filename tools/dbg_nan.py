import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vlo_oracle as O
from videollm_online_amd.engine import Engine, EngineConfig
from videollm_online_amd.synthetic import gpu_synthetic_frames
spec, vspec = O.LLM_SPECS["llama-3-8b-2l"], O.VIT_SPECS["siglip-l16-384-2l"]
w, vw = O.init_llm_weights(spec, seed=31), O.init_vit_weights(vspec, seed=32)
cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                   num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                   rope_theta=spec.rope_theta, rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=4096,
                   frame_num_tokens=vspec.frame_num_tokens, frame_token_pooled=vspec.pooled,
                   vit=dict(hidden_size=vspec.hidden_size, intermediate_size=vspec.intermediate_size, num_layers=vspec.num_layers,
                            num_heads=vspec.num_heads, image_size=vspec.image_size, patch_size=vspec.patch_size, ln_eps=vspec.ln_eps))
eng = Engine(cfg); eng.load_weights(w); eng.load_weights(vw); eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta)); eng.finalize()
frames = gpu_synthetic_frames(64, seed=1234)
print("frames", frames.shape, frames.dtype, frames.float().mean().item())
for B in (1, 2, 8, 28, 1, 28):
    e = eng.visual_embed(frames[:B]).float()
    torch.cuda.synchronize()
    print("B", B, "nan", torch.isnan(e).sum().item(), "inf", torch.isinf(e).sum().item(), "absmax", e.abs().max().item())
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    e = eng.visual_embed(frames[1:29], stream=st).float()
st.synchronize()
print("stream B28 nan", torch.isnan(e).sum().item(), "absmax", e.abs().max().item())
tok = eng.vision_tokens(frames[:3]).float(); torch.cuda.synchronize()
print("vision_tokens nan", torch.isnan(tok).sum().item(), tok.abs().max().item())
g = O.LlamaOracle(spec, w, torch.float32).visual_embed(vw, vspec, frames[:2].cpu())
e = eng.visual_embed(frames[:2]).float().cpu()
print("vs gold err", (e - g).abs().max().item(), "scale", g.abs().max().item())
