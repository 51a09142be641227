"""ViT (SigLIP-L/16-384 + connector) timing probe with random weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videollm_online_amd.engine import Engine, EngineConfig


def load_random_vit(eng, D=1024, I=4096, L=24, P=16, S=576, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s, std=1.0: torch.randn(*s, generator=g, device="cuda") * std
    eng.load_weight("vision.embeddings.patch_embedding.weight", r(D, 3, P, P, std=(3 * P * P) ** -0.5))
    eng.load_weight("vision.embeddings.patch_embedding.bias", r(D, std=0.1))
    eng.load_weight("vision.embeddings.position_embedding.weight", r(S, D, std=0.5))
    def ln(p): eng.load_weight(p + ".weight", 1 + r(D, std=0.1)); eng.load_weight(p + ".bias", r(D, std=0.1))
    def lin(p, o, i): eng.load_weight(p + ".weight", r(o, i, std=i ** -0.5)); eng.load_weight(p + ".bias", r(o, std=0.1))
    for i in range(L):
        p = f"vision.encoder.layers.{i}."
        ln(p + "layer_norm1"); ln(p + "layer_norm2")
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"): lin(p + "self_attn." + n, D, D)
        lin(p + "mlp.fc1", I, D); lin(p + "mlp.fc2", D, I)
    ln("vision.post_layernorm")
    eng.load_weight("vision.head.probe", r(1, 1, D))
    eng.load_weight("vision.head.attention.in_proj_weight", r(3 * D, D, std=D ** -0.5))
    eng.load_weight("vision.head.attention.in_proj_bias", r(3 * D, std=0.1))
    lin("vision.head.attention.out_proj", D, D); ln("vision.head.layernorm")
    lin("vision.head.mlp.fc1", I, D); lin("vision.head.mlp.fc2", D, I)


if __name__ == "__main__":
    from probe_llm import random_llm_weights_to_engine
    cfg = EngineConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=1, num_attention_heads=32,
                       num_key_value_heads=4, vocab_size=32000, kv_pool_tokens=1024,
                       vit=dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16))
    eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    load_random_vit(eng)
    eng.finalize()
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    for B in (1, 2, 4, 8):
        frames = torch.randint(0, 256, (B, 3, 384, 384), dtype=torch.uint8, device="cuda")
        for _ in range(3): eng.visual_embed(frames)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20): eng.visual_embed(frames)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 20
        print(f"B={B}: {ms:.3f} ms per call, {ms/B:.3f} ms/frame, {384.4e9*B/(ms*1e-3)/1e12:.1f} TFLOP/s")
    # offline feature extraction shape (data/preprocess/encode.py: batch 256): the MFMA-bound regime
    for B in (32, 256):
        frames = torch.randint(0, 256, (B, 3, 384, 384), dtype=torch.uint8, device="cuda")
        eng.vision_tokens(frames); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); eng.vision_tokens(frames); eng.vision_tokens(frames); t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 2
        print(f"vision_tokens B={B}: {ms:.2f} ms per call, {ms/B:.3f} ms/frame, {384.0e9*B/(ms*1e-3)/1e12:.1f} TFLOP/s")
