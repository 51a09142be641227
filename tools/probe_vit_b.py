"""ViT encode at ONE batch size (for clean per-kernel rocprofv3 stats): python tools/probe_vit_b.py B [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from probe_llm import random_llm_weights_to_engine
from probe_vit import load_random_vit
from videollm_online_amd.engine import Engine, EngineConfig

if __name__ == "__main__":
    Bs = [int(b) for b in sys.argv[1].split(",")]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfg = EngineConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=1, num_attention_heads=32,
                       num_key_value_heads=4, vocab_size=32000, kv_pool_tokens=1024,
                       vit=dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16))
    gflop_per_frame = 384.4
    if os.environ.get("VLO_PROBE_VIT") == "so400m":        # SigLIP-so400m/14-384 (BASELINE.json configs[4])
        cfg.vit = dict(hidden_size=1152, intermediate_size=4304, num_layers=27, num_heads=16, image_size=384, patch_size=14)
        cfg.vision_hidden_size = 1152
        D, I, S, L = 1152, 4304, 729, 27
        gflop_per_frame = (L * (2 * S * D * (4 * D + 2 * I) + 4 * S * S * D) + 2 * S * 588 * D) / 1e9
    eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    if os.environ.get("VLO_PROBE_VIT") == "so400m":
        load_random_vit(eng, D=1152, I=4304, L=27, P=14, S=729)
    else:
        load_random_vit(eng)
    eng.finalize()
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    for B in Bs:
        frames = torch.randint(0, 256, (B, 3, 384, 384), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            eng.visual_embed(frames)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            eng.visual_embed(frames)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        print(f"B={B}: {ms:.3f} ms per call, {ms / B:.3f} ms/frame, {gflop_per_frame * 1e9 * B / (ms * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
