"""Does running two half-batches of the ViT encode on two HIP streams beat one full batch on one stream?  (tails / ramps of one
graph's kernels overlapping the other's)   python tools/probe_vit_2streams.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from probe_llm import random_llm_weights_to_engine
from probe_vit import load_random_vit
from videollm_online_amd.engine import Engine, EngineConfig

cfg = EngineConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=1, num_attention_heads=32,
                   num_key_value_heads=4, vocab_size=32000, kv_pool_tokens=1024,
                   vit=dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16))
eng = Engine(cfg)
random_llm_weights_to_engine(eng, cfg)
load_random_vit(eng)
eng.finalize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in (8, 14, 16):
    frames = torch.randint(0, 256, (B, 3, 384, 384), dtype=torch.uint8, device="cuda")
    h = B // 2
    outs = [torch.empty(h * 10, 2048, dtype=torch.bfloat16, device="cuda"), torch.empty((B - h) * 10, 2048, dtype=torch.bfloat16, device="cuda")]

    def one():
        with torch.cuda.stream(s1):
            eng.visual_embed(frames, stream=s1)

    def two():
        with torch.cuda.stream(s1):
            eng.visual_embed(frames[:h], stream=s1, out=outs[0])
        with torch.cuda.stream(s2):
            eng.visual_embed(frames[h:], stream=s2, out=outs[1])

    for name, fn in (("one stream", one), ("two streams", two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        for _ in range(10):
            fn()
        j1, j2 = torch.cuda.Event(), torch.cuda.Event()
        j1.record(s1); j2.record(s2)
        torch.cuda.current_stream().wait_event(j1); torch.cuda.current_stream().wait_event(j2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"B={B} {name}: {ms:.3f} ms per {B} frames, {ms / B:.3f} ms/frame, {384.4e9 * B / (ms * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
