import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videollm_online_amd.engine import Engine, EngineConfig
from probe_llm import random_llm_weights_to_engine
from probe_vit import load_random_vit
cfg = EngineConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=4,
                   vocab_size=32000, kv_pool_tokens=1024,
                   vit=dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16))
eng = Engine(cfg); random_llm_weights_to_engine(eng, cfg); load_random_vit(eng); eng.finalize()
frames = torch.randint(0, 256, (32, 3, 384, 384), dtype=torch.uint8, device="cuda")
for _ in range(3): eng.vision_tokens(frames)
torch.cuda.synchronize()
