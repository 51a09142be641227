"""BASELINE.json configs[3] rehearsed on ONE GPU before an 8-GPU box exists: Llama tensor-parallel over T = 8 LOGICAL ranks (the
sharding arithmetic, the per-shard paged KV, the exchanges — sum kernels or the peer-to-peer mailboxes — all on one device) through
the package's LiveInfer over a 30 min @ 2 FPS stream (3 600 frames, ~47 k tokens per KV shard), at 2-layer Llama-3-8B width +
2-layer SigLIP-L so that it takes a minute.  Checks: KV length == sum of the logged steps, no exchange timed out (numerical parity
of the sharded step at this context: tests/test_gpu_long.py).  Prints one JSON line.

    python tools/rehearse_config4.py [--frames 3600] [--allreduce p2p|kernel] [--tp 8]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from videollm_online_amd.engine import EngineConfig, TpGroup
from videollm_online_amd.inference import LiveInfer
from videollm_online_amd.modeling_live import LiveModel
from videollm_online_amd.synthetic import LLM_SHAPES, VIT_SHAPE, gpu_random_weights, gpu_synthetic_frames, stream_tokens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3600)
    ap.add_argument("--allreduce", default="p2p", choices=["p2p", "kernel"])
    ap.add_argument("--tp", type=int, default=8)
    ap.add_argument("--layers", type=int, default=2)
    args = ap.parse_args()
    T = args.frames
    shape = dict(LLM_SHAPES["llama-3-8b"], num_hidden_layers=args.layers)
    kv_tokens = 64 + 11 * T + (T // 10 + 2) * 24 + 4096
    cfg = EngineConfig(**shape, vision_hidden_size=1024, vit=dict(VIT_SHAPE, num_layers=2), kv_pool_tokens=kv_tokens)
    eng = TpGroup(cfg, args.tp, allreduce="p2p" if args.allreduce == "p2p" else "default")
    gpu_random_weights(eng, cfg, seed=0)
    eng.finalize()
    toks = stream_tokens(cfg.vocab_size)
    model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
    li = LiveInfer(model, tokens=toks, frame_fps=2, prefetch=True, prefetch_frames=28, schedule=lambda i: (i % 10 == 9, 16), record=1 << 22)
    frames = gpu_synthetic_frames(T, seed=1234)
    li.load_video(frames)
    li.input_query_stream("Please narrate the video in real time.", video_time=0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(T):
        li.input_video_stream(i / 2)
        li()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kv = len(li.past_key_values)
    steps = li.steps_total
    assert kv == sum(n for _, n in li.step_log), (kv, sum(n for _, n in li.step_log))
    st = eng.p2p_status()
    assert not st["timed_out"], st
    exchanges = steps * (2 * args.layers + 1)
    out = {"what": f"configs[3] rehearsal on one GPU: TP={args.tp} logical ranks, {args.allreduce} exchanges, {args.layers}-layer Llama-3-8B width",
           "frames": T, "kv_tokens_per_shard": kv, "llm_steps": steps, "exchanges": exchanges, "seconds": round(dt, 2), "frames_per_s": round(T / dt, 2),
           "p2p_status": st, "kv_pages_per_shard": (kv + 255) // 256}
    print(json.dumps(out), flush=True)
    li.reset()
    eng.close()


if __name__ == "__main__":
    main()
