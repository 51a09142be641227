"""A/B timing of the persistent layer kernel (VLO_PERSISTENT, csrc/layer.hip) against the launch-per-phase pipeline on the true
Llama-3-8B shape: decode steps (n = 1) and frame steps (n = 11) at a few cache lengths.  Every variant is timed on its own and
a failing one is reported, not fatal.

    python tools/probe_persistent.py [--model llama-3-8b] [--iters 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.probe_llm import SHAPES, random_llm_weights_to_engine
from videollm_online_amd.engine import Engine, EngineConfig


def timed(eng, sess, x, iters):
    for _ in range(3):
        eng.llm_step(sess, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.llm_step(sess, x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    cfg = EngineConfig(**SHAPES[args.model], kv_pool_tokens=65536)
    eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    eng.finalize()
    H = cfg.hidden_size
    variants = {"launches": {}, "persistent/layer": {"VLO_PERSISTENT": "1"},
                "persistent/layer/xcd": {"VLO_PERSISTENT": "1", "VLO_PERSISTENT_BARRIER": "xcd"},
                "persistent/step": {"VLO_PERSISTENT": "1", "VLO_PERSISTENT_STEP": "1"},
                "persistent/step/xcd": {"VLO_PERSISTENT": "1", "VLO_PERSISTENT_STEP": "1", "VLO_PERSISTENT_BARRIER": "xcd"}}
    sessions = {}
    for label, env in variants.items():
        os.environ.update(env)
        sessions[label] = eng.new_session()
        for k in env:
            os.environ.pop(k, None)
    ref = None
    fill = torch.randn(64, H, device="cuda").bfloat16()
    print(f"VLO_PERSISTENT_PREFETCH={os.environ.get('VLO_PERSISTENT_PREFETCH', '1 (default)')}")
    dead = set()
    for Lc in (0, 4096, 12288):
        for label, sess in sessions.items():
            if label in dead:
                continue
            try:
                while sess.get_seq_length() < Lc:
                    eng.llm_step(sess, fill, want_last=False)
            except Exception as ex:
                print(f"{label}: fill failed: {ex}")
                dead.add(label)
        for n in (1, 11):
            x = torch.randn(n, H, device="cuda").bfloat16()
            out = []
            for label, sess in sessions.items():
                if label in dead:
                    continue
                try:
                    out.append(f"{label} {timed(eng, sess, x, args.iters):.3f} ms")
                except Exception as ex:
                    print(f"{label}: FAILED at Lc~{Lc} n={n}: {ex}")
                    dead.add(label)
            print(f"Lc~{Lc:6d} n={n:2d}:  " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
