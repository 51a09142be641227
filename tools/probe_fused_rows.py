"""A/B timing of the opt-in fused short-chunk pipeline (VLO_FUSED_ROWS, csrc/engine.hip::run_chunk_fused) against the
default one on the true Llama-3-8B shape: decode steps (n = 1) and frame steps (n = 11) at a few cache lengths.

    python tools/probe_fused_rows.py [--model llama-3-8b] [--iters 40]
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
    sessions = {}
    for label, rows in (("default", None), ("fused<=1", 1), ("fused<=16", 16)):
        if rows is None:
            os.environ.pop("VLO_FUSED_ROWS", None)
        else:
            os.environ["VLO_FUSED_ROWS"] = str(rows)
        sessions[label] = eng.new_session()
    os.environ.pop("VLO_FUSED_ROWS", None)
    fill = torch.randn(64, H, device="cuda").bfloat16()
    for Lc in (0, 4096, 12288):
        for label, sess in sessions.items():
            while sess.get_seq_length() < Lc:
                eng.llm_step(sess, fill, want_last=False)
        for n in (1, 11):
            x = torch.randn(n, H, device="cuda").bfloat16()
            line = [f"Lc~{Lc:6d} n={n:2d}:"]
            for label, sess in sessions.items():
                line.append(f"{label} {timed(eng, sess, x, args.iters):.3f} ms")
            print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
