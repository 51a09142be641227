"""Llama step timing on the true Llama-3-8B shape: decode steps (n = 1) and frame steps (n = 11) at a few cache lengths, for the
pipeline variants selected by environment (each variant in its own process: the switches are read once).

    python tools/probe_step.py [--model llama-3-8b] [--iters 40]      # e.g. VLO_FIXUP=0 python tools/probe_step.py
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
    ap.add_argument("--weight-dtype", default="bf16")
    ap.add_argument("--lens", default="0,4096,12288", help="cache lengths to time at")
    ap.add_argument("--ns", default="1,11", help="new tokens per step")
    args = ap.parse_args()
    cfg = EngineConfig(**SHAPES[args.model], kv_pool_tokens=65536, weight_dtype=args.weight_dtype)
    eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    eng.finalize()
    H = cfg.hidden_size
    sess = eng.new_session()
    fill = torch.randn(64, H, device="cuda").bfloat16()
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VLO_")) or "defaults"
    for Lc in [int(v) for v in args.lens.split(",")]:
        while sess.get_seq_length() < Lc:
            eng.llm_step(sess, fill, want_last=False)
        print(f"[{tag}, {args.weight_dtype}] Lc~{Lc:6d}:  " + "  ".join(f"n={n}: {timed(eng, sess, torch.randn(n, H, device='cuda').bfloat16(), args.iters):.3f} ms"
                                                   for n in [int(v) for v in args.ns.split(",")]), flush=True)


if __name__ == "__main__":
    main()
