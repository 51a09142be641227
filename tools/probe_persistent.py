"""A/B timing of the persistent layer kernel (VLO_PERSISTENT, csrc/layer.hip) against the launch-per-phase pipeline on the true
Llama-3-8B shape: decode steps (n = 1) and frame steps (n = 11) at a few cache lengths, with and without the cross-phase
weight prefetch.

    python tools/probe_persistent.py [--model llama-3-8b] [--iters 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.probe_fused_rows import timed
from tools.probe_llm import SHAPES, random_llm_weights_to_engine
from videollm_online_amd.engine import Engine, EngineConfig


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
    sessions["launches"] = eng.new_session()
    os.environ["VLO_PERSISTENT"] = "1"          # one resident block per CU; VLO_PERSISTENT_PREFETCH (default 1) is read once per process
    sessions["persistent/layer"] = eng.new_session()
    os.environ["VLO_PERSISTENT_STEP"] = "1"     # all layers of a step in ONE launch
    sessions["persistent/step"] = eng.new_session()
    os.environ["VLO_PERSISTENT_BARRIER"] = "xcd"    # XCD-hierarchical grid barrier
    sessions["persistent/step/xcd"] = eng.new_session()
    os.environ.pop("VLO_PERSISTENT_BARRIER", None)
    os.environ.pop("VLO_PERSISTENT", None)
    os.environ.pop("VLO_PERSISTENT_STEP", None)
    fill = torch.randn(64, H, device="cuda").bfloat16()
    print(f"VLO_PERSISTENT_PREFETCH={os.environ.get('VLO_PERSISTENT_PREFETCH', '1 (default)')} for the persistent column; "
          f"run again with VLO_PERSISTENT_PREFETCH=0 for the other variant")
    for Lc in (0, 4096, 12288):
        for label, sess in sessions.items():
            while sess.get_seq_length() < Lc:
                eng.llm_step(sess, fill, want_last=False)
        for n in (1, 11):
            x = torch.randn(n, H, device="cuda").bfloat16()
            print(f"Lc~{Lc:6d} n={n:2d}:  " + "  ".join(f"{label} {timed(eng, sess, x, args.iters):.3f} ms" for label, sess in sessions.items()), flush=True)


if __name__ == "__main__":
    main()
