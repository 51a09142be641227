"""Teacher-forced prefill timing on the true Llama-3-8B shape: N tokens through vlo_llm_step (prefill path = GEMMs over blocks of up to
4096 tokens; VLO_PREFILL=0: the 64-token block path; VLO_BLOCK_PATH=0: 16-row chunks), with and without all-row logits + per-row
statistics (the stream_evaluate pass)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videollm_online_amd.engine import Engine, EngineConfig
from probe_llm import SHAPES, random_llm_weights_to_engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--weight-dtype", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--act", default="bf16", choices=["bf16", "fp8"], help="fp8 engines: X operand of the prefill GEMMs (EngineConfig.prefill_act_dtype; fp8 = native fp8 MFMA)")
    ap.add_argument("--gemm", action="store_true", help="time the W8A8 GEMM alone on the layer's four projection shapes (vlo_test_gemm_fp8) and exit")
    ap.add_argument("--tp", type=int, default=1, help="T logical tensor-parallel ranks on this one GPU (csrc/tp.hip::tp_prefill): the ranks' shards run one "
                    "after the other, so time / T is what ONE rank of a T-GPU group spends on its GEMMs + attention (its all-reduce over xGMI not included)")
    args = ap.parse_args()
    if args.gemm:
        from videollm_online_amd.checkpoint import quantize_fp8_per_channel
        from videollm_online_amd.engine import test_gemm_fp8
        sh = SHAPES[args.model]
        H, I, hd = sh["hidden_size"], sh["intermediate_size"], sh["hidden_size"] // sh["num_attention_heads"]
        qd = sh["num_attention_heads"] * hd
        for name, N, K in (("qkv", qd + 2 * sh["num_key_value_heads"] * hd, H), ("o", H, qd), ("gate_up", 2 * I, H), ("down", H, I)):
            M = args.tokens
            x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
            q, s = quantize_fp8_per_channel((torch.randn(N, K, device="cuda") * 0.02).bfloat16())
            _, _, _, us = test_gemm_fp8(x, q, s, iters=20)
            print(f"[fp8 mfma gemm] {name:8s} M {M} N {N} K {K}: {us:9.1f} us = {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s")
        return
    cfg = EngineConfig(**SHAPES[args.model], kv_pool_tokens=max(16384, 2 * args.tokens), weight_dtype=args.weight_dtype, prefill_act_dtype=args.act)
    if args.tp > 1:
        from videollm_online_amd.engine import TpGroup
        eng = TpGroup(cfg, args.tp)
    else:
        eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    eng.finalize()
    H = cfg.hidden_size
    x = (torch.randn(args.tokens, H, device="cuda") * 0.5).bfloat16()
    labels = torch.randint(0, cfg.vocab_size, (args.tokens,), device="cuda")
    mode = "16-row chunks" if os.environ.get("VLO_BLOCK_PATH") == "0" else ("64-token blocks" if os.environ.get("VLO_PREFILL") == "0" else "prefill GEMMs")
    for want_all in (False, True):
        sess = eng.new_session()
        eng.llm_step(sess, x[:min(args.tokens, 600)], want_last=True, want_all=want_all)          # warm-up (allocates the block / prefill workspaces)
        sess.reset()
        torch.cuda.synchronize()
        t0 = time.time()
        if want_all:
            for a in range(0, args.tokens, 2048):
                _, lg = eng.llm_step(sess, x[a:a + 2048], want_last=False, want_all=True)
                eng.logit_rows(lg, labels[a:a + 2048], 11)
        else:
            eng.llm_step(sess, x)
        torch.cuda.synchronize()
        dt = time.time() - t0
        tp = f", TP={args.tp} logical ranks (per-rank share of the time: {dt * 1e3 / args.tp:.1f} ms)" if args.tp > 1 else ""
        print(f"[{mode}, {args.weight_dtype} weights, {args.act} activations{tp}] {args.tokens} tokens, all-row logits+stats={want_all}: {dt*1e3:.1f} ms = {args.tokens/dt:.0f} tok/s "
              f"({dt*1e3/args.tokens*64:.2f} ms per 64 tokens)")
        sess.close()


if __name__ == "__main__":
    main()
