"""Opt-in fused short-chunk path (csrc/engine.hip::run_chunk_fused, VLO_FUSED_ROWS): 6 launches per decoder layer, both
RMSNorms and the final norm on the GEMV operand loads, whole-K down-proj with the residual epilogue.

NOT YET RUN ON HARDWARE in its present form (the same flow was the default and green at commit a149b49; the kernels
have changed since), so the tests are opt-in (VLO_EXPERIMENTAL=1) until their first run.  The session reads
VLO_FUSED_ROWS when it is created, so the fused and the default pipeline run side by side in one process."""
import os

import pytest
import torch

from oracle import vlo_oracle as O

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("VLO_EXPERIMENTAL") != "1", reason="opt-in until first validated on a GPU (VLO_EXPERIMENTAL=1)")]


def _engine(spec, w):
    from videollm_online_amd.engine import Engine, EngineConfig
    cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                       num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                       num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size, rope_theta=spec.rope_theta,
                       rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=4096)
    eng = Engine(cfg, 0)
    eng.load_weights(w)
    eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    return eng.finalize()


@pytest.mark.parametrize("name,seed,rows", [("toy128", 3, 16), ("tinyllama-2l", 5, 16), ("llama-3-8b-2l", 6, 16), ("llama-3-8b-2l", 6, 1)])
def test_fused_rows_stream_parity(name, seed, rows, monkeypatch):
    spec = O.LLM_SPECS[name]
    w = O.init_llm_weights(spec, seed=seed)
    toks = O.default_tokens(spec, seed=7, n_start=35)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = _engine(spec, w)
    monkeypatch.setenv("VLO_FUSED_ROWS", str(rows))
    fused = eng.new_session()
    monkeypatch.delenv("VLO_FUSED_ROWS")
    plain = eng.new_session()
    g = torch.Generator().manual_seed(seed + 100)
    H = spec.hidden_size
    frame = lambda: torch.randn(10, H, generator=g).bfloat16()
    steps = [torch.cat([ref.embed(torch.tensor(toks.start_ids[:6])), frame()]),     # 16 rows: one chunk
             torch.cat([ref.embed(torch.tensor([toks.interval_id])), frame()]),      # n = 11
             ref.embed(torch.tensor(toks.stream_generation_ids)),                    # n = 4
             ref.embed(torch.tensor([17])),                                         # n = 1 (decode)
             ref.embed(torch.tensor([23])),                                         # n = 1
             torch.cat([ref.embed(torch.tensor([toks.eos_token_id] + toks.stream_prompt_ids)), frame()])]   # n = 13
    rc = gc = None
    diverged = False                          # once a fused chunk has written KV, later default-pipeline chunks see other history
    for i, x in enumerate(steps):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        outs = {}
        for kind, sess in (("fused", fused), ("plain", plain)):
            last, allr = eng.llm_step(sess, x.cuda(), want_last=True, want_all=True)
            torch.cuda.synchronize()
            assert sess.get_seq_length() == len(rc)
            assert torch.equal(last.cpu(), allr.cpu()[-1])
            outs[kind] = allr.cpu().float()
        e = (outs["fused"] - gl).abs().max().item()
        r = (rl.float() - gl).abs().max().item()
        d = (outs["fused"] - outs["plain"]).abs().max().item()
        scale = gl.abs().max().item()
        print(f"[fused<= {rows} {name}] step {i} n={x.shape[0]}: engine err {e:.4g} ref-bf16 err {r:.4g} vs default pipeline {d:.4g}")
        assert e <= 1.5 * r + 1e-3 * scale, f"step {i}: {e} vs {r}"
        if x.shape[0] <= rows:
            diverged = True
        elif not diverged:
            assert d == 0.0, "chunks longer than VLO_FUSED_ROWS must take the default pipeline"
    # last-row-only logits (the live path: want_all = False) and the samplers
    x = ref.embed(torch.tensor([29]))
    rl, rc = ref.forward(x, rc)
    a, _ = eng.llm_step(fused, x.cuda())
    b, _ = eng.llm_step(plain, x.cuda())
    torch.cuda.synchronize()
    assert (a.float() - b.float()).abs().max().item() <= 0.5 * (rl[-1].float() - a.cpu().float()).abs().max().item() + 0.05
    tf, _ = eng.stream_sample(fused, 0.725, toks.interval_id)
    tp, _ = eng.stream_sample(plain, 0.725, toks.interval_id)
    top2 = rl[-1].float().topk(2).values
    assert int(tf) == int(tp) or (top2[0] - top2[1]).item() < 0.12
    fused.close()
    plain.close()
    eng.close()
