"""Persistent layer kernel (csrc/layer.hip, VLO_PERSISTENT): one cooperative launch per decoder layer, resident blocks walking
the launch pipeline's virtual grids between grid barriers, next-phase weights prefetched through the barriers.

(VLO_PERSISTENT_PREFETCH is read once per process by the engine: the parametrisation exercises the value of the FIRST case that
steps a persistent session; run the file twice with -k to see both.)

NOT YET RUN ON HARDWARE (written without GPU time; bit-identical to the launch pipeline in the CPU emulation, where the grid
barriers are real but the GPU memory model is not).  Opt-in (VLO_EXPERIMENTAL=1) until the first run.  On the GPU the logits
must equal the default pipeline's BIT FOR BIT: same kernel bodies, same virtual grids, same summation orders."""
import os

import pytest
import torch

from oracle import vlo_oracle as O

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("VLO_EXPERIMENTAL") != "1", reason="opt-in until first validated on a GPU (VLO_EXPERIMENTAL=1)")]


@pytest.mark.parametrize("name,seed,prefetch,whole_step,barrier",
                         [("tinyllama-2l", 5, 0, 0, "flat"), ("tinyllama-2l", 5, 1, 1, "flat"), ("llama-3-8b-2l", 6, 0, 0, "flat"),
                          ("llama-3-8b-2l", 6, 1, 0, "flat"), ("llama-3-8b-2l", 6, 1, 1, "flat"),
                          ("llama-3-8b-2l", 6, 1, 0, "xcd"), ("llama-3-8b-2l", 6, 1, 1, "xcd"), ("tinyllama-2l", 5, 1, 1, "xcd")])
def test_persistent_equals_launch_pipeline(name, seed, prefetch, whole_step, barrier, monkeypatch):
    from videollm_online_amd.engine import Engine, EngineConfig
    spec = O.LLM_SPECS[name]
    w = O.init_llm_weights(spec, seed=seed)
    toks = O.default_tokens(spec, seed=7, n_start=35)
    ref = O.LlamaOracle(spec, w, torch.bfloat16)
    cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                       num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                       num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size, rope_theta=spec.rope_theta,
                       rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=4096)
    eng = Engine(cfg, 0)
    eng.load_weights(w)
    eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    eng.finalize()
    monkeypatch.setenv("VLO_PERSISTENT", "1")                 # one resident block per CU
    monkeypatch.setenv("VLO_PERSISTENT_PREFETCH", str(prefetch))
    monkeypatch.setenv("VLO_PERSISTENT_STEP", str(whole_step))     # 1: all layers of a step in ONE launch
    monkeypatch.setenv("VLO_PERSISTENT_BARRIER", barrier)          # xcd: hierarchical barrier grouped by HW_REG_XCC_ID
    ps = eng.new_session()
    monkeypatch.delenv("VLO_PERSISTENT")
    monkeypatch.delenv("VLO_PERSISTENT_STEP")
    monkeypatch.delenv("VLO_PERSISTENT_BARRIER")
    ds = eng.new_session()
    g = torch.Generator().manual_seed(seed + 100)
    H = spec.hidden_size
    frame = lambda: torch.randn(10, H, generator=g).bfloat16()
    steps = [torch.cat([ref.embed(torch.tensor(toks.start_ids)), frame()]),         # 45 tokens: block path + 16-row tail
             torch.cat([ref.embed(torch.tensor([toks.interval_id])), frame()]),      # n = 11
             ref.embed(torch.tensor(toks.stream_generation_ids)),                    # n = 4
             ref.embed(torch.tensor([17])), ref.embed(torch.tensor([23])),          # decode
             torch.cat([ref.embed(torch.tensor([toks.eos_token_id] + toks.stream_prompt_ids)), frame()])]   # n = 13
    for i, x in enumerate(steps * 3):                                               # 18 steps: the KV grows past several split geometries
        lp, ap = eng.llm_step(ps, x.cuda(), want_last=True, want_all=True)
        ld, ad = eng.llm_step(ds, x.cuda(), want_last=True, want_all=True)
        torch.cuda.synchronize()
        assert ps.get_seq_length() == ds.get_seq_length()
        assert torch.equal(ap, ad) and torch.equal(lp, ld), f"step {i} (n = {x.shape[0]}): persistent and launch-per-phase logits differ"
    ps.close()
    ds.close()
    eng.close()
