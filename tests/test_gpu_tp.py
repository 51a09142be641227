"""Tensor-parallel sharding arithmetic on ONE GPU: T logical ranks in one process (exchanges = device kernels),
same weights, same inputs -> same logits / tokens as the oracle at the usual 3-way tolerance, and the KV cache is
really sharded by kv head.  (The RCCL exchange path needs a multi-GPU node; it shares everything but the exchange.)"""
import pytest
import torch

from oracle import vlo_oracle as O

pytestmark = pytest.mark.gpu


def _group(spec, w, T):
    from videollm_online_amd.engine import EngineConfig, TpGroup
    cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                       num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                       num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size, rope_theta=spec.rope_theta,
                       rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=2048)
    g = TpGroup(cfg, T)
    g.load_weights(w)
    g.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    return g.finalize()


@pytest.mark.parametrize("name,seed,T", [("toy128", 3, 2), ("tinyllama-2l", 5, 4), ("llama-3-8b-2l", 6, 8), ("llama-3-8b-2l", 6, 2)])
def test_tp_stream_parity(name, seed, T):
    spec = O.LLM_SPECS[name]
    w = O.init_llm_weights(spec, seed=seed)
    toks = O.default_tokens(spec, seed=7, n_start=35)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    grp = _group(spec, w, T)
    sess = grp.new_session()
    g = torch.Generator().manual_seed(seed + 100)
    H = spec.hidden_size
    frame = lambda: torch.randn(10, H, generator=g).bfloat16()
    steps = [torch.cat([ref.embed(torch.tensor(toks.start_ids)), frame()]),         # 45 tokens: 3 chunks
             torch.cat([ref.embed(torch.tensor([toks.interval_id])), frame()]),      # n = 11
             ref.embed(torch.tensor(toks.stream_generation_ids)),                    # n = 4
             ref.embed(torch.tensor([17])),                                         # n = 1
             torch.cat([ref.embed(torch.tensor([toks.eos_token_id] + toks.stream_prompt_ids)), frame()])]   # n = 13
    rc = gc = None
    for i, x in enumerate(steps):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = grp.llm_step(sess, x.cuda(), want_last=True, want_all=True)
        torch.cuda.synchronize()
        allr, last = allr.cpu(), last.cpu()
        assert sess.get_seq_length() == len(rc)
        assert torch.equal(last, allr[-1])
        e = (allr.float() - gl).abs().max().item()
        r = (rl.float() - gl).abs().max().item()
        scale = gl.abs().max().item()
        print(f"[tp{T} {name}] step {i}: engine err {e:.4g} ref-bf16 err {r:.4g} scale {scale:.3g}")
        assert e <= 1.5 * r + 1e-3 * scale, f"step {i}: {e} vs {r}"
    # samplers + generation through the group
    tok, p = grp.stream_sample(sess, 0.725, toks.interval_id)
    rt, rp = O.stream_sample(last.clone(), toks.interval_id, 0.725)
    assert int(tok) == rt
    ids = torch.zeros(6, dtype=torch.long, device="cuda")
    n = grp.greedy_generate(sess, grp.embed(torch.tensor(toks.stream_generation_ids)), toks.eos_token_id, ids, force_len=5)
    out = ids[:n].cpu().tolist()
    assert n == 5 and out[-1] == toks.eos_token_id and toks.eos_token_id not in out[:-1]
    assert sess.get_seq_length() == len(rc) + 4 + 4
    # the KV cache is sharded: each local engine holds num_kv_heads / T heads
    assert all(e.cfg.tp_size == T for e in grp.engines)
    sess.close()
    grp.close()


def test_tp_rejects_bad_partitions():
    from videollm_online_amd.engine import Engine, EngineConfig
    spec = O.LLM_SPECS["toy128"]          # 2 kv heads
    cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=1,
                       num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                       tp_rank=0, tp_size=4)
    with pytest.raises(RuntimeError, match="tp_size must divide"):
        Engine(cfg)


def test_rccl_binding_one_rank_roundtrip():
    """The RCCL entry points the one-process-per-GPU mode dlopens, driven with a 1-rank communicator on this GPU:
    unique id -> ncclCommInitRank -> fp32 sum all-reduce -> byte all-gather, data checked in the library."""
    from videollm_online_amd import _C
    _C.check(_C.lib().vlo_tp_selftest(0))
