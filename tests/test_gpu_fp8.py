"""fp8 e4m3 weight streaming (BASELINE.json configs[4]: "Llama-3-70B TP=8 ... fp8 MFMA weights"; include/vlo.h
vlo_config.weight_dtype = 1): the Llama projections are stored as OCP e4m3 with one fp32 scale per output channel, streamed at
one byte per weight, expanded to bf16 in registers (exact) and multiplied on the bf16 matrix cores; activations, KV cache and
accumulation are unchanged.

The reference has no fp8 path (its weights are bf16; config 5 exceeds it).  Parity target = the reference's
arithmetic on the weights an fp8 store holds (oracle.fp8_dequantized_weights): 3-way as everywhere else —
err(engine, fp32 gold) <= 1.5 * err(bf16-activation reference, fp32 gold) + 1e-3 * max|logit| — plus the GEMV alone against
an fp64 matmul of the dequantised weights."""
import pytest
import torch

from oracle import vlo_oracle as O
from parity_util import fmt, ulp_report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,N,K", [(1, 256, 4096), (11, 6144, 4096), (11, 4096, 14336), (16, 1280, 8192), (11, 1024, 28672),
                                   (13, 8192, 1024), (11, 8192, 3584), (5, 2048, 2048), (11, 1000, 8192)])
def test_fp8_gemv_matches_dequantized_matmul(n, N, K):
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    from videollm_online_amd.engine import test_gemv_fp8
    g = torch.Generator().manual_seed(n * 1000 + N + K)
    x = torch.randn(n, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * K ** -0.5 * (1 + torch.rand(N, 1, generator=g) * 3)).bfloat16()     # rows of different scale
    q, s = quantize_fp8_per_channel(W.cuda())
    y = test_gemv_fp8(x.cuda(), q, s).cpu()
    Wd = q.cpu().float().double() * s.cpu().double()[:, None]
    ref = x.double() @ Wd.T
    err = (y.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()) * (K / 256) ** 0.5 + 1e-5, err
    # and the quantisation itself is what the oracle restates
    W2 = O.fp8_dequantized_weights({"model.layers.0.self_attn.q_proj.weight": W})["model.layers.0.self_attn.q_proj.weight"]
    assert torch.equal(W2.double(), Wd.float().double())


def _cfg(spec, **kw):
    from videollm_online_amd.engine import EngineConfig
    return EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                        num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                        rope_theta=spec.rope_theta, rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size,
                        kv_pool_tokens=2048, weight_dtype="fp8", **kw)


def _oracles(spec, w):
    wq = O.fp8_dequantized_weights(w)
    keep = {k for k in wq if k.endswith(O.FP8_STREAMED) and not k.startswith(("vision.", "connector."))}
    return O.LlamaOracle(spec, wq, torch.bfloat16, keep_fp32=keep), O.LlamaOracle(spec, wq, torch.float32)


def _steps(spec, ref, toks, seed):
    g = torch.Generator().manual_seed(seed + 100)
    H = spec.hidden_size
    frame = lambda: torch.randn(10, H, generator=g).bfloat16()
    return [torch.cat([ref.embed(torch.tensor(toks.start_ids)), frame()]),         # 45 tokens: 16-row chunks (no block path for fp8)
            torch.cat([ref.embed(torch.tensor([toks.interval_id])), frame()]),      # n = 11
            ref.embed(torch.tensor(toks.stream_generation_ids)),                    # n = 4
            ref.embed(torch.tensor([17])),                                         # n = 1
            torch.cat([ref.embed(torch.tensor([toks.eos_token_id] + toks.stream_prompt_ids)), frame()])]   # n = 13


def _check(tag, i, allr, rl, gl):
    e = (allr.float() - gl).abs().max().item()
    r = (rl.float() - gl).abs().max().item()
    scale = gl.abs().max().item()
    print(f"[{tag}] step {i}: engine err {e:.4g} ref err {r:.4g} scale {scale:.3g} | engine vs ref: {fmt(ulp_report(allr, rl))}")
    assert e <= 1.5 * r + 1e-3 * scale, f"{tag} step {i}: engine err {e} vs reference err {r}"


def test_fp8_llm_stream_parity_8b_width():
    from videollm_online_amd.engine import Engine
    spec = O.LLM_SPECS["llama-3-8b-2l"]
    w = O.init_llm_weights(spec, seed=6)
    toks = O.default_tokens(spec, seed=7, n_start=35)
    ref, gold = _oracles(spec, w)
    eng = Engine(_cfg(spec))
    eng.load_weights(w)                                        # bf16 in, quantised on the way (checkpoint.quantize_fp8_per_channel)
    eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    eng.finalize()
    bf16_bytes = 2 * sum(v.numel() for k, v in w.items() if k.endswith(O.FP8_STREAMED))
    assert eng.weight_bytes < 0.56 * bf16_bytes + 2 * w["model.embed_tokens.weight"].numel()      # the stream really is one byte per weight
    sess = eng.new_session()
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 6)):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = eng.llm_step(sess, x.cuda(), want_last=True, want_all=True)
        torch.cuda.synchronize()
        assert sess.get_seq_length() == len(rc) and torch.equal(last, allr[-1])
        _check("fp8 8b-2l", i, allr.cpu(), rl, gl)
    ids = torch.zeros(6, dtype=torch.long, device="cuda")
    n = eng.greedy_generate(sess, eng.embed(torch.tensor(toks.stream_generation_ids)), toks.eos_token_id, ids, force_len=5)
    assert n == 5 and ids[4].item() == toks.eos_token_id
    sess.close()
    eng.close()


def test_fp8_70b_width_tp8_logical_ranks():
    """configs[4]'s LLM half at its true layer shape: H 8192, I 28672, 64 q / 8 kv heads, sharded 8 ways (8 q heads + 1 kv head,
    3584 MLP columns, K = 1024 / 3584 row-wise slices per rank), fp8 weights, logical ranks on one GPU."""
    from videollm_online_amd.engine import TpGroup
    spec = O.LLM_SPECS["llama-3-70b-1l"]
    w = O.init_llm_weights(spec, seed=8)
    toks = O.default_tokens(spec, seed=7, n_start=35)
    ref, gold = _oracles(spec, w)
    grp = TpGroup(_cfg(spec), 8)
    grp.load_weights(w)
    grp.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    grp.finalize()
    sess = grp.new_session()
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 8)[:4]):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = grp.llm_step(sess, x.cuda(), want_last=True, want_all=True)
        torch.cuda.synchronize()
        assert sess.get_seq_length() == len(rc)
        _check("fp8 70b-1l tp8", i, allr.cpu(), rl, gl)
    sess.close()
    grp.close()


def test_fp8_rejected_where_the_image_cannot_be_built():
    """TinyLlama's down-proj (K = 5632 = 8 waves x 11 fragments) has an odd fragment count per wave: no two-fragment loads."""
    from videollm_online_amd.engine import Engine
    spec = O.LLM_SPECS["tinyllama-2l"]
    eng = Engine(_cfg(spec))
    eng.load_weights(O.init_llm_weights(spec, seed=5))
    with pytest.raises(RuntimeError, match="fp8 weight image"):
        eng.finalize()
    eng.close()
