"""The engine's LLM-path SOURCES (csrc/{gemv,prefill,llm_ops,engine,tp}.hip) compiled for the CPU against the
HIP-on-threads shim of tests/hip_emul/ and driven through the same C ABI as the product, at toy sizes, against the oracle.

What this pins WITHOUT a GPU: launch sequencing and argument wiring of the host code, index arithmetic of every kernel
(packed weight fragments, MFMA register maps, paged KV, split-KV attention and its combine, epilogues), rounding points.
What it cannot: the GPU memory model, the matrix core's internal summation order, performance.  It is how the code paths
written without GPU time (VLO_FUSED_ROWS pipeline, the tp_reduce_norm refactor, the peer-to-peer TP exchange between
logical ranks) were checked before their first run on hardware; the `-m gpu` suite remains the parity gate.

The emulation is slow (every GPU thread is an OS thread): the default set below takes ~2-3 minutes including the one-off
build of the emulated library; VLO_EMUL_FULL=1 adds the longer cases."""
import os

import pytest
import torch

from oracle import vlo_oracle as O

FULL = os.environ.get("VLO_EMUL_FULL") == "1"
TINY = O.LlmSpec(128, 192, 2, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=128)     # K = 128 / 192: 256- and 64-thread GEMV blocks


@pytest.fixture(scope="module")
def E():
    from tests.hip_emul import emul_engine
    if emul_engine.lib() is None:
        pytest.skip("no clang++ to build the emulated library")
    return emul_engine


def _three_way(name, i, out, rl, gl):
    e = (out.float() - gl).abs().max().item()
    r = (rl.float() - gl).abs().max().item()
    scale = gl.abs().max().item()
    print(f"[emul {name}] step {i}: engine err {e:.4g} ref-bf16 err {r:.4g} scale {scale:.3g}")
    assert e <= 1.5 * r + 1e-3 * scale, f"{name} step {i}: {e} vs {r}"


def _steps(spec, ref, toks, seed, lens):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in lens:
        ids = torch.tensor([toks.interval_id] + toks.stream_prompt_ids + toks.stream_generation_ids)[:max(1, n - 10)]
        rows = [ref.embed(ids)]
        if n > len(ids):
            rows.append(torch.randn(n - len(ids), spec.hidden_size, generator=g).bfloat16())
        out.append(torch.cat(rows)[:n])
    return out


def test_default_pipeline_matches_oracle(E):
    """run_chunk as shipped (7 launches per layer) on the 'toy' model (4 heads over 2 kv heads: the 2-heads-per-wave attention)."""
    spec = O.LLM_SPECS["toy"]
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 0, [11, 1] + ([13, 16] if FULL else []))):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = eng.llm_step(s, x)
        assert eng.session_len(s) == len(rc) and torch.equal(last, allr[-1])
        _three_way("toy default", i, allr, rl, gl)
    tok, p = eng.stream_sample(s, 0.725, toks.interval_id)
    rt, rp = O.stream_sample(rl[-1].clone(), toks.interval_id, 0.725)
    top2 = rl[-1].float().topk(2).values
    assert tok == rt or (top2[0] - top2[1]).item() < 0.12
    eng.close()


def test_fused_rows_pipeline(E, monkeypatch):
    """VLO_FUSED_ROWS (run_chunk_fused: norms on the operand loads, whole-K down-proj, no add_rmsnorm launch) against the
    oracle and against the default pipeline, same engine, two sessions."""
    spec = TINY
    w = O.init_llm_weights(spec, seed=5)
    toks = O.default_tokens(spec)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    monkeypatch.setenv("VLO_FUSED_ROWS", "16")
    fused = eng.new_session()
    monkeypatch.delenv("VLO_FUSED_ROWS")
    plain = eng.new_session()
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 1, [11, 1, 1] + ([16, 4] if FULL else []))):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        lf, af = eng.llm_step(fused, x)
        lp, ap = eng.llm_step(plain, x)
        assert eng.session_len(fused) == eng.session_len(plain) == len(rc)
        assert torch.equal(lf, af[-1])
        _three_way("tiny fused", i, af, rl, gl)
        _three_way("tiny default", i, ap, rl, gl)
        d = (af.float() - ap.float()).abs().max().item()
        assert d <= 0.5 * (rl.float() - gl).abs().max().item() + 1e-3 * gl.abs().max().item(), f"fused vs default: {d}"
    # the live path asks for the last row only (final norm on the lm_head operand load, row offset into the sum-of-squares partials)
    x = _steps(spec, ref, toks, 2, [11])[0]
    rl, rc = ref.forward(x, rc)
    gl, _ = gold.forward(x, gc)
    lf, _ = eng.llm_step(fused, x, want_all=False)
    _three_way("tiny fused last-row", 99, lf[None], rl[-1:], gl[-1:])
    eng.close()


@pytest.mark.parametrize("p2p", [False, True], ids=["sum-kernel", "p2p"])
def test_tensor_parallel_logical_ranks(E, p2p):
    """tp_chunk / tp_reduce_norm with T = 2 logical ranks: the sum-kernel exchange (the refactored default) and the
    peer-to-peer mailbox exchange (publish + collect kernels, p2p logits gather)."""
    spec = TINY
    w = O.init_llm_weights(spec, seed=6)
    toks = O.default_tokens(spec)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    grp = E.EmulTpGroup(spec, 2, w, O.rope_inv_freq(spec.head_dim, spec.rope_theta), p2p=p2p)
    assert grp.p2p_status()["enabled"] == int(p2p)
    s = grp.new_session()
    rc = gc = None
    lens = [11, 1] + ([19] if FULL else [])          # 19 = two chunks: the first one wants no logits (one exchange fewer)
    for i, x in enumerate(_steps(spec, ref, toks, 3, lens)):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = grp.llm_step(s, x)
        assert grp.session_len(s) == len(rc) and torch.equal(last, allr[-1])
        _three_way(f"tiny tp2 {'p2p' if p2p else 'sum'}", i, allr, rl, gl)
    if p2p:
        assert grp.p2p_status()["timed_out"] == 0
    grp.close()
