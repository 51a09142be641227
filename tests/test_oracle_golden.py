"""Pin the oracle (oracle/vlo_oracle.py) against fixtures produced by the REFERENCE's own
classes (oracle/make_golden.py, run in the build container where /root/reference exists).
CPU only; no /root/reference access at run time."""
import os

import numpy as np
import pytest
import torch

from oracle import vlo_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_vit_tokens_match_reference(golden_dir):
    g = _load(golden_dir, "vit_toy.npz")
    vspec = O.VIT_SPECS["toy"]
    W = O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(3, vspec.image_size, seed=1234)
    assert np.array_equal(np.frombuffer(frames.numpy().tobytes()[:64], dtype=np.uint8), g["frames_sha"])
    tok = O.siglip_vision_encode(W, vspec, frames)
    assert tok.shape == (3, 10, vspec.hidden_size)
    np.testing.assert_allclose(tok.numpy(), g["tokens"], rtol=0, atol=2e-5)


def _scripted_stream(name, seed, dtype):
    """Same script as oracle/make_golden.py, on the oracle."""
    spec = O.LLM_SPECS[name]
    vspec = O.VIT_SPECS["toy"]
    toks = O.default_tokens(spec, seed=7, n_start=19)
    w = O.init_llm_weights(spec, seed=seed, dtype=torch.bfloat16)
    vit_w = O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(3, vspec.image_size, seed=1234)
    m = O.LlamaOracle(spec, w, dtype)
    rec = {}
    fe = m.visual_embed(vit_w, vspec, frames)
    rec["frame_embeds"] = fe.float().numpy()
    fe = fe.split(10)
    cache, step = None, 0
    x = torch.cat([m.embed(torch.tensor(toks.start_ids)), fe[0]])
    lg, cache = m.forward(x, cache)
    rec[f"logits{step}"] = lg[-1].float().numpy(); step += 1
    x = torch.cat([m.embed(torch.tensor([toks.interval_id])), fe[1]])
    lg, cache = m.forward(x, cache)
    rec[f"logits{step}"] = lg[-1].float().numpy(); step += 1
    tok, p_int = O.stream_sample(lg[-1].clone(), toks.interval_id, 0.725)
    rec["stream_tok"], rec["p_interval"] = tok, p_int
    out, cache = O.fast_greedy_generate(m, m.embed(torch.tensor(toks.stream_generation_ids)), cache,
                                        toks.eos_token_id, max_new=8)
    rec["gen_ids"] = np.array(out)
    last = [out[-1]] + toks.stream_prompt_ids
    x = torch.cat([m.embed(torch.tensor(last)), fe[2]])
    lg, cache = m.forward(x, cache)
    rec[f"logits{step}"] = lg[-1].float().numpy()
    rec["cache_len"] = len(cache)
    return rec


@pytest.mark.parametrize("name,seed", [("toy", 0), ("toy128", 3)])
@pytest.mark.parametrize("dt_name", ["bf16", "fp32"])
def test_llm_scripted_stream_matches_reference(golden_dir, name, seed, dt_name):
    g = _load(golden_dir, f"llm_{name}_{dt_name}.npz")
    dtype = torch.bfloat16 if dt_name == "bf16" else torch.float32
    rec = _scripted_stream(name, seed, dtype)
    assert rec["cache_len"] == int(g["cache_len"])
    if dt_name == "bf16":
        # same torch CPU ops, same order, same rounding points => bit-exact
        np.testing.assert_array_equal(rec["frame_embeds"], g["frame_embeds"])
        for s in range(3):
            np.testing.assert_array_equal(rec[f"logits{s}"], g[f"logits{s}"])
    else:
        # fp32: the oracle spells ViT attention out (matmul/softmax/matmul) where HF calls sdpa;
        # summation order differs at the 1e-6 level
        np.testing.assert_allclose(rec["frame_embeds"], g["frame_embeds"], rtol=0, atol=2e-5)
        for s in range(3):
            np.testing.assert_allclose(rec[f"logits{s}"], g[f"logits{s}"], rtol=0, atol=2e-4)
    assert rec["gen_ids"].tolist() == g["gen_ids"].tolist()
    assert rec["stream_tok"] == int(g["stream_tok"])
    assert rec["p_interval"] == pytest.approx(float(g["p_interval"]), rel=1e-3)


@pytest.mark.parametrize("dt_name,dtype", [("bf16", torch.bfloat16), ("fp32", torch.float32)])
def test_stream_evaluate_matches_reference(golden_dir, dt_name, dtype):
    """oracle.stream_evaluate vs the reference's LiveMixin.stream_evaluate (models/modeling_live.py:44-168) on four
    teacher-forced samples that together take every branch (on-time, early, late with/without a later hit, no room,
    last turn).  The fixture generator checked bit-equality in the build container; here the discrete metrics must be
    exact and the perplexity may move by CPU-BLAS accumulation order only."""
    g = _load(golden_dir, "eval_toy128.npz")
    spec = O.LLM_SPECS["toy128"]
    for c in range(int(g["n_cases"])):
        w, toks, ids, labels, feats, thr = O.eval_case_from_golden(g, c, spec)
        m = O.LlamaOracle(spec, w, dtype)
        fe = O.connector(m.W, feats.to(dtype)).view(-1, spec.hidden_size)
        d = {}
        out = O.stream_evaluate(m, ids, labels, fe, v_placeholder_id=spec.vocab_size, interval_id=toks.interval_id,
                                eos_token_id=toks.eos_token_id, threshold=thr, detail=d).numpy()
        ref = g[f"c{c}_{dt_name}"]
        np.testing.assert_allclose(out[0], ref[0], rtol=2e-2 if dt_name == "bf16" else 1e-3)
        np.testing.assert_allclose(out[1:], ref[1:], rtol=0, atol=1e-6)
        if dt_name == "bf16":
            turns = np.array([[t[0], -1 if t[1] is None else t[1], 99 if t[2] is None else t[2]] for t in d["turns"]])
            assert np.array_equal(turns, g[f"c{c}_turns"])


def test_joint_embed_and_cache_prefix_rules():
    spec = O.LLM_SPECS["toy"]
    m = O.LlamaOracle(spec, O.init_llm_weights(spec, seed=0), torch.bfloat16)
    V, H = spec.vocab_size, spec.hidden_size
    ids = torch.tensor([5, V, V, 9, V + 7])                 # V = placeholder; V+7 is clamped to V-1 before lookup (:38)
    fe = torch.arange(2 * H, dtype=torch.float32).view(2, H).to(torch.bfloat16)
    x = O.joint_embed(m, ids, fe, V)
    assert torch.equal(x[1], fe[0]) and torch.equal(x[2], fe[1])
    assert torch.equal(x[0], m.embed(torch.tensor([5]))[0]) and torch.equal(x[4], m.embed(torch.tensor([V - 1]))[0])
    with pytest.raises(ValueError):
        O.joint_embed(m, ids, fe[:1], V)
    lg, cache = m.forward(x, None)
    pre = O.cache_prefix(cache, 3)
    assert len(pre) == 3 and len(cache) == 5
    lg2, _ = m.forward(x[3:], pre)                          # re-running the tail on the prefix reproduces the logits
    assert torch.allclose(lg2.float(), lg[3:].float(), atol=0.1)
    assert len(cache) == 5                                   # source cache untouched
