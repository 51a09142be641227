"""GPU parity of the teacher-forced evaluation surface: joint_embed, per-row logit statistics,
KV fork / crop and LiveModel.stream_evaluate, through the C ABI, against the CPU oracle and the fixtures the reference's
own LiveMixin.stream_evaluate produced (tests/golden/eval_toy128.npz, oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from tests.parity_util import within_band

from oracle import vlo_oracle as O
from test_gpu_llm import _engine

pytestmark = pytest.mark.gpu


def test_joint_embed_matches_masked_assignment():
    spec = O.LLM_SPECS["toy"]
    w = O.init_llm_weights(spec, seed=0)
    eng = _engine(spec, w)
    V, H = spec.vocab_size, spec.hidden_size
    g = torch.Generator().manual_seed(1)
    for k, nv in ((7, 0), (64, 20), (1500, 430), (2049, 2049)):          # spans several 1024-wide scan rounds
        ids = torch.randint(0, V + 40, (k,), generator=g)                 # ids >= V are clamped (:38) ...
        ids[ids == V] = V - 1
        pos = torch.randperm(k, generator=g)[:nv]
        ids[pos] = V                                                      # ... except the placeholder itself
        rows = torch.randn(nv, H, generator=g).bfloat16()
        ref = O.joint_embed(O.LlamaOracle(spec, w), ids, rows if nv else None, V)
        out = eng.joint_embed(ids.cuda(), rows.cuda() if nv else None, V).cpu()
        assert torch.equal(out, ref), (k, nv)
    with pytest.raises(RuntimeError, match="placeholder positions"):
        eng.joint_embed(ids.cuda(), rows[:5].cuda(), V)


@pytest.mark.parametrize("n,V", [(5, 1024), (37, 2048)])
def test_logit_rows_match_torch(n, V):
    spec = O.LLM_SPECS["toy" if V == 1024 else "toy128"]
    eng = _engine(spec, O.init_llm_weights(spec, seed=0))
    g = torch.Generator().manual_seed(n)
    lg = (torch.randn(n, V, generator=g) * 3).bfloat16()
    lg[1, 100] = lg[1, 7] = lg[1].max() + 1                              # exact tie: the lower index wins
    lg[2] = (lg[2].float() * 0.01).bfloat16()                            # flat row: bf16 softmax merges many entries
    labels = torch.randint(0, V, (n,), generator=g)
    labels[0], labels[3] = -100, V + 5
    interval = 11
    st = {k: v.cpu() for k, v in eng.logit_rows(lg.cuda(), labels.cuda(), interval).items()}
    f = lg.float()
    np.testing.assert_allclose(st["lse"].numpy(), torch.logsumexp(f, -1).numpy(), rtol=1e-6, atol=1e-5)
    assert torch.equal(st["argmax"], lg.argmax(-1))
    want = torch.where((labels >= 0) & (labels < V), f.gather(1, labels.clamp(0, V - 1)[:, None])[:, 0], torch.zeros(n))
    assert torch.equal(st["label_logit"], want)
    sm = lg.softmax(-1)                                                  # bf16 in, bf16 out: the reference's :107
    p_err = (st["p_interval"] - sm[:, interval].float()).abs().max().item()
    assert p_err <= 2.0 ** -8 * sm[:, interval].float().max().item()    # within one bf16 ulp of torch's rounding
    agree = st["p_argmax"] == sm.argmax(-1)
    for r in (~agree).nonzero().view(-1).tolist():                       # expf ulp differences may move a tie break
        a, b = int(st["p_argmax"][r]), int(sm[r].argmax())
        assert abs(float(f[r, a] - f[r, b])) <= 2.0 ** -6 * abs(float(f[r, b])) + 1e-3, r
    assert agree.float().mean() > 0.9


def test_session_fork_and_crop():
    spec = O.LLM_SPECS["toy128"]
    w = O.init_llm_weights(spec, seed=3)
    eng = _engine(spec, w, kv_pool_tokens=4096)
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(700, spec.hidden_size, generator=g) * 0.05).bfloat16().cuda()
    a = eng.new_session()
    _, full = eng.llm_step(a, x, want_last=False, want_all=True)           # 700 positions = 3 pages
    scale = full.float().abs().max().item()

    def same(got, want, aligned):
        # a restart that replays the same rows through the SAME kernels with the same attention sub-blocks is bit-identical; otherwise the
        # projection path (prefill GEMMs for >= 256 tokens, 64-token block GEMM, 16-row GEMV) or the sub-block boundaries (and with
        # them the fp32 summation order of the attention splits) move -> bf16 noise only
        if aligned:
            return torch.equal(got, want)
        return (got.float() - want.float()).abs().max().item() <= 0.03 * scale

    for keep in (0, 1, 255, 256, 304, 699):
        b = a.fork(keep)
        assert len(b) == keep and len(a) == 700
        _, tail = eng.llm_step(b, x[keep:keep + 64], want_last=False, want_all=True)     # 64 tokens: the block path; `full` came out of the prefill path
        assert same(tail, full[keep:keep + 64], False), keep
        # ... and exactly what the SAME continuation gives on a cropped copy of the prefix (fork == crop, bit for bit)
        c = a.fork(min(700, keep + 100))
        c.crop(keep)
        _, tail_c = eng.llm_step(c, x[keep:keep + 64], want_last=False, want_all=True)
        assert torch.equal(tail, tail_c), keep
        b.close()
        c.close()
    k5 = a.read_kv(1, 0, 0, 0, 700).clone()
    # a FORK on a page / tile boundary continued through the same path (444 tokens: the prefill GEMMs again, the same attention tiles) is the
    # uninterrupted run bit for bit, and its KV pages are copies of the source's (a wrong page copy at the boundary would show in both)
    f = a.fork(256)
    assert torch.equal(f.read_kv(1, 0, 0, 0, 256), k5[:256])          # read_kv: [tokens][head_dim]
    _, again_f = eng.llm_step(f, x[256:], want_last=False, want_all=True)
    assert len(f) == 700 and torch.equal(again_f, full[256:])
    assert torch.equal(f.read_kv(1, 0, 0, 0, 700), k5) and torch.equal(f.read_kv(1, 1, 1, 0, 700), a.read_kv(1, 1, 1, 0, 700))
    f.close()
    a.crop(256)
    assert len(a) == 256
    _, again = eng.llm_step(a, x[256:], want_last=False, want_all=True)    # 444 tokens: the prefill path again, the same 64-query sub-blocks
    assert same(again, full[256:], True)
    assert torch.equal(a.read_kv(1, 0, 0, 0, 700), k5)
    a.crop(304)
    _, again = eng.llm_step(a, x[304:], want_last=False, want_all=True)    # sub-blocks start at 304 now: other attention splits
    assert len(a) == 700 and same(again, full[304:], False)
    a.crop(77)                                                             # releases two pages, keeps a partial one
    _, again = eng.llm_step(a, x[77:], want_last=False, want_all=True)
    assert len(a) == 700 and same(again, full[77:], False)
    with pytest.raises(RuntimeError):
        a.fork(701)
    with pytest.raises(RuntimeError):
        a.crop(-1)
    # forks draw their pages from the shared pool and give them back
    forks = [a.fork(700) for _ in range(3)]
    for f in forks:
        f.close()
    a.crop(0)
    assert len(a) == 0


def test_prefill_attention_shortcuts_are_bit_identical(monkeypatch):
    """attn_prefill_kernel skips the causal select on tiles that are fully visible to a wave and the accumulator rescale on tiles that moved
    no running maximum (wave-uniform decisions); VLO_ATTN_NOSKIP=1 takes the long way.  Same logits, bit for bit, over 6 slabs of key tiles."""
    spec = O.LLM_SPECS["toy128"]
    w = O.init_llm_weights(spec, seed=3)
    eng = _engine(spec, w, kv_pool_tokens=4096)
    x = (torch.randn(1500, spec.hidden_size, generator=torch.Generator().manual_seed(5)) * 0.05).bfloat16().cuda()
    a = eng.new_session()
    _, short = eng.llm_step(a, x, want_last=False, want_all=True)
    monkeypatch.setenv("VLO_ATTN_NOSKIP", "1")
    b = eng.new_session()
    _, long = eng.llm_step(b, x, want_last=False, want_all=True)
    assert torch.equal(short, long)
    assert torch.equal(a.read_kv(1, 0, 0, 0, 1500), b.read_kv(1, 0, 0, 0, 1500))
    eng.close()


@pytest.mark.parametrize("spec_name,n", [("toy128", 1500), ("llama-3-8b-2l", 1300)])
def test_prefill_attention_variants_are_bit_identical(monkeypatch, spec_name, n):
    """The prefill attention that ships — a two-group ping-pong (attn_prefill_pp_body.inc: softmax of tile t on one wave of a SIMD while its partner
    runs P.V of tile t and Q.K of tile t + 1, fragment reads batched ahead of the MFMAs) — against the lock-step kernel (VLO_ATTN_PF=0): same tiles,
    same operations per element — logits and appended K bit for bit, ragged last block included."""
    spec = O.LLM_SPECS[spec_name]
    w = O.init_llm_weights(spec, seed=3)
    eng = _engine(spec, w, kv_pool_tokens=4096)
    x = (torch.randn(n, spec.hidden_size, generator=torch.Generator().manual_seed(5)) * 0.05).bfloat16().cuda()
    outs = []
    for variant in ("0", "1"):
        monkeypatch.setenv("VLO_ATTN_PF", variant)
        a = eng.new_session()
        _, lg = eng.llm_step(a, x[:n - 401], want_last=False, want_all=True)          # a block from an empty cache ...
        _, lg2 = eng.llm_step(a, x[n - 401:], want_last=False, want_all=True)         # ... and a ragged one over a prefix (pos0 > 0)
        torch.cuda.synchronize()
        outs.append((lg.clone(), lg2.clone(), a.read_kv(spec.num_layers - 1, 0, 0, 0, n).clone()))
        a.close()
    for v in (1,):
        for i in range(3):
            assert torch.equal(outs[0][i], outs[v][i]), f"VLO_ATTN_PF={v} differs from the lock-step kernel (output {i})"
    eng.close()


def test_full_logits_forward_matches_oracle():
    """model(input_ids=, frames=) returns every row (the evaluation path), 3-way checked against fp32 gold."""
    from videollm_online_amd.modeling_live import LiveModel
    spec = O.LLM_SPECS["toy128"]
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    ids, labels, T = O.synthetic_eval_sample(spec, toks, [(2, 3, 2), (3, 2, 3)])
    feats = torch.randn(T, 10, spec.vision_hidden_size, generator=torch.Generator().manual_seed(9))
    eng = _engine(spec, w)
    model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
    out = model(input_ids=ids[None].cuda(), frames=feats.cuda())
    assert out.logits.shape == (1, ids.numel(), spec.vocab_size) and len(out.past_key_values) == ids.numel()
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    lr, _ = ref.forward(O.joint_embed(ref, ids, O.connector(ref.W, feats.bfloat16()).view(-1, spec.hidden_size), spec.vocab_size), None)
    lgd, _ = gold.forward(O.joint_embed(gold, ids, O.connector(gold.W, feats).view(-1, spec.hidden_size), spec.vocab_size), None)
    e = (out.logits[0].float().cpu() - lgd).abs().max().item()
    r = (lr.float() - lgd).abs().max().item()
    assert within_band(e, r, 1e-3 * lgd.abs().max().item(), "test_gpu_eval.py:167"), (e, r)


def test_stream_evaluate_matches_reference_fixture(golden_dir):
    from videollm_online_amd.modeling_live import LiveModel
    g = np.load(os.path.join(golden_dir, "eval_toy128.npz"))
    spec = O.LLM_SPECS["toy128"]
    for c in range(int(g["n_cases"])):
        w, toks, ids, labels, feats, thr = O.eval_case_from_golden(g, c, spec)
        eng = _engine(spec, w)
        model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
        out = model.stream_evaluate(ids[None].cuda(), labels[None].cuda(), feats.cuda(), frame_token_interval_threshold=thr).cpu().numpy()
        ref_bf16, gold = g[f"c{c}_bf16"], g[f"c{c}_fp32"]
        # every decision in the fixture has a >= 0.3 logit margin (oracle/make_golden.py), so the discrete metrics are exact
        np.testing.assert_allclose(out[1:], ref_bf16[1:], rtol=0, atol=1e-6, err_msg=f"case {c}")
        # perplexity: bf16 logits noise; engine no farther from fp32 gold than the reference's own bf16 path (+1 %)
        assert abs(out[0] - gold[0]) <= 1.5 * abs(ref_bf16[0] - gold[0]) + 0.01 * gold[0], (c, out[0], ref_bf16[0], gold[0])
        eng.close()


def test_stream_evaluate_slabs_and_long_dialogue():
    """A dialogue longer than one slab and than two KV pages: slab size must not change the metrics."""
    from videollm_online_amd.modeling_live import LiveModel
    spec = O.LLM_SPECS["toy128"]
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    turns = [(6, 4, 6), (5, 3, 5), (7, 5, 7), (4, 4, 4), (6, 2, 6), (5, 3, 5), (8, 4, 8), (6, 3, 6), (7, 2, 7), (5, 4, 5)]
    ids, labels, T = O.synthetic_eval_sample(spec, toks, turns)
    assert ids.numel() > 600
    feats = torch.randn(T, 10, spec.vision_hidden_size, generator=torch.Generator().manual_seed(4))
    eng = _engine(spec, w)
    model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
    a = model.stream_evaluate(ids[None].cuda(), labels[None].cuda(), feats.cuda())
    orig = model._row_stats
    model._row_stats = lambda *args, **kw: orig(*args, slab=100, **{k: v for k, v in kw.items() if k != "slab"})
    b = model.stream_evaluate(ids[None].cuda(), labels[None].cuda(), feats.cuda())
    # slab boundaries move the 64-token block / 16-query sub-chunk boundaries, i.e. fp32 summation orders: the discrete
    # metrics must not move, the perplexity only by bf16 noise
    assert torch.equal(a[1:], b[1:]), (a, b)
    assert abs(float(a[0]) - float(b[0])) <= 0.02 * float(a[0]), (a, b)
    ref = O.LlamaOracle(spec, w, torch.bfloat16)
    fe = O.connector(ref.W, feats.bfloat16()).view(-1, spec.hidden_size)
    d = {}
    want = O.stream_evaluate(ref, ids, labels, fe, v_placeholder_id=spec.vocab_size, interval_id=toks.interval_id,
                             eos_token_id=toks.eos_token_id, detail=d)
    if min(d["margins"]) > 0.25:                                         # only clear-margin dialogues pin the discrete metrics
        np.testing.assert_allclose(a.cpu().numpy()[1:], want.numpy()[1:], atol=1e-6)
    np.testing.assert_allclose(a.cpu().numpy()[0], want.numpy()[0], rtol=0.05)
