"""Host bookkeeping of LiveModel.stream_evaluate (videollm-online_amd/modeling_live.py) on the CPU: the device calls it
makes (joint_embed, llm_step with all-row logits, logit_rows, session fork) are served by a stand-in engine built on the
oracle's arithmetic, so the per-turn logic of the PRODUCT code — turn splitting, lm / stream masks, early / on-time / late
branches with the forked KV prefix, fluency — is checked against the fixtures the reference class produced
(tests/golden/eval_toy128.npz) without a GPU.  The HIP side of the same calls is checked in tests/test_gpu_eval.py."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import videollm_online_amd  # noqa: F401  (package shim)
from oracle import vlo_oracle as O
from videollm_online_amd.modeling_live import LiveModel


class _Sess:
    def __init__(self, cache=None):
        self.cache = cache
        self.closed = False

    def __len__(self):
        return 0 if self.cache is None else len(self.cache)

    def fork(self, n):
        assert not self.closed and n <= len(self)
        return _Sess(O.cache_prefix(self.cache, n))

    def close(self):
        self.closed = True


class _OracleEngine:
    """Same method surface LiveModel uses on Engine, computed with the oracle on the CPU."""

    def __init__(self, spec, weights):
        self.m = O.LlamaOracle(spec, weights, torch.bfloat16)
        self.cfg = SimpleNamespace(hidden_size=spec.hidden_size, vocab_size=spec.vocab_size, frame_num_tokens=10,
                                   vision_hidden_size=spec.vision_hidden_size)
        self.device = torch.device("cpu")

    def new_session(self):
        return _Sess()

    def connector(self, feats):
        return O.connector(self.m.W, feats.to(torch.bfloat16)).view(-1, self.cfg.hidden_size)

    def joint_embed(self, ids, frame_rows, v_id):
        return O.joint_embed(self.m, ids, frame_rows, v_id)

    def llm_step(self, sess, embeds, want_last=True, want_all=False):
        logits, sess.cache = self.m.forward(embeds, sess.cache)
        return (logits[-1] if want_last else None), (logits if want_all else None)

    def logit_rows(self, logits, labels, interval_id):
        f = logits.float()
        n, V = f.shape
        sm = logits.softmax(-1)                                            # bf16 in, bf16 out (models/modeling_live.py:107)
        lab = torch.full((n,), -1) if labels is None else labels
        ok = (lab >= 0) & (lab < V)
        return dict(lse=torch.logsumexp(f, -1), argmax=logits.argmax(-1),
                    label_logit=torch.where(ok, f.gather(1, lab.clamp(0, V - 1)[:, None])[:, 0], torch.zeros(n)),
                    p_interval=sm[:, interval_id].float(), p_argmax=sm.argmax(-1))


@pytest.mark.parametrize("slab", [2048, 37])
def test_stream_evaluate_host_logic_matches_reference_fixture(golden_dir, slab):
    g = np.load(os.path.join(golden_dir, "eval_toy128.npz"))
    spec = O.LLM_SPECS["toy128"]
    late_cases = 0
    for c in range(int(g["n_cases"])):
        w, toks, ids, labels, feats, thr = O.eval_case_from_golden(g, c, spec)
        eng = _OracleEngine(spec, w)
        model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
        orig = model._row_stats
        model._row_stats = lambda *a, **k: orig(*a, **{**k, "slab": slab})
        out = model.stream_evaluate(ids[None], labels[None], feats, frame_token_interval_threshold=thr).numpy()
        ref, gold = g[f"c{c}_bf16"], g[f"c{c}_fp32"]
        np.testing.assert_allclose(out[1:], ref[1:], rtol=0, atol=1e-6, err_msg=f"case {c}")
        # perplexity: the product averages the per-row cross entropies in fp32, the reference's bf16 path rounds log_softmax
        # and the mean to bf16 — it must sit no farther from fp32 gold than the reference's own bf16 result (+1 %)
        assert abs(out[0] - gold[0]) <= 1.5 * abs(ref[0] - gold[0]) + 0.01 * gold[0], (c, out[0], ref[0], gold[0])
        late_cases += int((g[f"c{c}_turns"][:, 2] < 0).any())
    assert late_cases >= 2                                                  # the forked-prefix branch really ran


def test_stream_evaluate_rejects_batches_and_counts_frames():
    spec = O.LLM_SPECS["toy128"]
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec, seed=7, n_start=19)
    eng = _OracleEngine(spec, w)
    model = LiveModel(eng, eos_token_id=toks.eos_token_id, frame_token_interval_id=toks.interval_id)
    ids, labels, T = O.synthetic_eval_sample(spec, toks, [(2, 3, 2)])
    feats = torch.randn(T, 10, spec.vision_hidden_size)
    with pytest.raises(AssertionError):
        model.stream_evaluate(torch.stack([ids, ids]), torch.stack([labels, labels]), feats)
    with pytest.raises(ValueError):                                        # one frame short: placeholder / embedding count mismatch
        model.stream_evaluate(ids[None], labels[None], feats[:-1])
    out = model.stream_evaluate(ids[None], torch.full_like(labels, -100)[None], feats)
    assert out.tolist() == [1.0, 0.0, 1.0, 1.0]                           # nothing to learn: the reference's neutral values (:164-167)
