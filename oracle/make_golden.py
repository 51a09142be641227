"""Generate tests/golden/*.npz by running the REFERENCE's own classes on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/vlo_oracle.py header).  Runs only in the build
container, where /root/reference exists; the fixtures it writes are committed so the
GPU box (no /root/reference there) can check the oracle and the engine against them.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Import recipe follows SURVEY.md Appendix A: transformers' lazy symbols are resolved
first, then shims for ``peft`` (models/modeling_live.py:2 — never executed when
is_training=False and no checkpoint) and ``torchvision.transforms.functional.normalize``
(models/vision_live.py:4,12) are installed, then the reference's LiveLlamaForCausalLM,
fast_greedy_generate and _siglip_vision_encode are imported unchanged.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = os.environ.get("VLO_REFERENCE", "/root/reference")


def import_reference():
    from transformers import (TrainingArguments, HfArgumentParser, Trainer, AutoModelForCausalLM,  # noqa: F401
                              AutoTokenizer, AutoModel, LlamaForCausalLM, LlamaConfig, Cache,
                              SiglipVisionConfig, SiglipVisionModel)
    peft = types.ModuleType("peft")
    peft.LoraConfig = peft.PeftModel = type("X", (), {"__init__": lambda s, *a, **k: None})
    peft.get_peft_model = lambda m, c: m
    sys.modules["peft"] = peft
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def normalize(t, mean, std):
        m = torch.as_tensor(mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        return (t - m) / s

    tvf.normalize = normalize
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    sys.path.insert(0, REF)
    from models.live_llama import LiveLlamaConfig, LiveLlamaForCausalLM
    from models.modeling_live import fast_greedy_generate
    from models.vision_live import _siglip_vision_encode
    return LiveLlamaConfig, LiveLlamaForCausalLM, fast_greedy_generate, _siglip_vision_encode, SiglipVisionConfig, SiglipVisionModel


def build_ref_llm(LiveLlamaConfig, LiveLlamaForCausalLM, spec, weights, interval_id, eos_id, dtype):
    cfg = LiveLlamaConfig(
        hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
        num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
        rms_norm_eps=spec.rms_eps, rope_parameters={"rope_type": "default", "rope_theta": spec.rope_theta},
        vision_hidden_size=spec.vision_hidden_size, v_placeholder="<v>", stream_loss_weight=1.0, frame_token_cls=True,
        frame_token_pooled=[3, 3], frame_num_tokens=10, v_placeholder_id=spec.vocab_size,
        frame_token_interval_id=interval_id, eos_token_id=eos_id, attn_implementation="sdpa",
        tie_word_embeddings=False, max_position_embeddings=8192)
    model = LiveLlamaForCausalLM(cfg)
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in weights.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "vision" in m for m in missing), missing
    model = model.to(dtype).eval()
    # ``from_pretrained(torch_dtype=...)`` (models/modeling_live.py:197) keeps the non-persistent
    # RoPE ``inv_freq`` buffer in fp32; a blanket ``.to(bf16)`` would round it.  Restore it with
    # the class's own initialiser so the harness matches the real load path.
    rot = model.model.rotary_emb
    inv, _ = rot.compute_default_rope_parameters(model.config)
    rot.inv_freq = inv.float()
    rot.original_inv_freq = inv.float().clone()
    return model


def build_ref_vit(SiglipVisionConfig, SiglipVisionModel, vspec, vit_w):
    cfg = SiglipVisionConfig(hidden_size=vspec.hidden_size, intermediate_size=vspec.intermediate_size,
                             num_hidden_layers=vspec.num_layers, num_attention_heads=vspec.num_heads,
                             image_size=vspec.image_size, patch_size=vspec.patch_size, layer_norm_eps=vspec.ln_eps)
    vit = SiglipVisionModel(cfg).eval()
    sd = {k[len("vision."):]: v for k, v in vit_w.items()}
    vit.load_state_dict(sd, strict=True)
    return vit


@torch.no_grad()
def main():
    from oracle import vlo_oracle as O
    (LiveLlamaConfig, LiveLlamaForCausalLM, ref_fast_greedy_generate, ref_siglip_vision_encode,
     SiglipVisionConfig, SiglipVisionModel) = import_reference()
    torch.set_num_threads(8)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)

    # ---------------- (i) ViT + token selection, reference function, fp32 (CPU path) ----------------
    vspec = O.VIT_SPECS["toy"]
    vit_w = O.init_vit_weights(vspec, seed=1)
    vit = build_ref_vit(SiglipVisionConfig, SiglipVisionModel, vspec, vit_w)
    frames = O.synthetic_frames(3, vspec.image_size, seed=1234)
    ref_tokens = ref_siglip_vision_encode(vit, frames, frame_token_cls=True, frame_token_pooled=[3, 3])
    np.savez_compressed(os.path.join(out_dir, "vit_toy.npz"), tokens=ref_tokens.numpy(),
                        frames_sha=np.frombuffer(frames.numpy().tobytes()[:64], dtype=np.uint8))
    print("vit_toy", tuple(ref_tokens.shape), float(ref_tokens.abs().mean()))

    # ---------------- (ii)+(iii) LLM scripted stream on the reference model class -------------------
    for name, seed in (("toy", 0), ("toy128", 3)):
        spec = O.LLM_SPECS[name]
        toks = O.default_tokens(spec, seed=7, n_start=19)
        w = O.init_llm_weights(spec, seed=seed, dtype=torch.bfloat16)
        for dt_name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            model = build_ref_llm(LiveLlamaConfig, LiveLlamaForCausalLM, spec, w, toks.interval_id, toks.eos_token_id, dt)
            model.vision_encoder = vit
            from functools import partial
            model.vision_encode = partial(ref_siglip_vision_encode, frame_token_cls=True, frame_token_pooled=[3, 3])
            rec = {}
            # (iii) visual_embed through the reference LiveMixin (autocast is a no-op on CPU)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fe = model.visual_embed(frames)                       # [3*10, H]
            rec["frame_embeds"] = fe.float().numpy()
            fe = fe.split(10)
            emb = model.get_input_embeddings()
            past = None
            step = 0
            # first frame step: start_ids + 10 frame tokens (demo/inference.py:61-69)
            x = torch.cat([emb(torch.tensor([toks.start_ids])), fe[0][None]], dim=1)
            o = model(inputs_embeds=x, use_cache=True, past_key_values=past)
            past = o.past_key_values
            rec[f"logits{step}"] = o.logits[0, -1].float().numpy(); step += 1
            # steady frame step: interval + frame (n = 11)
            x = torch.cat([emb(torch.tensor([[toks.interval_id]])), fe[1][None]], dim=1)
            o = model(inputs_embeds=x, use_cache=True, past_key_values=past)
            past = o.past_key_values
            rec[f"logits{step}"] = o.logits[0, -1].float().numpy(); step += 1
            # streaming sampler exactly as demo/inference.py:76-79
            next_score = o.logits[:, -1:].softmax(dim=-1)
            rec["p_interval"] = np.array(float(next_score[0, 0, toks.interval_id]))
            if next_score[:, :, toks.interval_id] < 0.725:
                next_score[:, :, toks.interval_id].zero_()
            rec["stream_tok"] = np.array(int(next_score.argmax(dim=-1)))
            # response: "]\nAssistant:" then reference fast_greedy_generate, capped at 8 new tokens
            gen_in = emb(torch.tensor([toks.stream_generation_ids]))
            ids = torch.zeros(1, 8, dtype=torch.long)
            out_ids, past = ref_fast_greedy_generate(model=model, inputs_embeds=gen_in, past_key_values=past,
                                                     eos_token_id=toks.eos_token_id, inplace_output_ids=ids)
            rec["gen_ids"] = out_ids[0].numpy().copy()
            # post-response frame step: [last, stream_prompt_ids..] + frame (n = 13) (:63-64)
            last = [int(out_ids[0, -1])] + toks.stream_prompt_ids
            x = torch.cat([emb(torch.tensor([last])), fe[2][None]], dim=1)
            o = model(inputs_embeds=x, use_cache=True, past_key_values=past)
            past = o.past_key_values
            rec[f"logits{step}"] = o.logits[0, -1].float().numpy(); step += 1
            rec["cache_len"] = np.array(past.get_seq_length())
            np.savez_compressed(os.path.join(out_dir, f"llm_{name}_{dt_name}.npz"), **rec)
            print(f"llm_{name}_{dt_name}", "gen", rec["gen_ids"].tolist(), "stream_tok", int(rec["stream_tok"]),
                  "p_int", float(rec["p_interval"]), "cache", int(rec["cache_len"]))

    make_eval_golden(O, LiveLlamaConfig, LiveLlamaForCausalLM, out_dir)


# ---------------- (iv) stream_evaluate on the reference class (models/modeling_live.py:44-168) ----------------
EVAL_TURNS = [(3, 5, 3), (4, 4, 4), (1, 3, 1), (3, 3, 3), (2, 3, 2)]      # (frames, response length, learnt frames) per turn
EVAL_MIN_MARGIN = 0.3


def eval_case(O, spec, toks, w0, fseed, alpha):
    """One teacher-forced sample.  Random-init weights never pick the interval token, so the lm_head row of the interval
    id is replaced by alpha * mean(rows of the tokens the model predicts at frame ends): some frames then stay silent and
    the late-reply branches (:116-148) run.  Labels of the first text positions of every turn are set to the model's own
    (clear-margin) predictions so lm_correctness / fluency are not identically zero."""
    ids, labels, T = O.synthetic_eval_sample(spec, toks, EVAL_TURNS)
    g = torch.Generator().manual_seed(fseed)
    feats = torch.randn(T, 10, spec.vision_hidden_size, generator=g)
    orc = O.LlamaOracle(spec, w0, torch.bfloat16)
    fe = O.connector(orc.W, feats.to(torch.bfloat16)).view(-1, spec.hidden_size)
    lg, _ = orc.forward(O.joint_embed(orc, ids, fe, spec.vocab_size), None)
    at_v = ids == spec.vocab_size
    A = lg[at_v].argmax(-1)[9::10]
    row = (alpha * w0["lm_head.weight"][A].float().mean(0)).to(torch.bfloat16)
    w = dict(w0)
    w["lm_head.weight"] = w0["lm_head.weight"].clone()
    w["lm_head.weight"][toks.interval_id] = row
    orc = O.LlamaOracle(spec, w, torch.bfloat16)
    lg, _ = orc.forward(O.joint_embed(orc, ids, fe, spec.vocab_size), None)
    # teach the labels a few correct answers per turn
    stops = ((ids == toks.eos_token_id).nonzero().view(-1) + 1).tolist()
    starts = [0] + stops[:-1]
    for t, (a, b) in enumerate(zip(starts, stops)):
        text = ((labels[a:b] != -100) & ~at_v[a:b]).nonzero().view(-1) + a
        want = t % 3                                   # 0, 1, 2 leading correct tokens
        for j, pos in enumerate(text.tolist()):
            top = lg[pos].float().topk(2)
            if j < want and float(top.values[0] - top.values[1]) >= 0.5:
                labels[pos] = int(top.indices[0])
            else:
                if float(top.values[0] - lg[pos, labels[pos]].float()) < 0.5:      # make the miss a clear one
                    labels[pos] = int(lg[pos].float().argmin())
                break
    return ids, labels, feats, w, row


def trim_shim(self, past_key_values, start, stop):
    """models/modeling_live.py:170-171 iterates the cache as legacy (keys, values) tuples and returns a list of lists; on
    transformers 5.x a DynamicCache iterates differently and lists are no longer accepted as ``past_key_values``
    (SURVEY.md §8f-4 notes the breakage).  Same operation on the 5.x cache object: a NEW cache whose layers hold
    ``[:, :, start:stop]`` of the source; the source is untouched."""
    from transformers import DynamicCache
    out = DynamicCache(config=self.config)
    for i, layer in enumerate(past_key_values.layers):
        out.update(layer.keys[:, :, start:stop], layer.values[:, :, start:stop], i)
    return out


@torch.no_grad()
def make_eval_golden(O, LiveLlamaConfig, LiveLlamaForCausalLM, out_dir):
    LiveLlamaForCausalLM.trim_past_key_values = trim_shim
    spec = O.LLM_SPECS["toy128"]
    toks = O.default_tokens(spec, seed=7, n_start=19)
    w0 = O.init_llm_weights(spec, seed=3, dtype=torch.bfloat16)
    kw = dict(v_placeholder_id=spec.vocab_size, interval_id=toks.interval_id, eos_token_id=toks.eos_token_id)
    want = {"hit+", "hit0", "late-hit", "late-none", "late-no-room", "late-last-turn"}
    chosen, covered = [], set()
    for fseed in range(5, 60):
        for alpha in (3.0, 4.0, 5.0):
            ids, labels, feats, w, row = eval_case(O, spec, toks, w0, fseed, alpha)
            orc = O.LlamaOracle(spec, w, torch.bfloat16)
            fe = O.connector(orc.W, feats.to(torch.bfloat16)).view(-1, spec.hidden_size)
            for thr in (0.0, 0.4, 0.7):
                d = {}
                O.stream_evaluate(orc, ids, labels, fe, threshold=thr, detail=d, **kw)
                if min(d["margins"]) < EVAL_MIN_MARGIN:
                    continue
                tags = {("hit+" if t[2] > 0 else "hit0") if t[3] == "hit" else t[3] for t in d["turns"]}
                if any(t[1] for t in d["turns"]) and (tags - covered):
                    chosen.append((fseed, alpha, thr))
                    covered |= tags
            if covered >= want:
                break
        if covered >= want:
            break
    print("eval cases", chosen, "cover", sorted(covered))
    assert covered >= want, want - covered
    rec = {"n_cases": np.array(len(chosen))}
    for c, (fseed, alpha, thr) in enumerate(chosen):
        ids, labels, feats, w, row = eval_case(O, spec, toks, w0, fseed, alpha)
        rec[f"c{c}_ids"], rec[f"c{c}_labels"], rec[f"c{c}_feat_seed"] = ids.numpy(), labels.numpy(), np.array(fseed)
        rec[f"c{c}_interval_row"], rec[f"c{c}_threshold"] = row.float().numpy(), np.array(thr, dtype=np.float32)
        for dt_name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            model = build_ref_llm(LiveLlamaConfig, LiveLlamaForCausalLM, spec, w, toks.interval_id, toks.eos_token_id, dt)
            ref = model.stream_evaluate(ids[None], labels[None], feats.to(dt), frame_token_interval_threshold=thr)
            orc = O.LlamaOracle(spec, w, dt)
            fe = O.connector(orc.W, feats.to(dt)).view(-1, spec.hidden_size)
            d = {}
            mine = O.stream_evaluate(orc, ids, labels, fe, threshold=thr, detail=d, **kw)
            print(f"eval c{c} {dt_name} thr={thr} ref", ref.tolist(), "oracle", mine.tolist(), d["turns"], "min margin", min(d["margins"]))
            rec[f"c{c}_{dt_name}"] = ref.float().numpy()
            if dt_name == "bf16":
                rec[f"c{c}_turns"] = np.array([[t[0], -1 if t[1] is None else t[1], 99 if t[2] is None else t[2]] for t in d["turns"]])
    np.savez_compressed(os.path.join(out_dir, "eval_toy128.npz"), **rec)


if __name__ == "__main__":
    main()
