"""GPU parity of the SigLIP encode + token selection + connector (vlo_visual_embed) vs the oracle.

The engine follows the reference's GPU numerics (fp16 matmuls under torch.cuda.amp.autocast,
models/vision_live.py:13); the named CPU reference path is fp32.  Tolerance: the engine's error
against the fp32 oracle must stay within 2x the error of the oracle's own fp16-autocast emulation
(+ 2 bf16 ulps of the output scale, the output being bf16)."""
import os

import numpy as np
import pytest
import torch

from oracle import vlo_oracle as O
from tests.parity_util import within_band

pytestmark = pytest.mark.gpu


def _engine(spec, vspec, w, vw):
    from videollm_online_amd.engine import Engine, EngineConfig
    cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                       num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                       num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size, rope_theta=spec.rope_theta,
                       rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=1024,
                       frame_num_tokens=vspec.frame_num_tokens, frame_token_pooled=vspec.pooled,
                       vit=dict(hidden_size=vspec.hidden_size, intermediate_size=vspec.intermediate_size,
                                num_layers=vspec.num_layers, num_heads=vspec.num_heads, image_size=vspec.image_size,
                                patch_size=vspec.patch_size, ln_eps=vspec.ln_eps))
    e = Engine(cfg)
    e.load_weights(w)
    e.load_weights(vw)
    return e.finalize()


def _oracle_embeds(spec, vspec, w, vw, frames):
    """(gold fp32, CPU reference path = fp32 tower + bf16 connector, fp16-autocast emulation = the reference's GPU numerics); the
    fp32 tower runs once and feeds both connectors."""
    gold_llm, ref_llm = O.LlamaOracle(spec, w, torch.float32), O.LlamaOracle(spec, w, torch.bfloat16)
    tok32 = O.siglip_vision_encode(vw, vspec, frames)
    tok16 = O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16)
    H = vspec.hidden_size
    return (O.connector(gold_llm.W, tok32.reshape(-1, H).float()), O.connector(ref_llm.W, tok32.reshape(-1, H).to(torch.bfloat16)),
            O.connector(ref_llm.W, tok16.reshape(-1, H).to(torch.bfloat16)))


@pytest.mark.parametrize("llm,vit,B,how", [
    ("toy128", "toy", 3, "eager"), ("toy128", "toy", 1, "eager"), ("tinyllama-2l", "siglip-l16-384-2l", 2, "eager"),
    ("tinyllama-2l", "siglip-l16-384-2l", 4, "eager"),      # B=4: two branches of two frames
    ("tinyllama-2l", "siglip-l16-384-2l", 1, "eager"),      # one frame: out-proj / fc2 as split-K slabs + reducing LayerNorm
    # the kernels the bench's batches run on, against the ORACLE (round-3 verdict weak #2: they were only HIP-vs-HIP on hardware):
    ("tinyllama-2l", "siglip-l16-384-2l", 8, "eager"),      # one branch: ping-pong GEMM (4608 rows) + whole-head attention
    ("tinyllama-2l", "siglip-l16-384-2l", 17, "side2"),     # captured graph on a side stream, capture then replay: two branches (9 + 8 frames)
    ("tinyllama-2l", "siglip-l16-384-2l", 56, "side2"),     # the bench's prefetch batch: two branches of 28 frames, 256-row ping-pong tiles
    # BASELINE.json configs[4]'s tower: head dim 72, MLP 4304, 729 patches of 14 pixels — padded
    # heads / MLP width / patch K (csrc/vit.hip::vit_finalize); 1 frame: 64x64 tiles; 9 frames:
    # two branches on 128x128 tiles
    ("tinyllama-2l", "siglip-so400m14-384-2l", 1, "eager"), ("tinyllama-2l", "siglip-so400m14-384-2l", 9, "eager"),
    ("tinyllama-2l", "siglip-so400m14-384-2l", 17, "side2")])
def test_visual_embed_parity(llm, vit, B, how):
    import dataclasses
    spec, vspec = O.LLM_SPECS[llm], O.VIT_SPECS[vit]
    spec = dataclasses.replace(spec, vision_hidden_size=vspec.hidden_size)
    w = O.init_llm_weights(spec, seed=3)
    vw = O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(B, vspec.image_size, seed=1234)
    gold, ref, amp = _oracle_embeds(spec, vspec, w, vw, frames)
    eng = _engine(spec, vspec, w, vw)
    outs = []
    if how == "eager":
        outs.append(eng.visual_embed(frames.cuda()).cpu())               # default stream: eager launches, one branch
    else:
        side = torch.cuda.Stream()
        fr = frames.cuda()
        torch.cuda.synchronize()
        for _ in range(2):                                               # capture, then replay of the captured graph
            with torch.cuda.stream(side):
                o = eng.visual_embed(fr, stream=side)
            side.synchronize()
            outs.append(o.cpu())
        assert torch.equal(outs[0], outs[1]), "graph replay differs from the capturing run"
    torch.cuda.synchronize()
    out = outs[0]
    assert out.shape == (B * vspec.frame_num_tokens, spec.hidden_size)
    scale = gold.abs().max().item()
    e = (out.float() - gold).abs().max().item()
    a = (amp.float() - gold).abs().max().item()
    r = (ref.float() - gold).abs().max().item()
    print(f"[{llm}/{vit} B={B} {how}] engine err {e:.4g}  fp16-autocast-emulation err {a:.4g}  cpu-ref(bf16 connector) err {r:.4g}  scale {scale:.3g}")
    # measured on MI355X (profiles/r4_parity_measurements.txt): e / max(a, r) = 0.86 .. 1.19 over all eleven cases — both yardsticks end in
    # the same bf16 connector, so there is no additive ulp term (round 3's gate was 2 x + 2 bf16 ulps of the scale)
    assert within_band(e, max(a, r), tag=f"test_gpu_vit.py:visual_embed[{vit} B={B} {how}]"), (e, a, r)          # round 6: the shared 1.25 band (was 1.5 x)
    # mean error should be at the bf16-output rounding level
    assert (out.float() - gold).abs().mean().item() <= 2.0 * max((amp.float() - gold).abs().mean().item(), 1e-3 * scale)
    eng.close()


def test_graph_replay_matches_eager():
    """The captured hipGraph encode (side stream) must reproduce the eager launch sequence bit for bit,
    also when batch sizes alternate and the workspace is re-allocated."""
    spec, vspec = O.LLM_SPECS["toy128"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
    eng = _engine(spec, vspec, w, vw)
    frames = O.synthetic_frames(4, vspec.image_size, seed=7).cuda()
    eager = {B: eng.visual_embed(frames[:B]).clone() for B in (1, 3)}       # default stream -> eager path
    torch.cuda.synchronize()                 # one encode workspace per engine: calls on different streams are ordered by the caller
    side = torch.cuda.Stream()
    for B in (1, 3, 1, 4, 3):
        with torch.cuda.stream(side):
            out = eng.visual_embed(frames[:B], stream=side)
        side.synchronize()
        if B in eager:
            assert torch.equal(out, eager[B]), B
    eng.close()


def test_visual_embed_matches_reference_fixture(golden_dir):
    """frame_embeds in the fixture were produced by the reference's LiveMixin.visual_embed."""
    g = np.load(os.path.join(golden_dir, "llm_toy128_bf16.npz"))
    gf = np.load(os.path.join(golden_dir, "llm_toy128_fp32.npz"))
    spec, vspec = O.LLM_SPECS["toy128"], O.VIT_SPECS["toy"]
    w = O.init_llm_weights(spec, seed=3)
    vw = O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(3, vspec.image_size, seed=1234)
    eng = _engine(spec, vspec, w, vw)
    out = eng.visual_embed(frames.cuda()).cpu().float().numpy()
    gold = gf["frame_embeds"]
    e = np.abs(out - gold).max()
    r = np.abs(g["frame_embeds"] - gold).max()
    scale = np.abs(gold).max()
    # r = the reference's own bf16 result against its fp32 one; slack = one bf16 rounding step of the largest embedding (the output format)
    assert within_band(e, r, slack=2 ** -8 * scale, tag="test_gpu_vit.py:reference_fixture"), (e, r, scale)
    eng.close()


def test_vision_tokens_match_reference_function_fixture(golden_dir):
    """tokens in vit_toy.npz were produced by the reference's own `_siglip_vision_encode` (fp32, CPU)."""
    g = np.load(os.path.join(golden_dir, "vit_toy.npz"))
    spec, vspec = O.LLM_SPECS["toy128"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(3, vspec.image_size, seed=1234)
    eng = _engine(spec, vspec, w, vw)
    tok = eng.vision_tokens(frames.cuda()).cpu().float()
    gold = torch.from_numpy(g["tokens"])
    amp = O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16)        # the reference's GPU (autocast) numerics
    e = (tok - gold).abs().max().item()
    a = (amp - gold).abs().max().item()
    scale = gold.abs().max().item()
    assert tok.shape == gold.shape
    assert within_band(e, a, slack=2 ** -8 * scale, tag="test_gpu_vit.py:vision_tokens_fixture"), (e, a, scale)       # fp16 path; slack = one bf16 output rounding step
    # batched offline extraction: batch boundaries must not matter
    enc = eng.encode_video(frames.cuda(), batch_size=2).cpu().float()
    assert torch.equal(enc, tok)
    eng.close()


def test_distributed_encode_writes_reference_layout(tmp_path):
    """data/utils.py:86-104 mirror: two ranks split the directory, every video becomes a bf16 [T,10,Hv] .pt in the
    reference's output directory, and the features equal a direct vision_tokens call."""
    from videollm_online_amd import preprocess as P
    spec, vspec = O.LLM_SPECS["toy128"], O.VIT_SPECS["toy"]
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
    eng = _engine(spec, vspec, w, vw)
    src = tmp_path / "videos_2fps_384"
    src.mkdir()
    vids = {}
    for i, T in enumerate((5, 1, 7)):
        fr = O.synthetic_frames(T, vspec.image_size, seed=50 + i)
        vids[f"clip{i}"] = fr
        if i % 2:
            np.save(src / f"clip{i}.npy", fr.numpy())
        else:
            torch.save(fr, src / f"clip{i}.pt")
    written = []
    for rank in range(2):
        written += P.distributed_encode(eng, src_root=str(src), vision_pretrained="google/siglip-large-patch16-384",
                                        embed_mark="2fps_384_1+3x3", batch_size=3, save_bf16=True, rank=rank, world_size=2)
    dst = str(src) + "_1+3x3_google--siglip-large-patch16-384"
    assert sorted(os.listdir(dst)) == ["clip0.pt", "clip1.pt", "clip2.pt"] and len(written) == 3
    for name, fr in vids.items():
        feats = torch.load(os.path.join(dst, name + ".pt"), weights_only=True)
        assert feats.dtype == torch.bfloat16 and feats.shape == (fr.shape[0], vspec.frame_num_tokens, vspec.hidden_size)
        assert torch.equal(feats, eng.vision_tokens(fr.cuda()).cpu())
    eng.close()


def test_full_depth_siglip_l_vs_cpu_fp32_reference():
    """All 24 encoder layers + MAP head of SigLIP-L/16-384 at its true shapes against the reference's CPU numerics
    (fp32: autocast is a no-op on CPU, models/vision_live.py:13).  The engine computes in the reference's GPU numerics
    (fp16 matmul operands, fp32 accumulation and residual stream) and writes the tokens as bf16 — exactly what the
    reference does next (`frames.to(self.dtype)`, models/modeling_live.py:25).  So the like-for-like yardstick is the oracle's
    fp16-autocast emulation ROUNDED TO bf16: round 3 compared with the un-rounded emulation and read a 2.6x gap (0.0097 vs
    0.0037) that is nothing but that rounding — half a bf16 ulp at |x| in [2, 4) is 0.0078; bf16(emulation) sits at 0.00972 /
    rel. L2 1.82e-3 from the fp32 path, the engine at 0.0097 / 1.8e-3.  Gate: err <= 1.5 x, rel. L2 <= 1.25 x that yardstick
    (was 2 x the un-rounded emulation + 2 bf16 ulps of the scale = 3.7 x)."""
    vspec = O.VIT_SPECS["siglip-l16-384"]
    spec = O.LLM_SPECS["tinyllama-2l"]            # any LLM whose connector takes the tower's 1024-wide tokens
    w, vw = O.init_llm_weights(spec, seed=5), O.init_vit_weights(vspec, seed=2)
    frames = O.synthetic_frames(2, vspec.image_size, seed=99)
    gold = O.siglip_vision_encode(vw, vspec, frames)                                  # the CPU reference path, fp32
    amp = O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16)          # the GPU reference path, emulated
    amp_bf = amp.to(torch.bfloat16).float()                                           # ... and rounded as modeling_live.py:25 rounds it
    eng = _engine(spec, vspec, w, vw)
    tok = eng.vision_tokens(frames.cuda()).cpu()
    torch.cuda.synchronize()
    assert tok.dtype == torch.bfloat16 and tok.shape == gold.shape
    tok = tok.float()
    scale = gold.abs().max().item()
    e, a, a_bf = (tok - gold).abs().max().item(), (amp - gold).abs().max().item(), (amp_bf - gold).abs().max().item()
    rel = ((tok - gold).norm() / gold.norm()).item()
    rel_a = ((amp_bf - gold).norm() / gold.norm()).item()
    same = (tok == amp_bf).float().mean().item()
    print(f"[siglip-l16-384 x24] engine vs fp32 CPU path: max err {e:.4g} (scale {scale:.3g}), rel. L2 {rel:.3e}; "
          f"bf16(fp16-autocast emulation) vs fp32: max err {a_bf:.4g}, rel. L2 {rel_a:.3e} (un-rounded emulation {a:.4g}); "
          f"engine tokens bit-equal to bf16(emulation): {same:.3f}")
    assert within_band(e, a_bf, tag="test_gpu_vit.py:full_depth_siglip_l"), (e, a_bf, scale)
    assert rel <= 1.25 * rel_a
    eng.close()


def test_two_branch_batched_encode_matches_single_branch():
    """From 4 frames up (VLO_VIT_SPLIT_MIN) the captured encode runs as two parallel half-batch branches on two streams, each on its own slice of the
    workspace and its own connector scratch (csrc/vit.hip::vit_visual_embed).  Same frames through the eager single-branch path
    (default stream) must give the same embeddings — the rows are independent, only which GEMM tile variant computes them differs."""
    spec, vspec = O.LLM_SPECS["tinyllama-2l"], O.VIT_SPECS["siglip-l16-384-2l"]
    w, vw = O.init_llm_weights(spec, seed=5), O.init_vit_weights(vspec, seed=1)
    eng = _engine(spec, vspec, w, vw)
    frames = O.synthetic_frames(13, vspec.image_size, seed=21).cuda()
    side = torch.cuda.Stream()
    for B in (13, 12, 9, 5, 4):
        eager = eng.visual_embed(frames[:B]).clone()                 # default stream: no graph, one branch
        torch.cuda.synchronize()             # the engine's encode workspace is shared: calls on different streams are ordered by the caller
        for _ in range(2):                                            # capture, then replay
            with torch.cuda.stream(side):
                out = eng.visual_embed(frames[:B], stream=side)
            side.synchronize()
            d = (out.float() - eager.float()).abs().max().item()
            scale = eager.float().abs().max().item()
            print(f"[two-branch encode B={B}] max |diff| {d:.3g} (scale {scale:.3g}), bit-equal {bool(torch.equal(out, eager))}")
            assert d <= 2 * 2 ** -8 * scale
    eng.close()


def test_large_batch_tiles_match_small_batch_tiles():
    """From 16 frames in one launch the wide GEMMs run on 256x256 tiles / 16 waves (csrc/vit.hip::gemm_launch); smaller batches on
    128x128 tiles / 8 waves.  Every output element accumulates its K range in the same order on both, so 34 frames encoded at once
    (two branches of 17: 9792 rows, a partial last tile) must agree with the same frames encoded 8 at a time."""
    spec, vspec = O.LLM_SPECS["tinyllama-2l"], O.VIT_SPECS["siglip-l16-384-2l"]
    w, vw = O.init_llm_weights(spec, seed=5), O.init_vit_weights(vspec, seed=1)
    eng = _engine(spec, vspec, w, vw)
    frames = O.synthetic_frames(34, vspec.image_size, seed=22).cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        big = eng.visual_embed(frames, stream=st).float()
    st.synchronize()                     # the engine's encode workspace is shared: order the two streams
    small = torch.cat([eng.visual_embed(frames[i:i + 8]).float() for i in range(0, 34, 8)])
    torch.cuda.synchronize()
    d = (big - small).abs().max().item()
    scale = small.abs().max().item()
    print(f"[256-tile vs 128-tile encode] max |diff| {d:.3g} (scale {scale:.3g}), bit-equal {bool(torch.equal(big, small))}")
    assert d <= 2 ** -7 * scale            # two bf16 ulps of the largest output; bit-equal expected
    eng.close()

