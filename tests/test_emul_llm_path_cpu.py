"""The engine's LLM-path SOURCES (csrc/{gemv,prefill,llm_ops,engine,tp}.hip) compiled for the CPU against the
HIP-on-threads shim of tests/hip_emul/ and driven through the same C ABI as the product, at toy sizes, against the oracle.

What this pins WITHOUT a GPU: launch sequencing and argument wiring of the host code, index arithmetic of every kernel
(packed weight fragments, MFMA register maps, paged KV, split-KV attention and its combine, epilogues), rounding points.
What it cannot: the GPU memory model, the matrix core's internal summation order, performance.  It is how code paths
written without GPU time (the tp_reduce_norm refactor, the peer-to-peer TP exchange between logical ranks) were checked
before their first run on hardware; the `-m gpu` suite remains the parity gate.

The emulation is slow (every GPU thread is an OS thread): the default set below takes ~2-3 minutes including the one-off
build of the emulated library; VLO_EMUL_FULL=1 adds the longer cases."""
import os

import pytest
import torch

from tests.parity_util import within_band

from oracle import vlo_oracle as O

FULL = os.environ.get("VLO_EMUL_FULL") == "1"
TINY = O.LlmSpec(128, 192, 2, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=128)     # K = 128 / 192: 256- and 64-thread GEMV blocks


@pytest.fixture(scope="module")
def E():
    import resource
    soft, _ = resource.getrlimit(resource.RLIMIT_NPROC)
    if soft != resource.RLIM_INFINITY and soft < 4096:
        pytest.skip(f"the emulation runs every GPU thread of a block as an OS thread (up to 1024): RLIMIT_NPROC = {soft}")
    from tests.hip_emul import emul_engine
    if emul_engine.lib() is None:
        pytest.skip("no clang++ to build the emulated library")
    return emul_engine


def _three_way(name, i, out, rl, gl):
    e = (out.float() - gl).abs().max().item()
    r = (rl.float() - gl).abs().max().item()
    scale = gl.abs().max().item()
    print(f"[emul {name}] step {i}: engine err {e:.4g} ref-bf16 err {r:.4g} scale {scale:.3g}")
    assert within_band(e, r, 1e-3 * scale, "test_emul_llm_path_cpu.py:40"), f"{name} step {i}: {e} vs {r}"


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


TINY_GQA = O.LlmSpec(128, 192, 2, 2, 1, 256, 10000.0, 1e-5, vision_hidden_size=128)  # 2 query heads share one kv head


def test_default_pipeline_matches_oracle(E):
    """run_chunk as shipped (7 launches per layer) with grouped-query attention (the 2-heads-per-wave attention kernel);
    VLO_EMUL_FULL: on the 'toy' model (4 heads over 2 kv heads, 512-thread GEMV blocks)."""
    spec = O.LLM_SPECS["toy"] if FULL else TINY_GQA
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
        _three_way("default pipeline", i, allr, rl, gl)
    tok, p = eng.stream_sample(s, 0.725, toks.interval_id)
    rt, rp = O.stream_sample(rl[-1].clone(), toks.interval_id, 0.725)
    top2 = rl[-1].float().topk(2).values
    assert tok == rt or (top2[0] - top2[1]).item() < 0.12
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
    # the exchange micro-benchmark bench.py reports for tensor-parallel runs: runs, returns a time, leaves the group usable
    xs = grp.new_session()
    assert grp.bench_exchange(xs, 3, 2) >= 0.0
    if p2p:
        assert grp.p2p_status()["timed_out"] == 0
    grp.close()


def _rank_worker(rank, T, conns, q, lens):
    """one emulated 'GPU process': its own copy of the emulated library (statics, thread pool), mailboxes in POSIX shm"""
    os.environ["VLO_TP_P2P_TIMEOUT_MS"] = "900000"       # emulated ranks take seconds per step: waiting is not a failure here
    from tests.hip_emul import emul_engine as E
    spec = TINY
    w = O.init_llm_weights(spec, seed=6)
    toks = O.default_tokens(spec)
    ref = O.LlamaOracle(spec, w, torch.bfloat16)

    def exchange(mine):                                  # all-gather of the handles over pipes, rank order
        for c in conns:
            c.send((rank, mine))
        got = dict([(rank, mine)] + [c.recv() for c in conns])
        return [got[r] for r in range(T)]

    r = E.EmulTpRank(spec, T, rank, w, O.rope_inv_freq(spec.head_dim, spec.rope_theta), exchange)
    outs = [r.llm_step(x)[1].float().numpy() for x in _steps(spec, ref, toks, 3, lens)]
    q.put((rank, outs, r.p2p_status()))
    r.close()


def test_p2p_between_processes(E):
    """One process per rank, as on a multi-GPU node: mailbox export / open (hipIpc emulated over POSIX shared memory), the
    FUSED publish + collect kernel running concurrently in both ranks, a group without any RCCL communicator."""
    import torch.multiprocessing as mp
    T = 2
    lens = [11, 1] + ([19] if FULL else [])
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_worker, args=(0, T, [a], q, lens)), ctx.Process(target=_rank_worker, args=(1, T, [b], q, lens))]
    [p.start() for p in ps]
    res = {}
    for _ in range(T):
        rank, outs, st = q.get(timeout=900)
        res[rank] = (outs, st)
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    spec = TINY
    w = O.init_llm_weights(spec, seed=6)
    toks = O.default_tokens(spec)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 3, lens)):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        o0, o1 = torch.from_numpy(res[0][0][i]), torch.from_numpy(res[1][0][i])
        assert torch.equal(o0, o1), "both ranks must hold the same logits (same sum order on every rank)"
        _three_way("tiny tp2 p2p, 2 processes", i, o0, rl, gl)
    assert all(res[r][1]["enabled"] == 1 and res[r][1]["timed_out"] == 0 for r in range(T))


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1")
def test_block_path_generation_and_evaluation_surface(E):
    """The 64-token block path (run_block / gemm64_kernel) for a 40-token first step, greedy generation, the streaming
    sampler, KV fork / crop, joint_embed and the per-row logit statistics of stream_evaluate."""
    spec = TINY
    w = O.init_llm_weights(spec, seed=9)
    toks = O.default_tokens(spec, n_start=30)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    g = torch.Generator().manual_seed(4)
    frame = torch.randn(10, spec.hidden_size, generator=g).bfloat16()
    # joint_embed == embedding rows with the placeholder rows replaced, in order
    v_id = spec.vocab_size
    ids = torch.tensor(toks.start_ids + [v_id] * 10)
    x = eng.joint_embed(ids, frame, v_id)
    assert torch.equal(x, torch.cat([ref.embed(torch.tensor(toks.start_ids)), frame]))
    assert torch.equal(eng.embed(torch.tensor(toks.start_ids)), ref.embed(torch.tensor(toks.start_ids)))
    rl, rc = ref.forward(x, None)
    gl, gc = gold.forward(x, None)
    last, allr = eng.llm_step(s, x)                                     # 40 rows: one 64-token block
    assert eng.session_len(s) == 40 and torch.equal(last, allr[-1])
    _three_way("tiny block path", 0, allr, rl, gl)
    # per-row statistics of the teacher-forced logits vs torch on the SAME logits
    labels = torch.randint(0, spec.vocab_size, (40,), generator=g)
    st = eng.logit_rows(allr, labels, toks.interval_id)
    lf = allr.float()
    assert torch.allclose(st["lse"], torch.logsumexp(lf, -1), atol=2e-3)
    assert torch.equal(st["argmax"], lf.argmax(-1))
    assert torch.equal(st["label_logit"], lf.gather(1, labels[:, None])[:, 0])
    sm = allr.softmax(-1)                                               # bf16 softmax, as the reference computes it
    assert torch.allclose(st["p_interval"], sm[:, toks.interval_id].float(), atol=4e-3)
    # KV fork keeps a prefix, crop forgets a suffix; both continue exactly like the oracle's cache of that length
    f = eng.fork(s, 35)
    assert eng.session_len(f) == 35 and eng.session_len(s) == 40
    x2 = ref.embed(torch.tensor([17, 23, 5]))

    def prefix(cache, n):                                               # what trim_past_key_values(past, 0, n) keeps
        c = O.KVCacheOracle(spec.num_layers)
        c.k, c.v = [k[:, :n] for k in cache.k], [v[:, :n] for v in cache.v]
        return c

    rl_f, _ = ref.forward(x2, prefix(rc, 35))
    gl_f, _ = gold.forward(x2, prefix(gc, 35))
    lf2, af2 = eng.llm_step(f, x2)
    eng.crop(s, 35)
    ls2, as2 = eng.llm_step(s, x2)
    assert torch.equal(af2, as2), "a forked prefix and a cropped session of the same length must continue identically"
    _three_way("tiny fork(35)", 1, af2, rl_f, gl_f)
    # greedy generation through the C loop (forced length: random weights never emit EOS on their own schedule)
    out = eng.greedy_generate(s, ref.embed(torch.tensor(toks.stream_generation_ids)), toks.eos_token_id, 6, force_len=4)
    assert len(out) == 4 and out[-1] == toks.eos_token_id and toks.eos_token_id not in out[:-1]
    assert eng.session_len(s) == 38 + len(toks.stream_generation_ids) + 3
    eng.close()


@pytest.mark.parametrize("K,N,n,plan", [(1792, 48, 11, (4, 14, 1, 1)), (704, 32, 3, (2, 11, 1, 1)), (512, 64, 16, (8, 2, 1, 1))])
def test_gemv_plans_in_emulation(E, K, N, n, plan):
    """The weight-streaming GEMV on reduction lengths whose (waves, fragments) combination is rare: K = 1792 is the
    Llama-3-8B down-proj shard at TP = 8 (4 waves x 14 fragments, one K slice)."""
    assert E.gemv_plan(K, True) == plan
    g = torch.Generator().manual_seed(K)
    x = torch.randn(n, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    y = E.test_gemv(x, W)
    want = x.float() @ W.float().T
    assert torch.allclose(y, want, rtol=1e-4, atol=1e-4), (y - want).abs().max()


@pytest.mark.parametrize("B", [1, 3] if FULL else [1])
def test_vision_tower_and_connector(E, B):
    """csrc/vit.hip on the toy tower (96x96 frames, 36 patches): fused uint8 preprocessing + patch embed, encoder layers
    (fp16 MFMA GEMMs through XOR-swizzled LDS, attention, LayerNorms), MAP head, CLS + 3x3 pooling, connector — eager and
    as the captured-and-replayed graph — with the tolerance of tests/test_gpu_vit.py."""
    spec, vspec = O.LLM_SPECS["toy"], O.VIT_SPECS["toy"]
    w = O.init_llm_weights(spec, seed=3)
    vw = O.init_vit_weights(vspec, seed=1)
    frames = O.synthetic_frames(B, vspec.image_size, seed=1234)
    gold_llm, ref_llm = O.LlamaOracle(spec, w, torch.float32), O.LlamaOracle(spec, w, torch.bfloat16)
    gold = gold_llm.visual_embed(vw, vspec, frames)
    ref = ref_llm.visual_embed(vw, vspec, frames)
    amp = ref_llm.visual_embed(vw, vspec, frames, mm_dtype=torch.float16)        # the reference's GPU numerics, emulated
    eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    out = eng.visual_embed(frames)
    assert out.shape == (B * vspec.frame_num_tokens, spec.hidden_size)
    scale = gold.abs().max().item()
    e = (out.float() - gold).abs().max().item()
    a = (amp.float() - gold).abs().max().item()
    r = (ref.float() - gold).abs().max().item()
    print(f"[emul vit toy B={B}] engine err {e:.4g}  fp16-autocast-emulation err {a:.4g}  cpu-ref err {r:.4g}  scale {scale:.3g}")
    assert e <= 2.0 * max(a, r) + 2 * 2 ** -8 * scale
    if FULL:
        # the captured graph replays the same launches: bit-identical, also on the second replay with other frames in between
        import ctypes
        fake_stream = ctypes.c_void_p(0x10)
        g1 = eng.visual_embed(frames, stream=fake_stream)
        assert torch.equal(g1, out)
        other = O.synthetic_frames(B, vspec.image_size, seed=99)
        g2 = eng.visual_embed(other, stream=fake_stream)
        assert not torch.equal(g2, out)
        assert torch.equal(eng.visual_embed(frames, stream=fake_stream), out)
        # pre-connector tokens (offline feature extraction, vlo_vision_tokens) against the oracle's encode
        tok = eng.vision_tokens(frames)
        assert tok.shape == (B, vspec.frame_num_tokens, vspec.hidden_size)
        want = O.siglip_vision_encode(vw, vspec, frames)
        assert (tok.float() - want.float()).abs().max().item() <= 2.0 * (O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16).float()
                                                                        - want.float()).abs().max().item() + 2 * 2 ** -8 * want.abs().max().item()
    eng.close()


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1")
def test_tensor_parallel_four_ranks_p2p_after_block_path(E):
    """T = 4 logical ranks (one kv head each) through the peer-to-peer exchange, next to a plain engine, with a block-path
    first step followed by decode and frame steps."""
    spec = O.LlmSpec(256, 384, 2, 4, 4, 512, 10000.0, 1e-5, vision_hidden_size=128)
    w = O.init_llm_weights(spec, seed=12)
    toks = O.default_tokens(spec, n_start=20)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    inv = O.rope_inv_freq(spec.head_dim, spec.rope_theta)
    grp = E.EmulTpGroup(spec, 4, w, inv, p2p=True)
    ts = grp.new_session()
    eng = E.EmulEngine(spec).load_weights(w, inv)
    fs = eng.new_session()
    g = torch.Generator().manual_seed(5)
    first = torch.cat([ref.embed(torch.tensor(toks.start_ids)), torch.randn(10, spec.hidden_size, generator=g).bfloat16()])   # 30 rows
    steps = [first, ref.embed(torch.tensor([17])), ref.embed(torch.tensor([29])),
             torch.cat([ref.embed(torch.tensor([toks.interval_id])), torch.randn(10, spec.hidden_size, generator=g).bfloat16()])]
    rc = gc = None
    for i, x in enumerate(steps):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        _, at = grp.llm_step(ts, x)
        _, af = eng.llm_step(fs, x)
        assert grp.session_len(ts) == eng.session_len(fs) == len(rc)
        _three_way("tp4 p2p", i, at, rl, gl)
        _three_way("tp1 after block path", i, af, rl, gl)
    assert grp.p2p_status() == dict(enabled=1, timed_out=0, uncached_mailbox=grp.p2p_status()["uncached_mailbox"])
    grp.close()
    eng.close()


GQA4 = O.LlmSpec(256, 192, 1, 4, 1, 256, 10000.0, 1e-5, vision_hidden_size=128)      # 4 query heads on one kv head, head_dim 64


def test_column_packed_attention_kernel(E):
    """attn_cols_kernel (llm_ops.hip): the (token, head) columns of a short step packed into 1, 2 and 3 MFMA column tiles (4 heads per kv
    head: n = 1 / 4, 5 / 8, 9 / 12), the permuted key order inside a 32-key block, eight key sub-splits merged pairwise, two KV splits and a
    page boundary (cache > 256 tokens), next to attn_chunk_kernel for n = 13 / 16 — all through vlo_llm_step against the oracle."""
    spec = GQA4
    w = O.init_llm_weights(spec, seed=31)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    g = torch.Generator().manual_seed(5)
    rc = gc = None
    for i, n in enumerate([12, 1, 9, 200, 11, 13, 5, 1, 4, 8, 16, 3]):      # cache: 12, 13, 22, 222, 233, 246, 251, 252, 256, 264, 280, 283
        x = (torch.randn(n, spec.hidden_size, generator=g) * 0.7).bfloat16()
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = eng.llm_step(s, x)
        assert eng.session_len(s) == len(rc)
        _three_way(f"column-packed attention n={n} L={len(rc)}", i, allr, rl, gl)
    eng.close()


VIT256_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle import vlo_oracle as O
from tests.hip_emul import emul_engine as E
spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=256)
vspec = O.VitSpec(hidden_size=256, intermediate_size=512, num_layers=1, num_heads=4, image_size=96, patch_size=16, pooled=(3, 3))
w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
frames = O.synthetic_frames(3, vspec.image_size, seed=7)
gold = O.LlamaOracle(spec, w, torch.float32).visual_embed(vw, vspec, frames)
ref = O.LlamaOracle(spec, w, torch.bfloat16)
amp = ref.visual_embed(vw, vspec, frames, mm_dtype=torch.float16)
cpu = ref.visual_embed(vw, vspec, frames)
eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
out = eng.visual_embed(frames)
scale = gold.abs().max().item()
e = (out.float() - gold).abs().max().item()
a = (amp.float() - gold).abs().max().item()
r = (cpu.float() - gold).abs().max().item()
print(f"[emul vit 256-tile] engine err {e:.4g} fp16-autocast err {a:.4g} cpu-ref err {r:.4g} scale {scale:.3g}")
assert e <= 2.0 * max(a, r) + 2 * 2 ** -8 * scale, (e, a, r)
eng.close()
print("OK256")
"""


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (62 s; the ping-pong kernel that ships runs in the default suite below, "
                    "this 16-wave tile on hardware in tests/test_gpu_vit.py)")
def test_vit_gemm_256_tile_kernel_in_emulation(E):
    """vit_gemm_kernel<256,256,4,4,...> (16 waves, one 1024-thread block per tile; csrc/vit.hip::gemm_launch takes it from 9216 rows)
    forced onto a small tower (hidden 256: N = 256 / 512 / 768, M = 108 rows = one partial tile) in a child process — the dispatch
    thresholds are read once per process — against the oracle with the tolerance of the ViT tests."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VLO_VIT_BIG_TILES="1", VLO_VIT_256_MIN_ROWS="1", VLO_VIT_SPLIT_MIN="0")
    r = subprocess.run([sys.executable, "-c", VIT256_CHILD % root], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-600:])
    assert r.returncode == 0 and "OK256" in r.stdout, r.stderr[-2000:]


VITPP_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle import vlo_oracle as O
from tests.hip_emul import emul_engine as E
spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=256)
vspec = O.VitSpec(hidden_size=256, intermediate_size=512, num_layers=1, num_heads=4, image_size=128, patch_size=16, pooled=(3, 3))
w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=1)
frames = O.synthetic_frames(5, vspec.image_size, seed=7)          # 64 tokens per frame, 320 token rows: one full 256-row tile + a partial one
gold = O.LlamaOracle(spec, w, torch.float32).visual_embed(vw, vspec, frames)
ref = O.LlamaOracle(spec, w, torch.bfloat16)
amp = ref.visual_embed(vw, vspec, frames, mm_dtype=torch.float16)
cpu = ref.visual_embed(vw, vspec, frames)
eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
out = eng.visual_embed(frames)
scale = gold.abs().max().item()
e = (out.float() - gold).abs().max().item()
a = (amp.float() - gold).abs().max().item()
r = (cpu.float() - gold).abs().max().item()
print(f"[emul vit ping-pong] engine err {e:.4g} fp16-autocast err {a:.4g} cpu-ref err {r:.4g} scale {scale:.3g}")
assert e <= 2.0 * max(a, r) + 2 * 2 ** -8 * scale, (e, a, r)
if len(sys.argv) > 1:
    torch.save(out, sys.argv[1])
eng.close()
print("OKPP")
"""


def _vitpp_child(env_extra, save=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VLO_VIT_SPLIT_MIN="0", **env_extra)     # one branch: the row counts above are chosen for full + partial tiles
    r = subprocess.run([sys.executable, "-c", VITPP_CHILD % root] + ([save] if save else []), env=env, capture_output=True, text=True, timeout=1800)
    print(r.stdout[-400:])
    assert r.returncode == 0 and "OKPP" in r.stdout, r.stderr[-2000:]


def test_vit_gemm_pingpong_kernel_in_emulation(E):
    """vit_gemm_pp_kernel (256 x 256 tiles, two wave groups alternating read / MFMA segments, half-tiles restaged by direct-to-LDS
    loads under COUNTED vmcnt waits; csrc/vit_gemm.inc) forced onto a small tower (hidden 256: N = 256 / 512 / 768, K = 256 / 512,
    320 rows = a full tile + a partial one, 2 column groups in the XCD split where the width allows; persistent: 4 emulated CUs walk
    2 - 6 tiles), with the emulated direct-to-LDS loads landing as LATE as the hardware may land them (VLO_EMUL_GLDS=late: only at the
    vmcnt wait that retires them) — a fragment read that is not covered by its wait + barriers reads stale shared memory and fails
    here — against the oracle.  The same run forces vit_attn_head_kernel (whole head in LDS, 12 waves x 48 queries; 64 tokens: two
    waves with work, a masked tail) instead of the 64-query-tile attention kernel."""
    _vitpp_child(dict(VLO_VIT_PP_MIN_ROWS="1", VLO_VIT_PP_BM="256", VLO_VIT_PP_CB="2", VLO_EMUL_GLDS="late", VLO_VIT_ATTN_HEAD_MIN="1"))


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (2 minutes; the hardware parity cases are tests/test_gpu_vit.py SigLIP-L B = 1 .. 4)")
def test_vit_small_tile_direct_to_lds_kernels_in_emulation(E, tmp_path):
    """Few frames: the 64 x 64 / 32 x 64 tiles as direct-to-LDS kernels with 4 or 6 stages (csrc/vit_gemm.inc::gemm_launch,
    VLO_VIT_SMALL_STAGES) — K = 256 / 512 here is 4 / 8 K tiles, i.e. a prologue SHORTER than the 6-stage pipeline, the steady state and
    the drain, each with its counted vmcnt — with the emulated loads landing only at the wait that retires them: inside the oracle's
    band and bit-identical to the register-ring form of the same kernels (same MFMA, same k order)."""
    base = str(tmp_path / "ring.pt")
    _vitpp_child(dict(VLO_VIT_PP="0", VLO_VIT_ATTN_HEAD_MIN="0", VLO_VIT_SMALL_STAGES="0"), base)
    want = torch.load(base)
    for stages in (("4", "6") if FULL else ("4",)):          # 4 is what ships (3 differs only in the stage count); 6 with VLO_EMUL_FULL=1
        f = str(tmp_path / f"glds_{stages}.pt")
        _vitpp_child(dict(VLO_VIT_PP="0", VLO_VIT_ATTN_HEAD_MIN="0", VLO_VIT_SMALL_STAGES=stages, VLO_EMUL_GLDS="late"), f)
        assert torch.equal(torch.load(f), want), stages
    # the two residual GEMMs (out-proj K = 256, fc2 K = 512) as split-K slices (2 and 4) reduced by the LayerNorm that follows them:
    # another summation order, so inside the oracle's band (checked by the child) and within a few bf16 ulps of the un-split result
    f = str(tmp_path / "splitk.pt")
    _vitpp_child(dict(VLO_VIT_PP="0", VLO_VIT_ATTN_HEAD_MIN="0", VLO_EMUL_GLDS="late", VLO_VIT_SPLITK="4", VLO_VIT_SPLITK_MIN_TILES="2",
                      VLO_VIT_PP_GRID="64"), f)
    got = torch.load(f)
    assert not torch.equal(got, want), "split-K did not run"
    assert (got.float() - want.float()).abs().max().item() <= 4 * 2 ** -8 * want.float().abs().max().item()


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1")
def test_vit_gemm_pingpong_kernel_bit_identical_to_small_tile_kernels(E, tmp_path):
    """Same MFMA, same k order: the ping-pong kernel's outputs equal the 64 x 64 / 128 x 128 kernels' bit for bit, for both tile
    heights and both landing models of the emulated direct-to-LDS loads."""
    base = str(tmp_path / "base.pt")
    _vitpp_child(dict(VLO_VIT_PP="0", VLO_VIT_ATTN_HEAD_MIN="0"), base)
    want = torch.load(base)
    for bm, mode, cb in (("256", "late", "0"), ("128", "late", "2"), ("256", "sync", "2"), ("128", "sync", "0")):
        f = str(tmp_path / f"pp_{bm}_{mode}.pt")
        _vitpp_child(dict(VLO_VIT_PP_MIN_ROWS="1", VLO_VIT_PP_BM=bm, VLO_VIT_PP_CB=cb, VLO_EMUL_GLDS=mode, VLO_VIT_ATTN_HEAD_MIN="0"), f)
        assert torch.equal(torch.load(f), want), (bm, mode, cb)


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1")
@pytest.mark.parametrize("n", [17, 64, 65, 81])
def test_step_chunking_boundaries(E, n):
    """vlo_llm_step's split of a long input: blocks of 64 tokens through the block path, a tail of <= 16 through the 16-row
    pipeline (17 = one block of 17; 64 = one full block; 65 = block + 1-row chunk; 81 = block + block of 17)."""
    spec = TINY_GQA
    w = O.init_llm_weights(spec, seed=20 + n)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, spec.hidden_size, generator=g).bfloat16()
    rl, rc = ref.forward(x, None)
    gl, gc = gold.forward(x, None)
    last, allr = eng.llm_step(s, x)
    assert eng.session_len(s) == n and torch.equal(last, allr[-1])
    _three_way(f"chunking n={n}", 0, allr, rl, gl)
    x2 = torch.randn(3, spec.hidden_size, generator=g).bfloat16()          # and the cache it left behind is usable
    rl2, _ = ref.forward(x2, rc)
    gl2, _ = gold.forward(x2, gc)
    _, a2 = eng.llm_step(s, x2)
    _three_way(f"chunking n={n} + 3", 1, a2, rl2, gl2)
    eng.close()


@pytest.mark.parametrize("H,W,R,layout,a", [(90, 160, 64, 0, -0.6), (160, 90, 64, 1, -0.6), (48, 64, 64, 0, -0.5), (64, 64, 64, 1, -0.6),
                                            (101, 333, 96, 0, -0.75)])
def test_frame_ingest_kernels_match_the_oracle(E, H, W, R, layout, a):
    """csrc/ingest.hip (the kernel SOURCES, emulated) vs oracle/ingest_oracle.py: geometry of the ffmpeg scale + pad filter
    (data/utils.py:64), antialiased bicubic, rounding, padding, both source layouts; odd widths exercise the unaligned row
    staging.  fp32 tap weights vs the oracle's float64: a pixel may differ by one level where the exact value sits on a
    rounding boundary."""
    import ctypes as C

    import numpy as np

    from oracle import ingest_oracle as G
    spec = TINY
    eng = E.EmulEngine(spec).load_weights(O.init_llm_weights(spec, seed=5), O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    rng = np.random.default_rng(H * 7 + W)
    fr = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    fr[1] = (np.add.outer(np.arange(H), np.arange(W))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)     # smooth content
    want = G.ingest(fr, R, a)
    src = torch.from_numpy(fr if layout == 0 else np.ascontiguousarray(fr.transpose(0, 3, 1, 2)))
    got = eng.frame_ingest(src, layout, R, a).numpy()
    ow, oh, x0, y0 = (C.c_int(), C.c_int(), C.c_int(), C.c_int())
    E.check(E.lib().vlo_frame_ingest_geometry(W, H, R, C.byref(ow), C.byref(oh), C.byref(x0), C.byref(y0)))
    assert (ow.value, oh.value, x0.value, y0.value) == G.ffmpeg_scale_pad_geometry(W, H, R)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.995, (d.max(), (d == 0).mean())
    if H == W == R:
        assert np.array_equal(got, fr.transpose(0, 3, 1, 2))          # no resampling: bit-exact pass-through
    eng.close()


@pytest.mark.parametrize("n,N,K", [(11, 64, 512), (3, 48, 1024), (16, 32, 2048), (7, 64, 1792),      # K = 1792: 4 waves x 14 fragments (the 8B down-proj shard at TP = 8)
                                   (5, 8336, 512)])      # 521 tiles: two groups per block (both register sets re-loaded), last group has one tile
def test_fp8_weight_image_gemv(E, n, N, K):
    """The fp8 e4m3 weight image (two MFMA fragments per 16-byte lane load, e4m3 -> bf16 expansion in registers, per-output-channel
    scale in the epilogue; csrc/gemv.hip, gemv_body.inc WQ = 1) against an fp64 matmul of the dequantised weights; quantisation by
    checkpoint.quantize_fp8_per_channel == the oracle's fp8_dequantized_weights."""
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    g = torch.Generator().manual_seed(n + N + K)
    x = torch.randn(n, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * K ** -0.5 * (1 + 3 * torch.rand(N, 1, generator=g))).bfloat16()
    W[1] = 0                                                     # an all-zero row: scale 1, zeros
    W[2, :8] = W[2].abs().max() * 2 ** -9                        # values that quantise to e4m3 subnormals
    q, s = quantize_fp8_per_channel(W)
    y = E.test_gemv_fp8(x, q, s)
    Wd = q.float().double() * s.double()[:, None]
    ref = x.double() @ Wd.T
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()) * (K / 256) ** 0.5 + 1e-5
    W2 = O.fp8_dequantized_weights({"lm_head.weight": W})["lm_head.weight"]
    assert torch.equal(W2.double(), Wd.float().double())


FP8_TINY = O.LlmSpec(512, 1024, 1, 4, 2, 256, 10000.0, 1e-5, vision_hidden_size=128)    # K = 512 / 1024: the smallest shapes the fp8 image takes


def test_fp8_mfma_prefill_gemm_in_emulation(E):
    """The W8A8 prefill GEMM (csrc/prefill.h: quantize_rows_fp8_kernel + vit_gemm_pp_kernel<.., F8 = 1> on v_mfma_f32_16x16x128_f8f6f4, the shim's
    emulation of it): one 128-row tile, four K tiles.  Codes and scales bit for bit against the oracle's rule; the sums against fp64 on the same
    codes (the emulated instruction accumulates in fp32 — the hardware's narrower adder is measured in tests/test_gpu_fp8_mfma.py)."""
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    g = torch.Generator().manual_seed(21)
    M, N, K = 70, 256, 512
    x = torch.randn(M, K, generator=g)
    x[:, ::61] *= 25.0
    x[3] = 0.0
    x = x.bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    q, s = quantize_fp8_per_channel(W)
    y, codes, xs = E.test_gemm_fp8(x, q, s)
    oq, osc = O.fp8_quantize_rows(x)
    assert torch.equal(xs, osc[:, 0])
    assert torch.equal(codes.float(), oq)
    ref = (oq.double() @ q.float().double().T) * s.double()[None, :] * osc.double()
    assert (y.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1")
def test_fp8_mfma_engine_prefill_path_in_emulation(E):
    """An fp8 engine with prefill_act_dtype = 1 on the prefill path: 300 tokens W8A8 (decoder-layer projections; the lm_head keeps bf16
    activations), then a decode step on bf16 activations — 3-way against the oracle run with the same rule per call."""
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    spec = O.LlmSpec(512, 512, 1, 4, 1, 256, 10000.0, 1e-5, vision_hidden_size=128)
    w = O.init_llm_weights(spec, seed=12)
    eng_w, ora_w, keep = {}, {}, set()
    for k, v in w.items():
        if k.endswith(O.FP8_STREAMED):
            q, sc = quantize_fp8_per_channel(v)
            eng_w[k], eng_w[k + "_scale"] = q, sc
            ora_w[k] = q.float() * sc[:, None]
            keep.add(k)
        else:
            eng_w[k] = ora_w[k] = v
    ref, gold = O.LlamaOracle(spec, ora_w, torch.bfloat16, keep_fp32=keep), O.LlamaOracle(spec, ora_w, torch.float32)
    eng = E.EmulEngine(spec, kv_pool_tokens=1024, weight_dtype=1, prefill_act_dtype=1).load_weights(eng_w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    g = torch.Generator().manual_seed(4)
    rc = gc = None
    for i, n in enumerate((300, 1)):
        x = (torch.randn(n, spec.hidden_size, generator=g) * 0.7).bfloat16()
        rl, rc = ref.forward(x, rc, act_fp8=(n >= 256))
        gl, gc = gold.forward(x, gc, act_fp8=(n >= 256))
        last, allr = eng.llm_step(s, x)
        assert eng.session_len(s) == len(rc)
        _three_way("fp8 mfma prefill path", i, allr, rl, gl)
    eng.close()


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (1.5 minutes; the bf16 prefill path runs in the default suite, the hardware case is tests/test_gpu_fp8.py::test_fp8_prefill_path_teacher_forced_rows)")
def test_fp8_engine_prefill_path_in_emulation(E):
    """An fp8 engine on the prefill path: the e4m3 image expanded to the packed bf16 image per GEMM (prefill.hip::expand_fp8_image_kernel)
    and the per-channel scales applied in the GEMM epilogues (plain, SwiGLU, residual): 300 tokens, then a decode step, all rows' logits
    3-way against the reference arithmetic on the dequantised weights."""
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    spec = O.LlmSpec(512, 512, 1, 4, 1, 256, 10000.0, 1e-5, vision_hidden_size=128)      # K = 512: the smallest the fp8 image takes; head dim 128, GQA 4
    w = O.init_llm_weights(spec, seed=12)
    eng_w, ora_w, keep = {}, {}, set()
    for k, v in w.items():
        if k.endswith(O.FP8_STREAMED):
            q, sc = quantize_fp8_per_channel(v)
            eng_w[k], eng_w[k + "_scale"] = q, sc
            ora_w[k] = q.float() * sc[:, None]
            keep.add(k)
        else:
            eng_w[k] = ora_w[k] = v
    ref, gold = O.LlamaOracle(spec, ora_w, torch.bfloat16, keep_fp32=keep), O.LlamaOracle(spec, ora_w, torch.float32)
    eng = E.EmulEngine(spec, kv_pool_tokens=1024, weight_dtype=1).load_weights(eng_w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    g = torch.Generator().manual_seed(4)
    rc = gc = None
    for i, n in enumerate((300, 1)):
        x = (torch.randn(n, spec.hidden_size, generator=g) * 0.7).bfloat16()
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = eng.llm_step(s, x)
        assert eng.session_len(s) == len(rc)
        _three_way("fp8 prefill path", i, allr, rl, gl)
    eng.close()


def test_fp8_engine_block_path_and_chunks(E):
    """An fp8 engine (weight_dtype = 1) end to end in emulation: a 45-token first step through the 64-token block path
    (csrc/prefill.hip::gemm64_kernel<KF, EPI, WQ = 1>: one e4m3 -> bf16 expansion per fragment pair feeds four token tiles, scales on
    the reduced sums), then 11- and 1-row steps through the 16-row GEMV path of the same image, against the reference arithmetic on
    the dequantised weights (the band of tests/test_gpu_fp8.py); with VLO_BLOCK_PATH=0 semantics covered by the chunk steps."""
    from videollm_online_amd.checkpoint import quantize_fp8_per_channel
    spec = FP8_TINY
    w = O.init_llm_weights(spec, seed=11)
    eng_w, ora_w, keep = {}, {}, set()
    for k, v in w.items():
        if k.endswith(O.FP8_STREAMED):
            q, sc = quantize_fp8_per_channel(v)
            eng_w[k], eng_w[k + "_scale"] = q, sc
            ora_w[k] = q.float() * sc[:, None]
            keep.add(k)
        else:
            eng_w[k] = ora_w[k] = v
    ref, gold = O.LlamaOracle(spec, ora_w, torch.bfloat16, keep_fp32=keep), O.LlamaOracle(spec, ora_w, torch.float32)
    toks = O.default_tokens(spec, n_start=35)
    eng = E.EmulEngine(spec, weight_dtype=1).load_weights(eng_w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    s = eng.new_session()
    rc = gc = None
    for i, x in enumerate(_steps(spec, ref, toks, 12, [45, 11, 1])):
        rl, rc = ref.forward(x, rc)
        gl, gc = gold.forward(x, gc)
        last, allr = eng.llm_step(s, x)
        assert eng.session_len(s) == len(rc) and torch.equal(last, allr[-1])
        _three_way("fp8 tiny", i, allr, rl, gl)
    eng.close()


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (2.5 minutes; the hardware parity cases are tests/test_gpu_vit.py so400m B = 1 / 9 / 17)")
def test_vision_tower_with_padded_head_dim_mlp_width_and_patch_k(E):
    """The shapes of SigLIP-so400m/14 (BASELINE.json configs[4]) at toy size — head dim 72 (stored as 96 columns per head in q | k and 80
    rows per head in V^T, vit_attn_kernel<96, 80>), MLP width 336 (zero-padded to 512 at load), 14-pixel patches (K = 588 padded to
    640, element-wise im2col), 9 tokens per frame (not a multiple of 4) — against the oracle with the tolerance of the ViT tests."""
    vspec = O.VIT_SPECS["toy-hd72"]
    spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=vspec.hidden_size)
    w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=2)
    frames = O.synthetic_frames(2, vspec.image_size, seed=5)
    gold = O.LlamaOracle(spec, w, torch.float32).visual_embed(vw, vspec, frames)
    ref = O.LlamaOracle(spec, w, torch.bfloat16)
    amp = ref.visual_embed(vw, vspec, frames, mm_dtype=torch.float16)
    cpu = ref.visual_embed(vw, vspec, frames)
    eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    out = eng.visual_embed(frames)
    scale = gold.abs().max().item()
    e = (out.float() - gold).abs().max().item()
    a = (amp.float() - gold).abs().max().item()
    r = (cpu.float() - gold).abs().max().item()
    print(f"[emul vit hd72] engine err {e:.4g} fp16-autocast err {a:.4g} cpu-ref err {r:.4g} scale {scale:.3g}")
    assert e <= 2.0 * max(a, r) + 2 * 2 ** -8 * scale, (e, a, r)
    tok = eng.vision_tokens(frames)
    want = O.siglip_vision_encode(vw, vspec, frames)
    assert (tok.float() - want.float()).abs().max().item() <= 2.0 * (O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16).float()
                                                                    - want.float()).abs().max().item() + 2 * 2 ** -8 * want.abs().max().item()
    eng.close()



ONE_FRAME_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle import vlo_oracle as O
from tests.hip_emul import emul_engine as E
vspec = O.VitSpec(hidden_size=320, intermediate_size=1280, num_layers=1, num_heads=5, image_size=64, patch_size=16, pooled=(3, 3))
spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=vspec.hidden_size)
w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=4)
ref = O.LlamaOracle(spec, w, torch.bfloat16)
eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
import os
for B in ((1, 2) if os.environ.get("VLO_EMUL_FULL") == "1" else (1,)):
    frames = O.synthetic_frames(B, vspec.image_size, seed=7 + B)
    gold = O.LlamaOracle(spec, w, torch.float32).visual_embed(vw, vspec, frames)
    amp = ref.visual_embed(vw, vspec, frames, mm_dtype=torch.float16)
    cpu = ref.visual_embed(vw, vspec, frames)
    out = eng.visual_embed(frames)
    scale = gold.abs().max().item()
    e = (out.float() - gold).abs().max().item()
    a = (amp.float() - gold).abs().max().item()
    r = (cpu.float() - gold).abs().max().item()
    print(f"[emul one-frame path B={B}] engine err {e:.4g} fp16-autocast err {a:.4g} cpu-ref err {r:.4g} scale {scale:.3g}")
    assert e <= 2.0 * max(a, r) + 2 * 2 ** -8 * scale, (B, e, a, r)
eng.close()
print("OKPP")
"""


def test_one_frame_path_tall_tiles_split_attention_and_row_head_in_emulation(E):
    """The one-frame encode of round 6 (csrc/vit_tall.inc: 144 x 64 tiles with the half-tile software pipeline and counted waits, fc2 as 4 K slices
    into slabs; vit_attn.inc::vit_attn_split8_kernel; vit.hip::vit_im2col_kernel + the tall patch-embed GEMM; vit_rowvec_kernel in the MAP head) on
    the smallest tower that takes it (hidden 320 = 5 heads of 64, MLP 1280, 16 + 1 tokens), one frame (two with VLO_EMUL_FULL=1), against the oracle with the
    tolerance of the ViT tests — with the emulated direct-to-LDS loads landing as LATE as the hardware may land them (only at the counted
    s_waitcnt that retires them: a wait count one piece too generous reads stale shared memory and fails)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VLO_EMUL_GLDS="late")
    r = subprocess.run([sys.executable, "-c", ONE_FRAME_CHILD % root], env=env, capture_output=True, text=True, timeout=1800)
    print(r.stdout[-600:])
    assert r.returncode == 0 and "OKPP" in r.stdout, r.stderr[-2000:]


def test_unsupported_head_dims_are_rejected_not_silently_wrong(E):
    """vit_finalize takes head dim 64 exactly or 68..80 (padded storage).  52 / 56 / 60 also round up to 64 columns / rows, but the
    unpadded kernel would store 64 columns per head at a stride of hd — the engine must refuse them (round-3 advisor finding)."""
    import dataclasses
    base = O.VIT_SPECS["toy"]
    spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=base.hidden_size)
    for hd, nh in (((56, 8), (60, 32)) if FULL else ((56, 8),)):      # hidden sizes 448 / 1920: multiples of 64 the connector's GEMV has a plan for
        vs = dataclasses.replace(base, hidden_size=hd * nh, num_heads=nh, intermediate_size=128)
        sp = dataclasses.replace(spec, vision_hidden_size=vs.hidden_size)
        w, vw = O.init_llm_weights(sp, seed=3), O.init_vit_weights(vs, seed=2)
        eng = E.EmulEngine(sp, vit=vs)
        with pytest.raises(RuntimeError, match="vision tower shape"):
            eng.load_weights({**w, **vw}, O.rope_inv_freq(sp.head_dim, sp.rope_theta))
        eng.close()


def test_greedy_eos_leaves_no_logits_on_either_path(E):
    """vlo_greedy_generate after an EOS: the session holds no logits whether the EOS was found behind a speculative step (inside
    owned pages) or on the blocking path (no speculation: page / position boundary, or the last allowed token — max_new = 1 here)
    — a sampler call then fails with VLO_E_STATE on both (round-3 advisor finding)."""
    spec = TINY
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec)
    ref = O.LlamaOracle(spec, w, torch.bfloat16)
    eng = E.EmulEngine(spec).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
    x = ref.embed(torch.tensor(toks.stream_generation_ids))
    t0 = eng.greedy_generate(eng.new_session(), x, -1, 1)[0]             # the first token this prompt produces (eos = -1: never)
    for max_new in (4, 1):                                               # speculative path | blocking path
        s = eng.new_session()
        assert eng.greedy_generate(s, x, t0, max_new) == [t0]             # ... declared EOS: the response ends right there
        assert eng.session_len(s) == x.shape[0]                           # the EOS token is never fed to the model
        with pytest.raises(RuntimeError):
            eng.stream_sample(s, 0.725, toks.interval_id)
    eng.close()


GEMV_BATCH_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from tests.hip_emul import emul_engine as E
from videollm_online_amd.checkpoint import quantize_fp8_per_channel
K, N, n = 8192, 22 * 16 + 8, 11          # 23 column tiles (the last one half full) = 12 groups, the last with ONE tile, on VLO_GEMV_CUS = 2 blocks:
g = torch.Generator().manual_seed(8192)  # block 0 walks 6 groups (a batch of 4 + a batch of 2), block 1 likewise
x = torch.randn(n, K, generator=g).bfloat16()
W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
assert E.gemv_plan(K, False) == (8, 16, 2, 1), E.gemv_plan(K, False)        # 8 waves x 16 fragments, KC = 2 chunks, one K slice
y = E.test_gemv(x, W)
want = x.double() @ W.double().T
assert torch.allclose(y.double(), want, rtol=1e-4, atol=1e-4), (y.double() - want).abs().max()
q, s = quantize_fp8_per_channel(W)
y8 = E.test_gemv_fp8(x, q, s)
ref = x.double() @ (q.float().double() * s.double()[:, None]).T
assert (y8.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()) * (K / 256) ** 0.5 + 1e-5
torch.save((y, y8), sys.argv[1])
print("OKGEMV")
"""


def test_gemv_chunk_outer_batches_in_emulation(E, tmp_path):
    """K = 8192 as ONE K slice (the 70B model's qkv / o / gate-up / lm_head): 8 waves x 16 fragments x KC = 2 chunks.  The batch loop
    (gemv_body.inc GB = 4: chunk-outer over four groups, the activation fragments of a chunk built once per batch) against an fp64
    matmul, bf16 and fp8 images, full + partial batches and a one-tile last group — and bit-identical to the group-outer loop
    (VLO_GEMV_BATCH=0): same MFMAs in the same k order per output."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for batch in ("1", "0"):
        f = str(tmp_path / f"b{batch}.pt")
        env = dict(os.environ, VLO_TEST_GEMV_WHOLE_K="1", VLO_GEMV_CUS="2", VLO_GEMV_BATCH=batch)
        r = subprocess.run([sys.executable, "-c", GEMV_BATCH_CHILD % root, f], env=env, capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0 and "OKGEMV" in r.stdout, r.stderr[-2000:]
        outs[batch] = torch.load(f)
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])



@pytest.mark.parametrize("shape", ["hd128-g2", "hd64-g2", "hd128-g4"] if FULL else ["hd64-g2"])     # (hd 128 / GQA 4 also runs in the fp8 prefill test below)
def test_prefill_path_gemms_in_emulation(E, shape):
    """Inputs of >= 256 tokens take the prefill path (csrc/engine.hip::run_prefill): the projections as ping-pong GEMMs over the PACKED
    weight image (vit_gemm.inc instantiated for bf16 with the Llama epilogues: plain bf16, SwiGLU over the interleaved gate/up tile,
    bf16 residual read-modify-write), RoPE + KV append as a row kernel, flash-style attention over LDS-staged key tiles.  300 tokens (a full
    256-row tile + a partial one; the 128-row tile variant) then a frame step and a decode step on the cache the prefill wrote,
    all rows' logits 3-way against the oracle — with the emulated direct-to-LDS loads landing as late as the hardware may."""
    # (hidden, MLP, layers, heads, kv heads, vocab): every projection width a multiple of 256; the flash-style prefill attention
    # (llm_ops.hip::attn_prefill_kernel) for head dim 128 / 64 and GQA groups of 2 / 4
    spec = {"hd128-g2": O.LlmSpec(256, 256, 2, 2, 1, 256, 10000.0, 1e-5, vision_hidden_size=128),
            "hd64-g2": O.LlmSpec(256, 256, 1, 4, 2, 256, 10000.0, 1e-5, vision_hidden_size=128),
            "hd128-g4": O.LlmSpec(512, 256, 1, 4, 1, 256, 10000.0, 1e-5, vision_hidden_size=128)}[shape]
    w = O.init_llm_weights(spec, seed=9)
    toks = O.default_tokens(spec)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    os.environ["VLO_EMUL_GLDS"] = "late"
    try:
        eng = E.EmulEngine(spec, kv_pool_tokens=2048).load_weights(w, O.rope_inv_freq(spec.head_dim, spec.rope_theta))     # three sessions of <= 312 tokens
        s = eng.new_session()
        g = torch.Generator().manual_seed(4)
        rc = gc = None
        first = None
        for i, n in enumerate((300, 11, 1)):
            x = (torch.randn(n, spec.hidden_size, generator=g) * 0.7).bfloat16()
            rl, rc = ref.forward(x, rc)
            gl, gc = gold.forward(x, gc)
            last, allr = eng.llm_step(s, x)
            assert eng.session_len(s) == len(rc) and torch.equal(last, allr[-1])
            _three_way("prefill path", i, allr, rl, gl)
            first = first or (x, allr.clone())
        # the attention kernel's wave-uniform shortcuts (no causal select on fully visible tiles, no accumulator rescale when no maximum moved) are
        # bit-identical to the long way
        os.environ["VLO_ATTN_NOSKIP"] = "1"
        s2 = eng.new_session()
        _, again = eng.llm_step(s2, first[0])
        assert torch.equal(again, first[1])
        eng.close()
    finally:
        os.environ.pop("VLO_EMUL_GLDS", None)
        os.environ.pop("VLO_ATTN_NOSKIP", None)


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (95 s; tp_prefill runs in the default suite across two processes — "
                    "tests/test_multiproc_cpu.py::test_tp_prefill_path_two_processes_rccl_standin — and on hardware in tests/test_gpu_tp.py)")
@pytest.mark.parametrize("p2p", [False, True], ids=["sum-kernel", "p2p-group"])
def test_tensor_parallel_prefill_path_in_emulation(E, p2p):
    """csrc/tp.hip::tp_prefill with T = 2 logical ranks: inputs of >= 256 tokens run the ranks' SHARDS of the projections as GEMMs over the packed
    images (column-sharded q|k|v and gate|up, row-sharded o / down as fp32 partial matrices through the EP_LLM_F32 epilogue), RoPE + KV append and
    the prefill attention on each rank's own kv heads, the [m][H] fp32 all-reduce as the sum kernel, `h += bf16(sum); x = RMSNorm(h) w` as the row
    kernel, the vocabulary shards 16 rows at a time through the GEMV + the logits gather.  300 tokens (a 256-row tile + a partial one), then a frame
    step and a decode step of the 16-row TP pipeline on the cache the prefill wrote; all rows' logits 3-way against the oracle.  A group whose
    16-row exchanges are the p2p mailboxes takes the same prefill path (its matrix-sized all-reduce is the sum kernel: all ranks are local)."""
    spec = O.LlmSpec(256, 256, 1, 4, 2, 256, 10000.0, 1e-5, vision_hidden_size=128)     # per rank: 2 heads of 64 on 1 kv head, 128 MLP columns, 128 logits
    w = O.init_llm_weights(spec, seed=9)
    ref, gold = O.LlamaOracle(spec, w, torch.bfloat16), O.LlamaOracle(spec, w, torch.float32)
    os.environ["VLO_EMUL_GLDS"] = "late"
    try:
        grp = E.EmulTpGroup(spec, 2, w, O.rope_inv_freq(spec.head_dim, spec.rope_theta), p2p=p2p, kv_pool_tokens=2048)      # three sessions of <= 312 tokens
        s = grp.new_session()
        g = torch.Generator().manual_seed(4)
        rc = gc = None
        for i, n in enumerate((300, 11, 1)):
            x = (torch.randn(n, spec.hidden_size, generator=g) * 0.7).bfloat16()
            rl, rc = ref.forward(x, rc)
            gl, gc = gold.forward(x, gc)
            last, allr = grp.llm_step(s, x)
            assert grp.session_len(s) == len(rc) and torch.equal(last, allr[-1])
            _three_way(f"tp2 prefill {'p2p' if p2p else 'sum'}", i, allr, rl, gl)
        if FULL:        # only the last row wanted: the same logits as the all-rows call's last row (VLO_EMUL_FULL=1: two more 300-token passes)
            s2 = grp.new_session()
            g = torch.Generator().manual_seed(4)
            x = (torch.randn(300, spec.hidden_size, generator=g) * 0.7).bfloat16()
            last2, _ = grp.llm_step(s2, x, want_all=False)
            s3 = grp.new_session()
            last3, all3 = grp.llm_step(s3, x)
            assert torch.equal(last2, last3) and torch.equal(last3, all3[-1])
        if p2p:
            assert grp.p2p_status()["timed_out"] == 0
        grp.close()
    finally:
        os.environ.pop("VLO_EMUL_GLDS", None)


VIT_TILES_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle import vlo_oracle as O
from tests.hip_emul import emul_engine as E
vspec = O.VIT_SPECS["toy-hd72"]                  # head dim 72 (96 / 80 padded), 9 tokens per frame: one 32-key tile with a masked tail
spec = O.LlmSpec(128, 192, 1, 2, 2, 256, 10000.0, 1e-5, vision_hidden_size=vspec.hidden_size)
w, vw = O.init_llm_weights(spec, seed=3), O.init_vit_weights(vspec, seed=2)
frames = O.synthetic_frames(2, vspec.image_size, seed=5)
eng = E.EmulEngine(spec, vit=vspec).load_weights({**w, **vw}, O.rope_inv_freq(spec.head_dim, spec.rope_theta))
tok = eng.vision_tokens(frames)
torch.save(tok, sys.argv[1])
eng.close()
print("OKTILES")
"""


@pytest.mark.skipif(not FULL, reason="longer emulation cases: VLO_EMUL_FULL=1 (4 minutes: a 1152-wide tower on OS threads; the hardware parity cases are tests/test_gpu_vit.py so400m B = 9 / 17)")
def test_vit_padded_head_tile_streamed_attention_in_emulation(E, tmp_path):
    """vit_attn.inc::vit_attn_tiles_kernel (padded heads, batched frames: 256-query workgroups over LDS-staged 32-key tiles, direct-to-LDS
    loads in inline asm landing only at their vmcnt wait) forced onto the toy so400m-shaped tower (VLO_VIT_ATTN_TILES_MIN=1): inside
    the oracle's band and within fp16 noise of the 64-query-tile kernel it replaces at batched sizes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tiles_min in ("1", "0"):
        f = str(tmp_path / f"t{tiles_min}.pt")
        env = dict(os.environ, VLO_VIT_ATTN_TILES_MIN=tiles_min, VLO_EMUL_GLDS="late")
        r = subprocess.run([sys.executable, "-c", VIT_TILES_CHILD % root, f], env=env, capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0 and "OKTILES" in r.stdout, r.stderr[-2000:]
        outs[tiles_min] = torch.load(f).float()
    vspec = O.VIT_SPECS["toy-hd72"]
    vw = O.init_vit_weights(vspec, seed=2)
    frames = O.synthetic_frames(2, vspec.image_size, seed=5)
    want = O.siglip_vision_encode(vw, vspec, frames).float()
    amp = O.siglip_vision_encode(vw, vspec, frames, mm_dtype=torch.float16).float()
    band = 2.0 * (amp - want).abs().max().item() + 2 * 2 ** -8 * want.abs().max().item()
    assert (outs["1"] - want).abs().max().item() <= band
    assert (outs["1"] - outs["0"]).abs().max().item() <= 2 * 2 ** -8 * want.abs().max().item()
