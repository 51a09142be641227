"""Model shapes and synthetic inputs shared by bench.py, the CLI and the tests: no checkpoints, tokenizer files or videos
exist offline, so throughput runs use seeded random-init weights at the TRUE shapes, synthetic uint8 frames
and injected token ids."""
import torch

LLM_SHAPES = {
    "llama-3-8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                       num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0, rms_norm_eps=1e-5),
    # BASELINE.json configs[4]'s language model (fp8 weights: 69.5 GB, one MI355X holds it at TP = 1)
    "llama-3-70b": dict(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64,
                        num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0, rms_norm_eps=1e-5),
    "tinyllama-1.1b": dict(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32,
                           num_key_value_heads=4, vocab_size=32000, rope_theta=10000.0, rms_norm_eps=1e-5),
}
VIT_SHAPE = dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16)
# BASELINE.json configs[4]'s tower (HF google/siglip-so400m-patch14-384): head dim 72, 729 patches of 14 pixels
VIT_SHAPE_SO400M = dict(hidden_size=1152, intermediate_size=4304, num_layers=27, num_heads=16, image_size=384, patch_size=14)
VIT_SHAPES = {"siglip-l16-384": VIT_SHAPE, "siglip-so400m14-384": VIT_SHAPE_SO400M}


def vit_gflop_per_frame(v: dict, llm_hidden: int) -> float:
    """Encoder layers + patch embed + the MAP head's K / V projection + the 10-token connector (SURVEY.md §8d counts the same terms:
    384.4 for SigLIP-L into a 4096-wide LLM; this formula gives 384.2)."""
    D, I, L, P = v["hidden_size"], v["intermediate_size"], v["num_layers"], v["patch_size"]
    S = (v["image_size"] // P) ** 2
    enc = L * (2 * S * D * (4 * D + 2 * I) + 4 * S * S * D) + 2 * S * 3 * P * P * D + 2 * S * D * 2 * D
    conn = 2 * 10 * (D * llm_hidden + llm_hidden * llm_hidden)
    return (enc + conn) / 1e9


def gpu_random_weights(eng, cfg, seed=0):
    """Seeded random-init weights generated on the GPU and handed to the engine (device pointers)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    hd = H // nh

    def r(*shape, std=1.0, dtype=torch.bfloat16, mean=0.0):
        return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * std + mean).to(dtype)

    eng.load_weight("model.embed_tokens.weight", r(V, H))
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        eng.load_weight(p + "input_layernorm.weight", r(H, std=0.1, mean=1.0))
        eng.load_weight(p + "self_attn.q_proj.weight", r(nh * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.k_proj.weight", r(nkv * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.v_proj.weight", r(nkv * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.o_proj.weight", r(H, nh * hd, std=H ** -0.5))
        eng.load_weight(p + "post_attention_layernorm.weight", r(H, std=0.1, mean=1.0))
        eng.load_weight(p + "mlp.gate_proj.weight", r(I, H, std=H ** -0.5))
        eng.load_weight(p + "mlp.up_proj.weight", r(I, H, std=H ** -0.5))
        eng.load_weight(p + "mlp.down_proj.weight", r(H, I, std=I ** -0.5))
    eng.load_weight("model.norm.weight", r(H, std=0.1, mean=1.0))
    eng.load_weight("lm_head.weight", r(V, H, std=2 * H ** -0.5))
    Hv = cfg.vision_hidden_size
    eng.load_weight("connector.0.weight", r(H, Hv, std=Hv ** -0.5))
    eng.load_weight("connector.0.bias", r(H, std=0.1))
    eng.load_weight("connector.2.weight", r(H, H, std=H ** -0.5))
    eng.load_weight("connector.2.bias", r(H, std=0.1))
    if cfg.vit:
        v = cfg.vit
        D, Iv, P = v["hidden_size"], v["intermediate_size"], v["patch_size"]
        S = (v["image_size"] // P) ** 2
        f32 = torch.float32
        eng.load_weight("vision.embeddings.patch_embedding.weight", r(D, 3, P, P, std=(3 * P * P) ** -0.5, dtype=f32))
        eng.load_weight("vision.embeddings.patch_embedding.bias", r(D, std=0.1, dtype=f32))
        eng.load_weight("vision.embeddings.position_embedding.weight", r(S, D, std=0.5, dtype=f32))

        def ln(p):
            eng.load_weight(p + ".weight", r(D, std=0.1, mean=1.0, dtype=f32))
            eng.load_weight(p + ".bias", r(D, std=0.1, dtype=f32))

        def lin(p, o, i):
            eng.load_weight(p + ".weight", r(o, i, std=i ** -0.5, dtype=f32))
            eng.load_weight(p + ".bias", r(o, std=0.1, dtype=f32))

        for i in range(v["num_layers"]):
            p = f"vision.encoder.layers.{i}."
            ln(p + "layer_norm1"); ln(p + "layer_norm2")
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(p + "self_attn." + n, D, D)
            lin(p + "mlp.fc1", Iv, D); lin(p + "mlp.fc2", D, Iv)
        ln("vision.post_layernorm")
        eng.load_weight("vision.head.probe", r(1, 1, D, dtype=f32))
        eng.load_weight("vision.head.attention.in_proj_weight", r(3 * D, D, std=D ** -0.5, dtype=f32))
        eng.load_weight("vision.head.attention.in_proj_bias", r(3 * D, std=0.1, dtype=f32))
        lin("vision.head.attention.out_proj", D, D); ln("vision.head.layernorm")
        lin("vision.head.mlp.fc1", Iv, D); lin("vision.head.mlp.fc2", D, Iv)


def gpu_synthetic_frames(num_frames, res=384, seed=1234):
    """uint8 [T,3,R,R]: noise + a moving low-frequency gradient, generated directly in HBM."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.randint(0, 64, (num_frames, 3, res, res), generator=g, device="cuda", dtype=torch.int16)
    yy, xx = torch.meshgrid(torch.arange(res, device="cuda"), torch.arange(res, device="cuda"), indexing="ij")
    t = torch.arange(num_frames, device="cuda").view(-1, 1, 1)
    grad = (((xx[None] + 3 * t) % res + (yy[None] + 2 * t) % res) * 191 // (2 * res)).to(torch.int16)
    return (base + grad[:, None]).clamp_(0, 255).to(torch.uint8)


def stream_tokens(vocab, n_start=35, seed=7):
    """Injected ids standing in for the tokenizer-derived ones (demo/inference.py:33-35,42): start prompt, "\\n[", "]\\nAssistant:",
    eos, the interval token and one user query."""
    from .inference import StreamTokens
    g = torch.Generator().manual_seed(seed)
    eos = min(128009, vocab - 2)
    interval = 11 if vocab != 32000 else 29892

    def rnd(k):
        return [i + 1 if i in (eos, interval) else i for i in torch.randint(12, vocab - 4, (k,), generator=g).tolist()]

    return StreamTokens(start_ids=[min(128000, vocab - 3)] + rnd(n_start - 1), stream_prompt_ids=rnd(2),
                        stream_generation_ids=rnd(4), eos_token_id=eos, interval_id=interval,
                        query_ids={"Please narrate the video in real time.": rnd(12)})
