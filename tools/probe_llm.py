"""Quick LLM-step timing probe on the true Llama-3-8B shape (random weights generated on the GPU)."""
import argparse
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videollm_online_amd.engine import Engine, EngineConfig

SHAPES = {
    "llama-3-8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                       num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0),
    "tinyllama-1.1b": dict(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32,
                           num_key_value_heads=4, vocab_size=32000, rope_theta=10000.0),
}


def random_llm_weights_to_engine(eng, cfg, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    hd = H // cfg.num_attention_heads
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads

    def r(*shape, std):
        return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * std).bfloat16()

    eng.load_weight("model.embed_tokens.weight", r(V, H, std=1.0))
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        eng.load_weight(p + "input_layernorm.weight", (1 + r(H, std=0.1).float()).bfloat16())
        eng.load_weight(p + "self_attn.q_proj.weight", r(nh * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.k_proj.weight", r(nkv * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.v_proj.weight", r(nkv * hd, H, std=H ** -0.5))
        eng.load_weight(p + "self_attn.o_proj.weight", r(H, nh * hd, std=H ** -0.5))
        eng.load_weight(p + "post_attention_layernorm.weight", (1 + r(H, std=0.1).float()).bfloat16())
        eng.load_weight(p + "mlp.gate_proj.weight", r(I, H, std=H ** -0.5))
        eng.load_weight(p + "mlp.up_proj.weight", r(I, H, std=H ** -0.5))
        eng.load_weight(p + "mlp.down_proj.weight", r(H, I, std=I ** -0.5))
    eng.load_weight("model.norm.weight", (1 + r(H, std=0.1).float()).bfloat16())
    eng.load_weight("lm_head.weight", r(V, H, std=2 * H ** -0.5))
    Hv = cfg.vision_hidden_size
    eng.load_weight("connector.0.weight", r(H, Hv, std=Hv ** -0.5))
    eng.load_weight("connector.0.bias", r(H, std=0.1))
    eng.load_weight("connector.2.weight", r(H, H, std=H ** -0.5))
    eng.load_weight("connector.2.bias", r(H, std=0.1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--frames", type=int, default=1200)
    args = ap.parse_args()
    cfg = EngineConfig(**SHAPES[args.model], kv_pool_tokens=16384)
    t0 = time.time()
    eng = Engine(cfg)
    random_llm_weights_to_engine(eng, cfg)
    eng.finalize()
    torch.cuda.synchronize()
    print(f"engine ready in {time.time()-t0:.1f}s, packed weights {eng.weight_bytes/1e9:.2f} GB")
    sess = eng.new_session()
    H = cfg.hidden_size
    x45 = torch.randn(45, H, device="cuda").bfloat16()
    x11 = torch.randn(11, H, device="cuda").bfloat16()
    x1 = torch.randn(1, H, device="cuda").bfloat16()
    eng.llm_step(sess, x45)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.frames + 1)]
    t0 = time.time()
    ev[0].record()
    for i in range(args.frames):
        eng.llm_step(sess, x11)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.time() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.frames)]
    L = sess.get_seq_length()
    print(f"{args.frames} frame steps (n=11): wall {wall:.3f}s = {args.frames/wall:.1f} steps/s; final Lc={L}")
    for lo in (0, args.frames // 2, args.frames - 50):
        seg = ms[lo:lo + 50]
        Lc = 45 + 11 * lo
        b = eng.step_algorithmic_bytes(Lc, 11)
        print(f"  steps {lo}..{lo+50}: {sum(seg)/len(seg):.3f} ms/step  Lc~{Lc}  alg {b/1e9:.2f} GB -> {b/ (sum(seg)/len(seg)*1e-3)/1e12:.2f} TB/s")
    # decode steps at long context
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(33)]
    ev[0].record()
    for i in range(32):
        eng.llm_step(sess, x1)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(32)]
    b = eng.step_algorithmic_bytes(L, 1)
    print(f"decode n=1 at Lc={L}: {sum(ms)/32:.3f} ms/step -> {b/(sum(ms)/32*1e-3)/1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
