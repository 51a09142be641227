import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vlo_oracle as O
from videollm_online_amd.engine import Engine, EngineConfig
name = sys.argv[1] if len(sys.argv) > 1 else "toy"
spec = O.LLM_SPECS[name]
w = O.init_llm_weights(spec, seed=0)
cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                   num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                   rope_theta=spec.rope_theta, rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=1024)
eng = Engine(cfg); eng.load_weights(w); eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta)); eng.finalize()
ref = O.LlamaOracle(spec, w, torch.bfloat16)
for n in (1, 5, 11, 16):
    sess = eng.new_session()
    x = torch.randn(n, spec.hidden_size, generator=torch.Generator().manual_seed(n)).bfloat16()
    taps = {"_layers": (0,)}
    rl, rc = ref.forward(x, None, taps)
    last, allr = eng.llm_step(sess, x.cuda(), want_all=True)
    torch.cuda.synchronize()
    allr = allr.cpu().float()
    k = sess.read_kv(0, 0, 0, 0, n).cpu().float(); v = sess.read_kv(0, 1, 0, 0, n).cpu().float()
    print(f"n={n}: logits err {(allr - rl.float()).abs().max():.4f} (|ref| {rl.float().abs().max():.2f}, |eng| {allr.abs().max():.2f}, nan {torch.isnan(allr).any().item()})"
          f"  K0 err {(k - rc.k[0][0].float()).abs().max():.4f}  V0 err {(v - rc.v[0][0].float()).abs().max():.4f}")
    sess.close()
print("--- multi-step")
sess = eng.new_session(); rc = None
g = torch.Generator().manual_seed(5)
for i, n in enumerate((16, 16, 13, 1, 11)):
    x = torch.randn(n, spec.hidden_size, generator=g).bfloat16()
    rl, rc = ref.forward(x, rc)
    last, allr = eng.llm_step(sess, x.cuda(), want_all=True)
    torch.cuda.synchronize()
    allr = allr.cpu().float()
    L = len(rc)
    k = sess.read_kv(0, 0, 0, 0, L).cpu().float(); v = sess.read_kv(0, 1, 0, 0, L).cpu().float()
    k1 = sess.read_kv(1, 0, 1, 0, L).cpu().float()
    for ly in range(spec.num_layers):
        for hh in range(spec.num_kv_heads):
            kk_ = sess.read_kv(ly, 0, hh, 0, L).cpu().float(); vv_ = sess.read_kv(ly, 1, hh, 0, L).cpu().float()
            ke = (kk_ - rc.k[ly][hh].float()).abs().max(dim=1).values; ve = (vv_ - rc.v[ly][hh].float()).abs().max(dim=1).values
            bad = [int(t) for t in torch.nonzero((ke > 0.1) | (ve > 0.1)).flatten()]
            if bad: print(f"   layer {ly} kvh {hh}: bad token rows {bad}")
    rowerr = (allr - rl.float()).abs().max(dim=1).values
    print(f"step {i} n={n} L={L}: row errs {[round(float(e),3) for e in rowerr]}  K0 {(k - rc.k[0][0].float()).abs().max():.4f} V0 {(v - rc.v[0][0].float()).abs().max():.4f} K1h1 {(k1 - rc.k[1][1].float()).abs().max():.4f}")
