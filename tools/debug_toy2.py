import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vlo_oracle as O
from videollm_online_amd.engine import Engine, EngineConfig
from videollm_online_amd import _C
spec = O.LlmSpec(256, 704, 1, 4, 2, 1024, 10000.0, 1e-5, vision_hidden_size=128)
w = O.init_llm_weights(spec, seed=0)
cfg = EngineConfig(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                   num_attention_heads=spec.num_heads, num_key_value_heads=spec.num_kv_heads, vocab_size=spec.vocab_size,
                   rope_theta=spec.rope_theta, rms_norm_eps=spec.rms_eps, vision_hidden_size=spec.vision_hidden_size, kv_pool_tokens=1024)
eng = Engine(cfg); eng.load_weights(w); eng.load_weight("rope.inv_freq", O.rope_inv_freq(spec.head_dim, spec.rope_theta)); eng.finalize()
ref = O.LlamaOracle(spec, w, torch.bfloat16)
def rd(sess, which, rows, cols):
    t = torch.empty(16, cols, dtype=torch.bfloat16, device="cuda")
    _C.check(_C.lib().vlo_debug_read(sess._h, which, C.c_void_p(t.data_ptr()), 16 * cols * 2, None))
    torch.cuda.synchronize()
    return t[:rows].cpu().float()
sess = eng.new_session(); rc = None
g = torch.Generator().manual_seed(5)
for i, n in enumerate((16, 16, 13)):
    x = torch.randn(n, spec.hidden_size, generator=g).bfloat16()
    taps = {"_layers": (0,)}
    rl, rc = ref.forward(x, rc, taps)
    last, allr = eng.llm_step(sess, x.cuda(), want_all=True)
    torch.cuda.synchronize()
    q = rd(sess, 0, n, 256); at = rd(sess, 1, n, 256); h = rd(sess, 2, n, 256)
    qr = taps["q0"].transpose(0, 1).reshape(n, 256).float(); ar = taps["attn0"].float(); hr = taps["h0"].float()
    pr = lambda a, b: [round(float(e), 3) for e in (a - b).abs().max(dim=1).values]
    print(f"step {i}: q err {pr(q, qr)}\n        attn err {pr(at, ar)}\n        h err {pr(h, hr)}\n        logits err {pr(allr.cpu().float(), rl.float())}")
    if i == 1:
        e = (at - ar).abs()
        print("   attn err by head:", [round(float(e[:, hh*64:(hh+1)*64].max()), 3) for hh in range(4)])
        print("   attn err row1 cols>0.1:", torch.nonzero(e[1] > 0.1).flatten().tolist()[:40])
