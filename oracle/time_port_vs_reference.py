"""How fast is the oracle PORT next to the REFERENCE's own classes on the same CPU?  (TEST INFRASTRUCTURE; build container only —
needs /root/reference.)  bench.py's `cpu_baseline` times the port (kind = "port": the reference cannot travel to the GPU box); this
script times both on identical inputs and weights — the reference's LiveLlamaForCausalLM / fast_greedy_generate /
_siglip_vision_encode (imported as oracle/make_golden.py does) and oracle/vlo_oracle.py — so that the "port" label is backed by a
measured ratio.  Output: profiles/r3_port_vs_reference_cpu.txt.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_port_vs_reference.py [threads]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden as MG  # noqa: E402
from oracle import vlo_oracle as O  # noqa: E402


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


@torch.no_grad()
def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    (LiveLlamaConfig, LiveLlamaForCausalLM, ref_generate, ref_encode, SiglipVisionConfig, SiglipVisionModel) = MG.import_reference()
    lines = [f"# oracle port vs the reference's own classes, same CPU ({threads} threads, torch {torch.__version__}), same weights and inputs; best of 3"]
    # language model: TinyLlama-1.1B width, 4 layers (BASELINE configs[0]'s model family; the per-layer cost is what is compared)
    spec = O.LlmSpec(2048, 5632, 4, 32, 4, 32000, 10000.0, 1e-5)
    w = O.init_llm_weights(spec, seed=3)
    toks = O.default_tokens(spec)
    ref = MG.build_ref_llm(LiveLlamaConfig, LiveLlamaForCausalLM, spec, w, toks.interval_id, toks.eos_token_id, torch.bfloat16)
    port = O.LlamaOracle(spec, w, torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    H = spec.hidden_size
    for Lc in (0, 330, 2048):
        # bring both caches to Lc cached tokens
        past = cache = None
        if Lc:
            x = torch.randn(Lc, H, generator=g).bfloat16()
            past = ref(inputs_embeds=x[None], use_cache=True).past_key_values
            _, cache = port.forward(x, None, logits_from=Lc)
        for n, what in ((11, "frame step n=11"), (1, "decode step n=1")):
            x = torch.randn(n, H, generator=g).bfloat16()

            def run_ref():
                import copy
                p = copy.deepcopy(past) if past is not None else None
                t0 = time.perf_counter()
                ref(inputs_embeds=x[None], use_cache=True, past_key_values=p)
                return time.perf_counter() - t0

            def run_port():
                c = None
                if cache is not None:
                    c = O.KVCacheOracle(spec.num_layers)
                    c.k, c.v = list(cache.k), list(cache.v)
                t0 = time.perf_counter()
                port.forward(x, c)
                return time.perf_counter() - t0
            tr = min(run_ref() for _ in range(3))
            tp = min(run_port() for _ in range(3))
            lines.append(f"LLM {what:16s} Lc={Lc:5d}: reference {tr * 1e3:8.2f} ms   port {tp * 1e3:8.2f} ms   port/reference {tp / tr:.2f}")
    # vision tower: SigLIP-L/16-384 shape, 4 layers, one frame, fp32 (the CPU path: autocast is a no-op there)
    vspec = O.VitSpec(num_layers=4)
    vw = O.init_vit_weights(vspec, seed=1)
    vit = MG.build_ref_vit(SiglipVisionConfig, SiglipVisionModel, vspec, vw)
    frames = O.synthetic_frames(1, vspec.image_size, seed=1234)
    tr = best(lambda: ref_encode(vit, frames, frame_token_cls=True, frame_token_pooled=[3, 3]))
    tp = best(lambda: O.siglip_vision_encode(vw, vspec, frames, None))
    lines.append(f"ViT 4 layers, 1 frame, fp32      : reference {tr * 1e3:8.2f} ms   port {tp * 1e3:8.2f} ms   port/reference {tp / tr:.2f}")
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(os.path.dirname(HERE), "profiles"), exist_ok=True)
    open(os.path.join(os.path.dirname(HERE), "profiles", "r3_port_vs_reference_cpu.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
