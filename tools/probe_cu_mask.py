"""XCD / CU partition between the encoder stream and the Llama stream (hipExtStreamCreateWithCUMask).

north_star's pipeline runs encode(t + 1) on the encode stream BESIDE the Llama step of frame t.  Unpartitioned, the encoder's workgroups (whole CUs:
156 KiB of LDS) land on the CUs the GEMV's 256 persistent blocks sit on and the step pays ~0.8 ms per frame for a 1.6 ms encode.
This probe gives each stream its own CUs — the encoder N whole XCDs (or an even spread of 32 N CUs), the Llama step the rest, its GEMV grid sized
to match (VLO_GEMV_CUS, read once per process: one configuration per run) — and times step alone / encode alone / the pipelined pair.

    python tools/probe_cu_mask.py --vit-xcds 0            # today's shared chip
    python tools/probe_cu_mask.py --vit-xcds 3 [--spread]
"""
import argparse
import ctypes as C
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--vit-xcds", type=int, default=0, help="XCDs (32 CUs each) given to the encode stream; 0 = no partition")
ap.add_argument("--spread", action="store_true", help="the same number of CUs taken evenly from all XCDs instead of whole XCDs")
ap.add_argument("--kv", type=int, default=15500)
ap.add_argument("--iters", type=int, default=60)
args = ap.parse_args()
NV = args.vit_xcds
if NV:
    os.environ["VLO_GEMV_CUS"] = str(256 - 32 * NV)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from probe_llm import SHAPES, random_llm_weights_to_engine  # noqa: E402
from probe_vit import load_random_vit  # noqa: E402
from videollm_online_amd.engine import Engine, EngineConfig  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    """a HIP stream whose kernels run only on the CUs whose mask bit is set (bit i -> XCD i % 8, CU i / 8 of that XCD: the KFD spreads the mask
    round-robin over the XCCs)"""
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value)


if NV:
    if args.spread:
        vit_bits = {i for i in range(256) if (i // 8) < 4 * NV}           # 4 N CUs of every XCD
    else:
        vit_bits = {i for i in range(256) if (i % 8) < NV}                # XCDs 0 .. N-1 whole
    enc, main = masked_stream(vit_bits), masked_stream(set(range(256)) - vit_bits)
else:
    enc, main = torch.cuda.Stream(), torch.cuda.Stream()

cfg = EngineConfig(**SHAPES["llama-3-8b"], kv_pool_tokens=args.kv + 4096,
                   vit=dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=384, patch_size=16))
eng = Engine(cfg)
random_llm_weights_to_engine(eng, cfg)
load_random_vit(eng)
eng.finalize()
H = cfg.hidden_size
sess = eng.new_session()
with torch.cuda.stream(main):
    eng.llm_step(sess, (torch.randn(args.kv, H, device="cuda") * 0.5).bfloat16())
    torch.cuda.synchronize()
frame = torch.randint(0, 256, (1, 3, 384, 384), dtype=torch.uint8, device="cuda")
x11 = (torch.randn(11, H, device="cuda") * 0.5).bfloat16()
n0 = sess.get_seq_length()


def step():
    with torch.cuda.stream(main):
        eng.llm_step(sess, x11, stream=main)


def encode():
    with torch.cuda.stream(enc):
        eng.visual_embed(frame, stream=enc)


def timed(fn, streams, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_event(e0)
    for _ in range(iters):
        fn()
    ends = []
    for s in streams:
        ev = torch.cuda.Event()
        ev.record(s)
        ends.append(ev)
    for ev in ends:
        torch.cuda.current_stream().wait_event(ev)
    e1.record()
    torch.cuda.synchronize()
    sess.crop(n0)
    return e0.elapsed_time(e1) / iters


prev = [None]


def pipelined():
    # frame t's step waits for frame t's embedding (encoded during step t - 1); encode(t + 1) runs beside step(t)
    if prev[0] is not None:
        main.wait_event(prev[0])
    encode()
    ev = torch.cuda.Event()
    ev.record(enc)
    prev[0] = ev
    step()


tag = "shared chip" if not NV else (f"{32 * NV} CUs spread over all XCDs" if args.spread else f"{NV} whole XCD(s)") + f" for the encoder, GEMV grid {256 - 32 * NV}"
t_step = timed(step, [main], args.iters)
t_enc = timed(encode, [enc], args.iters)
t_pipe = timed(pipelined, [main, enc], args.iters)
print(f"[{tag}] KV {n0}: step(n=11) alone {t_step:.3f} ms | one-frame encode alone {t_enc:.3f} ms | pipelined pair {t_pipe:.3f} ms per frame "
      f"(serial would be {t_step + t_enc:.3f})", flush=True)
