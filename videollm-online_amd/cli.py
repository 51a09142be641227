"""CLI benchmark loop mirroring demo/cli.py:12-50 on the HIP engine: load a video (uint8 [T,3,384,384] tensor file
or synthetic frames), one query at t=0, N iterations of ``input_video_stream(i / fps)`` + ``liveinfer()`` timed with
wall-clock, running-mean FPS, conversation history dumped as JSON with the reference's keys
(``role/content/time/fps/cost``).

    python -m videollm_online_amd.cli --frames 100 [--video frames.pt] [--out history.json]

Weights: seeded random-init at the Llama-3-8B + SigLIP-L shapes unless --base/--adapter/--siglip point at real
checkpoints (safetensors; LoRA merged at load, see checkpoint.py).  Without a tokenizer the responses are printed as
token ids."""
import argparse
import json
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)                 # demo/cli.py:31 runs 100 iterations
    ap.add_argument("--frame_fps", type=float, default=2.0)
    ap.add_argument("--video", default=None, help="torch-saved uint8 frames (.pt: [T,3,384,384] prepared, or decoded [T,H,W,3] of any size), or "
                                                  "a video file decoded by an ffmpeg binary through a pipe (ingest.open_video); default: synthetic frames")
    ap.add_argument("--query", default="Please narrate the video in real time.")
    ap.add_argument("--out", default="history.json")
    ap.add_argument("--base", default=None)
    ap.add_argument("--adapter", default=None)
    ap.add_argument("--siglip", default=None)
    ap.add_argument("--tokenizer", default=None, help="HF tokenizer directory (with the live chat template installed)")
    args = ap.parse_args(argv)

    import torch
    from . import synthetic as B                        # shapes, synthetic frames, random weights, token ids
    from .engine import Engine, EngineConfig
    from .inference import LiveInfer
    from .modeling_live import LiveModel

    cfg = EngineConfig(**B.LLM_SHAPES["llama-3-8b"], vision_hidden_size=1024, vit=B.VIT_SHAPE,
                       kv_pool_tokens=64 + 11 * args.frames + 120 * (args.frames + 2) + 4096)
    eng = Engine(cfg, 0)
    if args.base:
        from .checkpoint import load_engine_weights
        load_engine_weights(eng, args.base, args.adapter, args.siglip)
    else:
        B.gpu_random_weights(eng, cfg, seed=0)
    eng.finalize()
    tokenizer = None
    if args.tokenizer:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer)
    toks = None if tokenizer is not None else B.stream_tokens(cfg.vocab_size)
    eos = tokenizer.eos_token_id if tokenizer is not None else toks.eos_token_id
    interval = toks.interval_id if toks is not None else tokenizer.convert_tokens_to_ids(",")
    model = LiveModel(eng, eos_token_id=eos, frame_token_interval_id=interval)
    liveinfer = LiveInfer(model, tokens=toks, tokenizer=tokenizer, frame_fps=args.frame_fps)
    feed = None
    if args.video and not args.video.endswith(".pt"):
        from .ingest import open_video                  # demo/cli.py:13-22: ffmpeg at frame_fps; scale + pad happen on the device
        video, feed = open_video(eng, args.video, args.frame_fps)
    else:
        video = torch.load(args.video) if args.video else B.gpu_synthetic_frames(args.frames + 1)
    liveinfer.load_video(video)
    liveinfer.input_query_stream(args.query, video_time=0.0)                      # demo/cli.py:23
    if toks is not None and args.query not in toks.query_ids:
        toks.query_ids[args.query] = next(iter(toks.query_ids.values()))

    timecosts = []
    history = {"video_path": args.video or "synthetic", "frame_fps": args.frame_fps, "conversation": []}
    for i in range(args.frames):                                                   # demo/cli.py:31-48
        start_time = time.time()
        liveinfer.input_video_stream(i / liveinfer.frame_fps)
        query, response = liveinfer()
        end_time = time.time()
        timecosts.append(end_time - start_time)
        fps = (i + 1) / sum(timecosts)
        if query:
            history["conversation"].append({"role": "user", "content": query, "time": liveinfer.video_time, "fps": fps, "cost": timecosts[-1]})
            print(query)
        if response:
            history["conversation"].append({"role": "assistant", "content": response, "time": liveinfer.video_time, "fps": fps, "cost": timecosts[-1]})
            print(response)
        if not query and not response:
            history["conversation"].append({"time": liveinfer.video_time, "fps": fps, "cost": timecosts[-1]})
    if feed is not None:
        if feed.error is not None:
            raise RuntimeError(f"the video decoder failed: {feed.error}") from feed.error
        feed.proc.kill()                                 # the loop may stop before the file ends
    json.dump(history, open(args.out, "w"), indent=4)
    print(f"Average Processing FPS: {fps:.1f}.  The conversation history has been saved to {args.out}.")
    return history


if __name__ == "__main__":
    main()
