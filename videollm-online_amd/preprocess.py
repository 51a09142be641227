"""Offline visual-feature extraction — the mirror of ``data/utils.py:86-104 distributed_encode`` driven by
``data/preprocess/encode.py``: every video file under ``src_root`` becomes one ``.pt`` holding
``[T, frame_num_tokens, vision_hidden]`` (CLS + 3x3 pooled SigLIP tokens, bf16 when ``save_bf16``), which is what
``data/stream.py:91`` later loads instead of raw frames.

Same work split as the reference (file ``i`` goes to rank ``i % num_tasks``, no exchange between ranks, batches of 256
frames), but the encoder is the engine's HIP ViT (``Engine.vision_tokens``) and uploads overlap with encoding: the next
batch is staged through pinned memory on a copy stream while the current one runs.

Decoding: this image has no video decoder (no ffmpeg binary, torchvision, PyAV or rocDecode), so the native container
is a raw frame tensor — ``.pt`` / ``.npy`` holding uint8 ``[T, 3, R, R]`` (the output format of the reference's
``ffmpeg_once`` scale+pad step once decoded).  ``.mp4`` etc. are read with ``torchvision.io.read_video`` exactly as the
reference does when torchvision is importable, and rejected with a clear error otherwise."""
from __future__ import annotations

import os

import numpy as np
import torch

RAW_EXT = (".pt", ".npy")


def read_frames(path: str) -> torch.Tensor:
    """-> uint8 [T, 3, H, W] on the host (data/utils.py:98 uses torchvision.io.read_video(..., output_format='TCHW'))."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".pt":
        frames = torch.load(path, weights_only=True)
    elif ext == ".npy":
        frames = torch.from_numpy(np.load(path))
    else:
        try:
            import torchvision
        except ImportError as ex:
            raise RuntimeError(f"{path}: no video decoder available (torchvision is not installed); provide raw uint8 "
                               f"[T,3,R,R] frames as .pt/.npy") from ex
        frames = torchvision.io.read_video(path, pts_unit="sec", output_format="TCHW")[0]
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[1] != 3:
        raise ValueError(f"{path}: expected uint8 [T,3,H,W], got {frames.dtype} {tuple(frames.shape)}")
    return frames


def encode_frames(engine, frames: torch.Tensor, batch_size: int = 256) -> torch.Tensor:
    """Host uint8 [T,3,R,R] -> host bf16 [T, frame_num_tokens, vision_hidden]; H2D of batch i+1 overlaps encode of batch i."""
    T = frames.shape[0]
    dev = engine.device
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.Stream(device=dev)         # a real (non-null) stream: large batches run as two parallel branches (csrc/vit.hip)
    staged = None

    def stage(i):
        chunk = frames[i:i + batch_size]
        with torch.cuda.stream(copy_stream):
            d = chunk.pin_memory().to(dev, non_blocking=True) if not chunk.is_cuda else chunk
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return d, ev

    outs = []
    if T:
        staged = stage(0)
    for i in range(0, T, batch_size):
        cur, ev = staged
        staged = stage(i + batch_size) if i + batch_size < T else None
        main.wait_event(ev)
        with torch.cuda.stream(main):
            outs.append(engine.vision_tokens(cur, stream=main))
        cur.record_stream(main)
    main.synchronize()
    if not outs:
        return torch.empty(0, engine.cfg.frame_num_tokens, engine.cfg.vision_hidden_size, dtype=torch.bfloat16)
    return torch.cat(outs).cpu()


def output_root(src_root: str, embed_mark: str, vision_pretrained: str) -> str:
    """data/utils.py:91 — e.g. datasets/ego4d/v2/full_scale_2fps_384 + '2fps_384_1+3x3' + 'google/siglip-large-patch16-384'
    -> datasets/ego4d/v2/full_scale_2fps_384_1+3x3_google--siglip-large-patch16-384"""
    src_root = src_root.rstrip("/")
    return f"{src_root}_{embed_mark.split('_')[-1]}_{vision_pretrained.replace('/', '--')}"


def my_files(src_root: str, rank: int, world_size: int) -> list[str]:
    """data/utils.py:93-95: file i of the directory listing belongs to rank i % world_size.  The reference enumerates
    `os.listdir(src_root)` as the OS returns it (arbitrary but identical on every rank of one node); here the listing is SORTED so
    that ranks on different nodes, whose listings may be ordered differently, still partition the directory without overlap.
    Which rank encodes which video therefore differs from the reference; the set of output files and their contents do not."""
    return [f for i, f in enumerate(sorted(os.listdir(src_root))) if i % world_size == rank]


def distributed_encode(engine, *, src_root: str, vision_pretrained: str, embed_mark: str, batch_size: int = 256,
                       save_bf16: bool = False, rank: int | None = None, world_size: int | None = None, **kwargs) -> list[str]:
    """data/utils.py:86-104.  ``rank`` / ``world_size`` default to torchrun's RANK / WORLD_SIZE (the reference reads
    them from submitit's JobEnvironment).  Returns the paths written by this rank."""
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
    src_root = src_root.rstrip("/")
    dst_root = output_root(src_root, embed_mark, vision_pretrained)
    os.makedirs(dst_root, exist_ok=True)
    written = []
    for file in my_files(src_root, rank, world_size):
        frame_path = os.path.join(src_root, file)
        save_path = (os.path.splitext(frame_path)[0] + ".pt").replace(src_root, dst_root)
        feats = encode_frames(engine, read_frames(frame_path), batch_size)
        if not save_bf16:
            feats = feats.float()
        torch.save(feats, save_path)
        written.append(save_path)
    return written
