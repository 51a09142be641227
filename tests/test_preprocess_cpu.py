"""Host logic of the offline feature extraction mirror (videollm-online_amd/preprocess.py vs data/utils.py:86-104):
naming, rank sharding, container checks — no GPU."""
import os

import numpy as np
import pytest
import torch

import videollm_online_amd  # noqa: F401  (package shim)
from videollm_online_amd import preprocess as P


def test_output_root_matches_reference_naming():
    assert P.output_root("datasets/ego4d/v2/full_scale_2fps_384/", "2fps_384_1+3x3", "google/siglip-large-patch16-384") == \
        "datasets/ego4d/v2/full_scale_2fps_384_1+3x3_google--siglip-large-patch16-384"


def test_rank_sharding_is_a_partition(tmp_path):
    for i in range(11):
        np.save(tmp_path / f"v{i:02d}.npy", np.zeros((1, 3, 4, 4), dtype=np.uint8))
    for world in (1, 2, 3, 8):
        shards = [P.my_files(str(tmp_path), r, world) for r in range(world)]
        flat = sorted(f for s in shards for f in s)
        assert flat == sorted(os.listdir(tmp_path)) and len(set(flat)) == 11
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_read_frames_containers(tmp_path):
    fr = torch.randint(0, 255, (5, 3, 8, 8), dtype=torch.uint8)
    torch.save(fr, tmp_path / "a.pt")
    np.save(tmp_path / "b.npy", fr.numpy())
    assert torch.equal(P.read_frames(str(tmp_path / "a.pt")), fr)
    assert torch.equal(P.read_frames(str(tmp_path / "b.npy")), fr)
    torch.save(fr.float(), tmp_path / "bad.pt")
    with pytest.raises(ValueError):
        P.read_frames(str(tmp_path / "bad.pt"))
    (tmp_path / "c.mp4").write_bytes(b"\x00")
    try:
        import torchvision  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="no video decoder"):
            P.read_frames(str(tmp_path / "c.mp4"))
