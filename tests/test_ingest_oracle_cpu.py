"""Pins oracle/ingest_oracle.py: geometry against hand-evaluated ffmpeg filter expressions (data/utils.py:64), the antialiased
bicubic against torch.nn.functional.interpolate(mode="bicubic", antialias=True) (the same algorithm with a = -0.5)."""
import numpy as np
import pytest
import torch

from oracle import ingest_oracle as G


@pytest.mark.parametrize("iw,ih,R,want", [
    (1920, 1080, 384, (384, 216, 0, 84)),       # 16:9 landscape: h = round(384*1080/1920/2)*2 = 216, y0 = (384-216)/2 = 84
    (1080, 1920, 384, (216, 384, 84, 0)),       # portrait: the `else` branch scales the height
    (640, 480, 384, (384, 288, 0, 48)),
    (500, 375, 384, (384, 288, 0, 48)),
    (384, 384, 384, (384, 384, 0, 0)),          # square: gt(iw,ih) false -> height = R, width = -2 -> 384
    (854, 480, 384, (384, 216, 0, 84)),         # 384*480/854 = 215.8 -> /2 = 107.9 -> 108 -> 216
    (1280, 722, 384, (384, 216, 0, 84)),        # 216.6 -> 108.3 -> 108 -> 216
    (1000, 563, 384, (384, 216, 0, 84)),        # 216.19
    (1000, 570, 384, (384, 218, 0, 82)),        # 218.88 -> 109.44 -> 109 -> 218; (384-218)/2 = 83 -> chroma grid -> 82
    (320, 240, 384, (384, 288, 0, 48)),         # up-scaling
])
def test_ffmpeg_geometry(iw, ih, R, want):
    assert G.ffmpeg_scale_pad_geometry(iw, ih, R) == want


@pytest.mark.parametrize("H,W,oh,ow", [(108, 192, 54, 96), (97, 131, 40, 57), (40, 57, 97, 131), (64, 64, 64, 64), (300, 50, 20, 40)])
def test_bicubic_aa_matches_torch(H, W, oh, ow):
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mine = G.resize_bicubic_aa(img, oh, ow, a=-0.5)
    x = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = torch.nn.functional.interpolate(x, size=(oh, ow), mode="bicubic", antialias=True, align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(mine - ref).max() < 1e-6 * 255, np.abs(mine - ref).max()


def test_ingest_layout_identity_and_pad():
    rng = np.random.default_rng(0)
    fr = rng.integers(0, 256, (2, 384, 384, 3), dtype=np.uint8)
    out = G.ingest(fr, 384)
    assert np.array_equal(out, fr.transpose(0, 3, 1, 2))            # no resampling needed: bit-exact pass-through
    fr = rng.integers(0, 256, (1, 90, 160, 3), dtype=np.uint8)
    out = G.ingest(fr, 64)                                           # 160x90 -> 64x36, y0 = 14
    assert out.shape == (1, 3, 64, 64) and not out[:, :, :14].any() and not out[:, :, 50:].any() and out[:, :, 14:50].any()
