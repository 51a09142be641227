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


def test_raw_frame_reader_and_decoder_command():
    """The host half of the decoder pipe (videollm-online_amd/ingest.py): whole frames out of a byte stream in chunks, a short last
    chunk, a stream that ends inside a frame; the decoder argv is the reference's `ffmpeg -i src -r fps` (data/utils.py:62-64) minus
    its scale / pad filter, writing packed RGB24 to stdout."""
    import io
    import threading
    import pytest
    import torch
    from videollm_online_amd import ingest
    H, W, n = 6, 10, 7
    data = np.random.default_rng(0).integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    chunks = list(ingest.read_raw_frames(io.BytesIO(data.tobytes()), H, W, chunk=3))
    assert [c.shape[0] for c in chunks] == [3, 3, 1]
    assert np.array_equal(torch.cat(chunks).numpy(), data)
    assert list(ingest.read_raw_frames(io.BytesIO(b""), H, W, chunk=3)) == []
    with pytest.raises(IOError):
        list(ingest.read_raw_frames(io.BytesIO(data.tobytes()[:-5]), H, W, chunk=3))

    class DribblingPipe:                                   # a pipe returns what is there, not what was asked for
        def __init__(self, b):
            self.b, self.i = b, 0

        def read(self, k):
            k = min(k, 37)
            out = self.b[self.i:self.i + k]
            self.i += len(out)
            return out
    assert np.array_equal(torch.cat(list(ingest.read_raw_frames(DribblingPipe(data.tobytes()), H, W, chunk=4))).numpy(), data)

    cmd = ingest.decoder_command("v.mp4", 2, "./ffmpeg/ffmpeg")
    assert cmd[0] == "./ffmpeg/ffmpeg" and cmd[cmd.index("-i") + 1] == "v.mp4" and cmd[cmd.index("-r") + 1] == "2"
    assert cmd[-5:] == ["-f", "rawvideo", "-pix_fmt", "rgb24", "-"] and "-vf" not in cmd
    assert ingest.probe_command("v.mp4")[-1] == "v.mp4"

    class FakeRing:                                        # the feeder thread against a ring without a GPU
        layout, chunk = "THWC", 3

        def __init__(self):
            self.got, self.closed, self.frames = [], False, torch.zeros(1)
            self.H, self.W = H, W

        def wait_free(self, k, timeout=None):
            return True

        def push(self, f):
            self.got.append(f.clone())

        def close(self):
            self.closed = True
    import sys
    ring = FakeRing()
    code = "import sys; sys.stdout.buffer.write(bytes(range(256)) * 100)"      # 25 600 bytes = 142.2 frames of 180 bytes
    feed = ingest.DecoderFeed([sys.executable, "-c", code], ring)
    with pytest.raises(IOError):
        feed.join(30)
    assert ring.closed and sum(f.shape[0] for f in ring.got) == 141               # whole chunks before the torn one
    ring = FakeRing()
    ok = ingest.DecoderFeed([sys.executable, "-c", "import sys; sys.stdout.buffer.write(bytes(180 * 5))"], ring)
    ok.join(30)
    assert ok.frames == 5 and ring.closed
    bad_exit = ingest.DecoderFeed([sys.executable, "-c", "import sys; sys.stdout.buffer.write(bytes(180)); sys.exit(3)"], FakeRing())
    with pytest.raises(RuntimeError):
        bad_exit.join(30)

