"""CPU oracle (numpy) of the video-ingest geometry and resampling that feed the hot path — TEST INFRASTRUCTURE ONLY (imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package).

What it restates: the frame preparation the reference performs with an external ffmpeg binary before `load_video`
(/root/reference/data/utils.py:51-66 `ffmpeg_once`, called from /root/reference/demo/cli.py:13-22):

    -sws_flags bicubic  -vf "scale='if(gt(iw,ih),R,-2)':'if(gt(iw,ih),-2,R)',pad=R:R:(ow-iw)/2:(oh-ih)/2:color='#000000'"

i.e. scale the LONGER side to R keeping the aspect ratio (the other side rounded to a multiple of 2), bicubic, then centre
on a black R x R canvas.  The arithmetic lives in third-party code that is not under /root/reference (the `./ffmpeg/ffmpeg`
binary the reference downloads, unpinned: README.md:61-66), so the published algorithms are restated:
  * output size of a negative scale argument: libavfilter/scale_eval.c `ff_scale_adjust_dimensions`:
        h = av_rescale(w, ih, iw * 2) * 2        (av_rescale rounds to nearest, halves away from zero)
  * pad offsets: libavfilter/vf_pad.c evaluates `(ow-iw)/2` in double, truncates, then rounds DOWN to the chroma
    subsampling grid of the frame format (yuv420p for mp4/H.264 input: multiples of 2);
  * bicubic: libswscale's SWS_BICUBIC is the Mitchell-Netravali family with B = 0, C = 0.6 (libswscale/utils.c initFilter), i.e.
    the Keys cubic with a = -0.6, with the filter support stretched by the down-scale factor (area-aware, "antialiased") and
    pixel-centre alignment.  The separable evaluation below uses the tap selection / normalisation of PIL's and torch's
    antialiased resize (`_compute_indices_weights_aa`), which is the same continuous filter; swscale evaluates it in 14-bit
    fixed point on YUV planes and the result is then H.264-encoded, so the reference pipeline is NOT bit-reproducible —
    PARITY UNPINNED against ffmpeg itself (no binary here).  The restatement is pinned instead against
    `torch.nn.functional.interpolate(mode="bicubic", antialias=True)` (which is this algorithm with a = -0.5) in
    tests/test_ingest_oracle_cpu.py, and the geometry against hand-evaluated ffmpeg expressions."""
import numpy as np


def av_rescale_near(a: int, b: int, c: int) -> int:
    """libavutil av_rescale(a, b, c) = a * b / c rounded to nearest, halves away from zero (AV_ROUND_NEAR_INF), positive inputs."""
    return (a * b + c // 2) // c


def ffmpeg_scale_pad_geometry(iw: int, ih: int, R: int):
    """(ow, oh, x0, y0): scaled size and top-left pad offset on the R x R canvas (data/utils.py:64)."""
    if iw > ih:
        ow = R
        oh = av_rescale_near(ow, ih, iw * 2) * 2
    else:
        oh = R
        ow = av_rescale_near(oh, iw, ih * 2) * 2
    ow, oh = max(ow, 2), max(oh, 2)
    if ow > R or oh > R:            # cannot happen for the longer-side rule; vf_pad would fail
        raise ValueError("scaled frame exceeds the pad canvas")
    x0 = ((R - ow) // 2) & ~1       # yuv420p: offsets rounded down to the 2x2 chroma grid
    y0 = ((R - oh) // 2) & ~1
    return ow, oh, x0, y0


def keys_cubic(x, a):
    x = np.abs(x)
    return np.where(x < 1, ((a + 2) * x - (a + 3)) * x * x + 1, np.where(x < 2, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def aa_taps(in_size: int, out_size: int, a: float):
    """Per output index: (first input index, normalised float64 weights) — antialiased separable cubic, pixel-centre aligned."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    taps = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        j = np.arange(xmin, xmax)
        w = keys_cubic((j - center + 0.5) * inv, a)
        s = w.sum()
        taps.append((xmin, w / s if s != 0 else w))
    return taps


def resize_bicubic_aa(img: np.ndarray, oh: int, ow: int, a: float = -0.6) -> np.ndarray:
    """img [H, W, C] (any real dtype) -> float64 [oh, ow, C]; horizontal pass then vertical pass, no intermediate rounding."""
    H, W, C = img.shape
    x = img.astype(np.float64)
    th = aa_taps(W, ow, a)
    tmp = np.empty((H, ow, C))
    for o, (x0, w) in enumerate(th):
        tmp[:, o] = np.tensordot(x[:, x0:x0 + len(w)], w, axes=([1], [0]))
    tv = aa_taps(H, oh, a)
    out = np.empty((oh, ow, C))
    for o, (y0, w) in enumerate(tv):
        out[o] = np.tensordot(w, tmp[y0:y0 + len(w)], axes=([0], [0]))
    return out


def ingest(frames_hwc_u8: np.ndarray, R: int = 384, a: float = -0.6, pad_rgb=(0, 0, 0)) -> np.ndarray:
    """uint8 [T, H, W, 3] decoded RGB frames -> uint8 [T, 3, R, R] (what read_video(..., output_format='TCHW') yields for the
    ffmpeg-prepared file, demo/inference.py:112)."""
    T, H, W, _ = frames_hwc_u8.shape
    ow, oh, x0, y0 = ffmpeg_scale_pad_geometry(W, H, R)
    out = np.empty((T, 3, R, R), dtype=np.uint8)
    out[:] = np.asarray(pad_rgb, dtype=np.uint8)[None, :, None, None]
    for t in range(T):
        r = resize_bicubic_aa(frames_hwc_u8[t], oh, ow, a)
        out[t, :, y0:y0 + oh, x0:x0 + ow] = np.clip(np.floor(r + 0.5), 0, 255).astype(np.uint8).transpose(2, 0, 1)
    return out
