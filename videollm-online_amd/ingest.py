"""Frame ingest in front of ``LiveInfer``: decoded frames arriving on the HOST (any size, RGB24 as a decoder emits them) become
the uint8 [3,R,R] frames the vision tower consumes, through a pinned-memory -> device ring, so a stream never needs the whole
video resident in HBM.

The reference prepares the file with an external ffmpeg and then loads ALL of it to the GPU (demo/cli.py:13-22 ->
data/utils.py:51-66; demo/inference.py:111-115 `read_video(...).to('cuda')`): 442 KB per frame, 1.6 GB per hour at 2 FPS.
Here a decoder thread / process calls ``ring.push(frames)``; the copy engine uploads the raw frames from pinned staging
buffers on its own HIP stream, `vlo_frame_ingest` (csrc/ingest.hip) scales + pads them into the ring, and
``LiveInfer.input_video_stream`` encodes windows of the ring.  Video DECODING itself (mp4 demux, H.264) is out of reach in
this image — no ffmpeg / torchvision / PyAV / rocDecode — so the ring starts at decoded frames.

Where a decoder binary exists, ``DecoderFeed`` is the process boundary to it: the reference's own ``ffmpeg -i src -r fps``
(data/utils.py:51-66) WITHOUT its scale / pad filter (that part runs on the device) writing packed RGB24 frames to a pipe, a
feeder thread reading whole frames from the pipe and pushing them into the ring under back-pressure.  Any program that writes
raw RGB24 frames to stdout can stand in for ffmpeg (the tests use a Python one)."""
import os
import shutil
import subprocess
import threading
import time

import torch


class FrameRing:
    def __init__(self, engine, height: int, width: int, capacity: int = 64, chunk: int = 8, slots: int = 3, layout: str = "THWC",
                 cubic_a: float = -0.6, resolution: int = 0):
        if layout not in ("THWC", "TCHW"):
            raise ValueError("layout must be THWC or TCHW")
        self.engine, self.layout, self.cubic_a = engine, layout, cubic_a
        self.R = resolution or engine.cfg.vit["image_size"]
        self.H, self.W, self.capacity, self.chunk = height, width, capacity, chunk
        dev = engine.device
        shape = (chunk, height, width, 3) if layout == "THWC" else (chunk, 3, height, width)
        self._pinned = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self._raw = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self._slot_done = [None] * slots                 # event: the slot's upload + ingest have finished
        self._slot = 0
        self.frames = torch.zeros(capacity, 3, self.R, self.R, dtype=torch.uint8, device=dev)    # the ring itself
        self.stream = torch.cuda.Stream(dev)             # copy + ingest stream
        self.head = 0                                    # frames pushed so far (absolute index of the next frame)
        self.tail = 0                                    # frames below this index may be overwritten
        self._ready = {}                                 # absolute frame index -> event (prepared frame is in the ring)
        self._released = None                            # event after which released positions are no longer read
        self.closed = False                              # no more frames will come
        self.feed = None                                 # the DecoderFeed filling this ring, if any
        self._cv = threading.Condition()                 # push() may run on a feeder thread, window() / release() on the consumer's

    def __len__(self):
        return self.head

    def free(self) -> int:
        return self.capacity - (self.head - self.tail)

    def push(self, frames: torch.Tensor) -> int:
        """``frames``: uint8 host tensor [n, H, W, 3] (or [n, 3, H, W] for layout TCHW), n <= chunk.  Returns the absolute index
        of the first frame.  Raises when the ring is full: the consumer has to release() frames first (back-pressure)."""
        n = frames.shape[0]
        if n == 0:
            return self.head
        if n > self.chunk or tuple(frames.shape[1:]) != tuple(self._pinned[0].shape[1:]) or frames.dtype != torch.uint8:
            raise ValueError(f"push takes uint8 {tuple(self._pinned[0].shape[1:])} frames, at most {self.chunk} at a time")
        if n > self.free():
            raise BufferError(f"frame ring full ({self.capacity} frames, {self.free()} free): release() consumed frames first")
        k = self._slot
        self._slot = (k + 1) % len(self._pinned)
        if self._slot_done[k] is not None:
            self._slot_done[k].synchronize()             # the staging slot's previous upload has left it
        self._pinned[k][:n].copy_(frames)                # host memcpy into pinned memory
        first = self.head
        with torch.cuda.stream(self.stream):
            if self._released is not None:
                self.stream.wait_event(self._released)   # the encode that read the positions about to be overwritten
            self._raw[k][:n].copy_(self._pinned[k][:n], non_blocking=True)
            done = 0
            while done < n:                              # a chunk may wrap around the end of the ring
                pos = (first + done) % self.capacity
                m = min(n - done, self.capacity - pos)
                self.engine.frame_ingest(self._raw[k][done:done + m], self.layout, self.R, self.cubic_a,
                                         out=self.frames[pos:pos + m], stream=self.stream)
                done += m
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_done[k] = ev
        with self._cv:
            for i in range(n):
                self._ready[first + i] = ev
            self.head += n
            self._cv.notify_all()
        return first

    def wait_for(self, count: int, timeout: float | None = None) -> bool:
        """Block until ``count`` frames have been pushed (True) or the ring was closed / the timeout passed with fewer (False)."""
        with self._cv:
            return self._cv.wait_for(lambda: self.head >= count or self.closed, timeout) and self.head >= count

    def wait_free(self, n: int, timeout: float | None = None) -> bool:
        """Block until ``n`` positions are free (the consumer released frames)."""
        with self._cv:
            return self._cv.wait_for(lambda: self.free() >= n or self.closed, timeout) and self.free() >= n

    def window(self, lo: int, hi: int):
        """(uint8 [hi-lo,3,R,R] device tensor, event to wait on) for frames [lo, hi) — all of them pushed and not yet released."""
        with self._cv:
            if not (self.tail <= lo < hi <= self.head):
                raise IndexError(f"frames [{lo}, {hi}) are not in the ring (holds [{self.tail}, {self.head}))")
            ev = self._ready[hi - 1]                     # pushes complete in order on one stream
        a, b = lo % self.capacity, (hi - 1) % self.capacity + 1
        if a < b:
            return self.frames[a:b], ev
        with torch.cuda.stream(self.stream):             # wrapped window: one contiguous copy
            out = torch.cat([self.frames[a:], self.frames[:b]])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def release(self, upto: int, after: "torch.cuda.Event | None" = None):
        """Frames below ``upto`` are consumed; ``after`` = event of the last kernel that reads them."""
        with self._cv:
            upto = min(upto, self.head)
            for i in range(self.tail, upto):
                self._ready.pop(i, None)
            self.tail = max(self.tail, upto)
            if after is not None:
                self._released = after
            self._cv.notify_all()

    def close(self):
        with self._cv:
            self.closed = True
            self._cv.notify_all()


# ---- external decoder -> ring --------------------------------------------------------------------------------------------------
def find_decoder() -> str | None:
    """The reference calls ``./ffmpeg/ffmpeg`` (data/utils.py:62); otherwise whatever ``ffmpeg`` is on PATH."""
    local = os.path.join(".", "ffmpeg", "ffmpeg")
    return local if os.path.isfile(local) and os.access(local, os.X_OK) else shutil.which("ffmpeg")


def decoder_command(path: str, fps: float, ffmpeg: str = "ffmpeg") -> list[str]:
    """data/utils.py:62-64 without the `-vf scale=...,pad=...` filter (csrc/ingest.hip does that on the device) and with the frames
    going to stdout as packed RGB24 instead of into a re-encoded file."""
    return [ffmpeg, "-nostdin", "-loglevel", "error", "-i", path, "-r", f"{fps:g}", "-f", "rawvideo", "-pix_fmt", "rgb24", "-"]


def probe_command(path: str, ffprobe: str = "ffprobe") -> list[str]:
    """Prints `width,height` of the first video stream."""
    return [ffprobe, "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=width,height", "-of", "csv=p=0", path]


def read_raw_frames(pipe, height: int, width: int, chunk: int = 8):
    """Yield uint8 [n <= chunk, H, W, 3] host tensors from a binary stream of packed RGB24 frames; a stream that ends inside a frame
    is an error (a crashed decoder must not turn into a silently shorter video)."""
    fb = height * width * 3
    while True:
        buf = bytearray()
        while len(buf) < fb * chunk:
            part = pipe.read(fb * chunk - len(buf))
            if not part:
                break
            buf += part
        if len(buf) % fb:
            raise IOError(f"decoder stream ended inside a frame ({len(buf) % fb} of {fb} bytes)")
        if not buf:
            return
        yield torch.frombuffer(buf, dtype=torch.uint8).view(-1, height, width, 3)
        if len(buf) < fb * chunk:
            return


class DecoderFeed(threading.Thread):
    """Runs ``argv`` (a decoder writing packed RGB24 frames of ``height`` x ``width`` to stdout), pushes its frames into ``ring`` as they
    arrive, waits while the ring is full, closes the ring at end of stream.  ``error`` holds what went wrong, ``join()`` re-raises it."""

    def __init__(self, argv, ring: FrameRing, full_timeout: float = 60.0):
        super().__init__(daemon=True)
        if ring.layout != "THWC":
            raise ValueError("a raw RGB24 stream is [T,H,W,3]: the ring must use layout THWC")
        self.argv, self.ring, self.full_timeout = list(argv), ring, full_timeout
        self.error = None
        self.frames = 0
        ring.feed = self                               # LiveInfer.input_video_stream reports a decoder failure instead of "frame N is not available"
        self.proc = subprocess.Popen(self.argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, bufsize=0)
        self.start()

    def run(self):
        ring = self.ring
        try:
            if ring.frames.is_cuda:
                torch.cuda.set_device(ring.frames.device)
            for frames in read_raw_frames(self.proc.stdout, ring.H, ring.W, ring.chunk):
                if not ring.wait_free(frames.shape[0], self.full_timeout):
                    raise TimeoutError(f"frame ring full for {self.full_timeout:g} s: the consumer stopped releasing frames")
                ring.push(frames)
                self.frames += frames.shape[0]
            rc = self.proc.wait()
            if rc != 0:
                raise RuntimeError(f"decoder exited with {rc}: {self.proc.stderr.read().decode(errors='replace')[-500:]}")
        except BaseException as ex:          # surfaced by join(); the consumer sees a closed ring
            self.error = ex
            if self.proc.poll() is None:
                self.proc.kill()
        finally:
            ring.close()

    def join(self, timeout=None):
        super().join(timeout)
        if self.error is not None:
            raise self.error


def open_video(engine, path: str, fps: float, height: int | None = None, width: int | None = None, capacity: int = 64, chunk: int = 8,
               ffmpeg: str | None = None) -> tuple[FrameRing, DecoderFeed]:
    """``path`` -> (ring, feeder): what demo/cli.py:13-22 + demo/inference.py:111-115 do with a re-encoded file and a resident tensor,
    as a pipe from the decoder into the device ring.  Needs an ffmpeg binary (and ffprobe unless the frame size is given)."""
    ffmpeg = ffmpeg or find_decoder()
    if ffmpeg is None:
        raise RuntimeError("no ffmpeg binary (./ffmpeg/ffmpeg or PATH): push decoded frames into a FrameRing yourself")
    if height is None or width is None:
        probe = os.path.join(os.path.dirname(ffmpeg), "ffprobe") if os.path.dirname(ffmpeg) else "ffprobe"
        out = subprocess.run(probe_command(path, probe), capture_output=True, text=True, check=True).stdout.strip().split(",")
        width, height = int(out[0]), int(out[1])
    ring = FrameRing(engine, height, width, capacity=capacity, chunk=chunk)
    return ring, DecoderFeed(decoder_command(path, fps, ffmpeg), ring)
