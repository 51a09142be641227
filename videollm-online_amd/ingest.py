"""Frame ingest in front of ``LiveInfer``: decoded frames arriving on the HOST (any size, RGB24 as a decoder emits them) become
the uint8 [3,R,R] frames the vision tower consumes, through a pinned-memory -> device ring, so a stream never needs the whole
video resident in HBM.

The reference prepares the file with an external ffmpeg and then loads ALL of it to the GPU (demo/cli.py:13-22 ->
data/utils.py:51-66; demo/inference.py:111-115 `read_video(...).to('cuda')`): 442 KB per frame, 1.6 GB per hour at 2 FPS.
Here a decoder thread / process calls ``ring.push(frames)``; the copy engine uploads the raw frames from pinned staging
buffers on its own HIP stream, `vlo_frame_ingest` (csrc/ingest.hip) scales + pads them into the ring, and
``LiveInfer.input_video_stream`` encodes windows of the ring.  Video DECODING itself (mp4 demux, H.264) is out of reach in
this image — no ffmpeg / torchvision / PyAV / rocDecode — so the ring starts at decoded frames."""
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
        for i in range(n):
            self._ready[first + i] = ev
        self.head += n
        return first

    def window(self, lo: int, hi: int):
        """(uint8 [hi-lo,3,R,R] device tensor, event to wait on) for frames [lo, hi) — all of them pushed and not yet released."""
        if not (self.tail <= lo < hi <= self.head):
            raise IndexError(f"frames [{lo}, {hi}) are not in the ring (holds [{self.tail}, {self.head}))")
        a, b = lo % self.capacity, (hi - 1) % self.capacity + 1
        ev = self._ready[hi - 1]                         # pushes complete in order on one stream
        if a < b:
            return self.frames[a:b], ev
        with torch.cuda.stream(self.stream):             # wrapped window: one contiguous copy
            out = torch.cat([self.frames[a:], self.frames[:b]])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def release(self, upto: int, after: "torch.cuda.Event | None" = None):
        """Frames below ``upto`` are consumed; ``after`` = event of the last kernel that reads them."""
        upto = min(upto, self.head)
        for i in range(self.tail, upto):
            self._ready.pop(i, None)
        self.tail = max(self.tail, upto)
        if after is not None:
            self._released = after

    def close(self):
        self.closed = True
