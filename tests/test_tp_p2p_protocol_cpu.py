"""Host-side checks of the one-shot peer-to-peer exchange (csrc/tp.hip "p2p exchange", include/vlo.h vlo_tp_p2p_*):

* the mailbox geometry the launch code uses (vlo_debug_p2p_layout — the same helpers) keeps every (region, slot, source,
  row) range disjoint and inside the mailbox, alternates slots per region and never produces the tag 0;
* a threaded model of the protocol on that geometry — T ranks in lock-step program order but with arbitrary skew, every
  rank publishing {tag, value} granules into all mailboxes and collecting its own by tag equality — always collects
  exactly what the sources published for THAT exchange: two slots per region are enough, also when reduce and gather
  exchanges interleave the way a step issues them (2 per layer, final norm, logits gather; chunks without logits).

The kernels themselves need a GPU (tests/test_zz_gpu_tp_p2p.py); this pins the arithmetic and the slot-reuse argument."""
import ctypes as C
import random
import threading
import time

import numpy as np
import pytest


def _layout(T, H, Vl, seq, epoch):
    from videollm_online_amd import _C
    out = (C.c_int64 * 6)()
    _C.check(_C.lib().vlo_debug_p2p_layout(T, H, Vl, seq, epoch, out))
    return dict(red_off=out[0], red_src=out[1], gat_off=out[2], gat_src=out[3], total=out[4], next_epoch=out[5])


@pytest.mark.parametrize("T,H,Vl", [(2, 128, 256), (4, 2048, 8000), (8, 4096, 16032), (8, 8192, 16032)])
def test_mailbox_regions_are_disjoint(T, H, Vl):
    ranges = []
    for seq in (0, 1):
        L = _layout(T, H, Vl, seq, 5)
        assert L["red_src"] == 16 * H and L["gat_src"] == 16 * (Vl // 2)
        for src in range(T):
            ranges.append((L["red_off"] + src * L["red_src"], L["red_off"] + (src + 1) * L["red_src"]))
            ranges.append((L["gat_off"] + src * L["gat_src"], L["gat_off"] + (src + 1) * L["gat_src"]))
        total = L["total"]
    ranges.sort()
    assert ranges[0][0] == 0 and ranges[-1][1] == total            # the regions tile the mailbox exactly
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0
    # slots alternate with the region's own sequence number, whatever the tag is
    assert _layout(T, H, Vl, 0, 1)["red_off"] == _layout(T, H, Vl, 2, 77)["red_off"] != _layout(T, H, Vl, 1, 1)["red_off"]
    assert _layout(T, H, Vl, 0, 1)["gat_off"] == _layout(T, H, Vl, 2, 77)["gat_off"] != _layout(T, H, Vl, 1, 1)["gat_off"]


def test_epoch_is_never_zero_and_bad_arguments_fail():
    from videollm_online_amd import _C
    assert _layout(2, 128, 256, 0, 0)["next_epoch"] == 1
    assert _layout(2, 128, 256, 0, 41)["next_epoch"] == 42
    assert _layout(2, 128, 256, 0, 0xFFFFFFFF)["next_epoch"] == 1      # wraps past the "empty mailbox" tag
    out = (C.c_int64 * 6)()
    for bad in [(1, 128, 256), (9, 128, 256), (2, 100, 256), (2, 128, 255)]:
        assert _C.lib().vlo_debug_p2p_layout(*bad, 0, 1, out) != 0


def _step_script(layers, with_logits):
    """region of every exchange a tp_chunk issues: 'R' x2 per layer (the second of the last layer only with logits), 'G'."""
    ops = []
    for l in range(layers):
        ops.append("R")
        if l + 1 < layers or with_logits:
            ops.append("R")
    if with_logits:
        ops.append("G")
    return ops


@pytest.mark.parametrize("T,seed", [(2, 0), (4, 1), (8, 2)])
def test_threaded_model_of_the_exchange(T, seed):
    H, Vl, rows = 8, 8, 2                      # tiny rows: the protocol does not depend on the sizes
    script = []
    for chunk in range(6):                     # multi-chunk steps (no logits) mixed with ordinary steps
        script += _step_script(3, with_logits=chunk % 3 != 1)
    total = _layout(T, H, Vl, 0, 1)["total"]
    mbox = [np.zeros(total, dtype=np.uint64) for _ in range(T)]
    errors = []
    deadline = time.time() + 60

    def value(epoch, src, row, col):
        return (epoch * 1315423911 + src * 2654435761 + row * 97 + col) & 0xFFFFFFFF

    def rank_main(me):
        rng = random.Random(seed * 100 + me)
        epoch, n = 0, {"R": 0, "G": 0}
        try:
            for op in script:
                L = _layout(T, H, Vl, n[op], epoch)
                epoch = L["next_epoch"]
                n[op] += 1
                off, stride, width = (L["red_off"], L["red_src"], H) if op == "R" else (L["gat_off"], L["gat_src"], Vl // 2)
                if rng.random() < 0.3:
                    time.sleep(rng.random() * 0.004)           # skew between the ranks
                for p in rng.sample(range(T), T):               # publish into every mailbox, any order
                    for row in range(rows):
                        base = off + me * stride + row * width
                        for col in range(width):
                            mbox[p][base + col] = np.uint64((epoch << 32) | value(epoch, me, row, col))
                for src in range(T):                            # collect by tag equality
                    for row in range(rows):
                        base = off + src * stride + row * width
                        for col in range(width):
                            while True:
                                g = int(mbox[me][base + col])
                                if g >> 32 == epoch:
                                    break
                                if time.time() > deadline:
                                    raise TimeoutError(f"rank {me} exchange {epoch}: granule of rank {src} never arrived")
                                time.sleep(0)
                            if g & 0xFFFFFFFF != value(epoch, src, row, col):
                                raise AssertionError(f"rank {me} exchange {epoch}: wrong payload from rank {src}")
        except Exception as ex:                                # noqa: BLE001 — reported by the main thread
            errors.append(ex)

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(T)]
    [t.start() for t in ts]
    [t.join(90) for t in ts]
    assert not any(t.is_alive() for t in ts), "model deadlocked"
    assert not errors, errors


def test_tpgroup_argument_validation_needs_no_gpu():
    """The exchange choice is validated before any engine (GPU) is touched."""
    from videollm_online_amd.engine import EngineConfig, TpGroup
    cfg = EngineConfig(64, 128, 1, 4, 2, 128)
    with pytest.raises(ValueError, match="allreduce"):
        TpGroup(cfg, 2, allreduce="ring")
    with pytest.raises(ValueError, match="handle_allgather"):
        TpGroup(cfg, 2, rank=0, allreduce="p2p")
    with pytest.raises(ValueError, match="unique id"):
        TpGroup(cfg, 2, rank=0)


def _handles_worker(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = bytes([rank + 1]) * 64               # stands for this rank's hipIpc mailbox handle
    out = [None] * world                        # bench.py --tp --tp-allreduce p2p: gather_handles()
    dist.all_gather_object(out, mine)
    q.put((rank, out))
    dist.destroy_process_group()


def test_handle_allgather_is_in_rank_order_gloo():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_handles_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    want = [bytes([1]) * 64, bytes([2]) * 64]
    assert res[0] == want and res[1] == want
