"""The SOURCE of the peer-to-peer exchange kernels (csrc/tp_p2p.cuh: tp_xchg_norm_kernel, tp_gather_kernel) executed on
the CPU through the HIP-on-threads shim of tests/hip_emul/, against a numpy restatement of what an exchange must produce:

    sum_r ( sum_s partial[r][s] )  in rank order, fp32  ->  h += bf16(sum) (bf16)  ->  x = w * bf16(h * rsqrt(mean(h^2) + eps))

on the mailbox geometry the launch code uses (vlo_debug_p2p_layout).  Checks the index arithmetic (slots, sources, rows,
multi-chunk rows), the tag protocol (stale tags never match, a silent rank times out instead of hanging, the sticky error
word stops later waits) and the rounding points.  It cannot check the GPU memory model — that is what
tests/test_zz_gpu_tp_p2p.py and `bench.py --tp --tp-allreduce p2p` are for."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "hip_emul")


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def emul():
    cc = _clang()
    if cc is None:
        pytest.skip("no clang++ (ext_vector_type / __bf16 on the host) to build the kernel harness")
    out_dir = os.path.join(EMUL, "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libp2p_emul.so")
    # -Bsymbolic: libvlo.so (RTLD_GLOBAL, loaded by other tests of the session) exports host stubs with the kernels' names
    san = os.environ.get("VLO_EMUL_SANITIZE")                  # e.g. address,undefined or thread (tools/emul_sanitize.sh)
    extra = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g"] if san else []
    if san:
        out_dir = os.path.join(out_dir, "san_" + san.replace(",", "_"))
        os.makedirs(out_dir, exist_ok=True)
        lib = os.path.join(out_dir, "libp2p_emul.so")
    cmd = [cc, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-DXCHG_THREADS=64", *extra, "-I", EMUL,
           "-I", os.path.join(ROOT, "videollm-online_amd", "csrc"), os.path.join(EMUL, "p2p_harness.cpp"), "-o", lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    L = C.CDLL(lib, mode=os.RTLD_LOCAL)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.emul_p2p_exchange.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, u64, u64, C.c_uint32, C.c_float, C.c_longlong, i32, vp]
    L.emul_p2p_gather.argtypes = [i32, i32, i32, vp, vp, vp, u64, u64, C.c_uint32, C.c_longlong, i32, vp]
    assert L.emul_xchg_threads() == 64
    return L


def _layout(T, H, Vl, seq, epoch):
    from videollm_online_amd import _C
    out = (C.c_int64 * 6)()
    _C.check(_C.lib().vlo_debug_p2p_layout(T, H, Vl, seq, epoch, out))
    return dict(red_off=out[0], gat_off=out[2], total=out[4])


def to_bf16(x):
    """float32 -> bf16 bits, round to nearest even (torch's .to(bfloat16), csrc/common.cuh f2bf)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def from_bf16(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def rbf(x):
    return from_bf16(to_bf16(x))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _expected(partials, h_bits, w_bits, m, eps, ranks):
    """numpy restatement for the rows < m; `ranks`: the sources whose granules exist (all of them normally)"""
    T, ks, _, H = partials.shape
    total = np.zeros((16, H), dtype=np.float32)
    for r in range(T):                       # rank order; inside a rank the K slabs in order, from 0
        if r not in ranks:
            continue                         # a source that timed out counts as zero
        d = np.zeros((16, H), dtype=np.float32)
        for s in range(ks):
            d = d + partials[r, s]
        total = total + d
    hn = rbf(from_bf16(h_bits) + rbf(total))
    ss = (hn.astype(np.float64) ** 2).sum(-1, keepdims=True)
    rs = (1.0 / np.sqrt(ss / H + eps)).astype(np.float32)
    x = to_bf16(from_bf16(w_bits)[None, :] * rbf(hn * rs))
    return to_bf16(hn)[:m], x[:m]


@pytest.mark.parametrize("T,m,H,ks,seq", [(2, 1, 64, 1, 0), (2, 11, 512, 1, 1), (4, 16, 1024, 3, 0), (8, 3, 1024, 4, 1), (8, 11, 2048, 1, 0)])
def test_exchange_kernel_source_matches_numpy(emul, T, m, H, ks, seq):
    rng = np.random.default_rng(1000 * T + m + H + ks)
    Vl = 64
    lay = _layout(T, H, Vl, seq, 41)
    epoch = 42
    partials = (rng.standard_normal((T, ks, 16, H)) * 0.5).astype(np.float32)
    h0 = to_bf16(rng.standard_normal((16, H)).astype(np.float32))
    w = to_bf16((1 + 0.1 * rng.standard_normal(H)).astype(np.float32))
    h = np.repeat(h0[None], T, 0).copy()
    x = np.full((T, 16, H), 0xBEEF, dtype=np.uint16)
    # a used mailbox: every granule carries an OLD tag (epoch - 2: the previous occupant of this slot) and junk data
    mbox = np.full((T, lay["total"]), ((epoch - 2) << 32) | 0x7FC00000, dtype=np.uint64)
    err = np.zeros(T, dtype=np.uint32)
    rc = emul.emul_p2p_exchange(T, m, H, ks, _ptr(partials), _ptr(h), _ptr(w), _ptr(x), _ptr(mbox), lay["total"], lay["red_off"],
                                epoch, 1e-5, 50_000_000, -1, _ptr(err))
    assert rc == 0 and not err.any()
    eh, ex = _expected(partials, h0, w, m, 1e-5, set(range(T)))
    for r in range(T):
        assert np.array_equal(h[r, :m], eh), f"rank {r}: residual stream differs"
        assert np.array_equal(h[r, m:], h0[m:]) and (x[r, m:] == 0xBEEF).all(), "rows >= m must not be touched"
        ulp = np.abs(x[r, :m].astype(np.int32) - ex.astype(np.int32))
        assert ulp.max() <= 1 and (ulp != 0).mean() < 0.01, f"rank {r}: normed rows differ by more than a bf16 ulp"
    assert all(np.array_equal(x[0], x[r]) for r in range(T)), "every rank must hold the same bits"
    # only this slot's first m rows of every source were written, with this epoch's tag; everything else keeps the old tag
    tags = (mbox >> np.uint64(32)).astype(np.int64)
    written = np.zeros_like(tags, dtype=bool)
    for src in range(T):
        base = lay["red_off"] + src * 16 * H
        written[:, base:base + m * H] = True
    assert (tags[written] == epoch).all() and (tags[~written] == epoch - 2).all()


def test_silent_rank_times_out_and_error_word_is_sticky(emul):
    T, m, H, ks = 4, 5, 256, 2
    rng = np.random.default_rng(7)
    lay = _layout(T, H, 64, 0, 8)
    partials = rng.standard_normal((T, ks, 16, H)).astype(np.float32)
    h0 = to_bf16(rng.standard_normal((16, H)).astype(np.float32))
    w = to_bf16(np.ones(H, dtype=np.float32))
    h = np.repeat(h0[None], T, 0).copy()
    x = np.zeros((T, 16, H), dtype=np.uint16)
    mbox = np.zeros((T, lay["total"]), dtype=np.uint64)
    err = np.zeros(T, dtype=np.uint32)
    # rank 2 never publishes: 2 ms timeout (200 000 ticks of the 100 MHz counter), every rank gives up and flags it
    rc = emul.emul_p2p_exchange(T, m, H, ks, _ptr(partials), _ptr(h), _ptr(w), _ptr(x), _ptr(mbox), lay["total"], lay["red_off"],
                                9, 1e-5, 200_000, 2, _ptr(err))
    assert rc != 0 and err.all()
    # results after a timeout are invalid by contract; what IS guaranteed, per 8-column chunk (one thread's unit of work): a
    # thread that waited adds the sources that arrived (the missing one counts as zero), a thread that started after the
    # sticky word went up does not wait and adds nothing
    eh, _ = _expected(partials, h0, w, m, 1e-5, {0, 1, 3})
    waited = 0
    for r in range(T):
        got, want, before = (a.reshape(-1, 8) for a in (h[r, :m], eh, h0[:m]))
        as_sum, untouched = (got == want).all(-1), (got == before).all(-1)
        assert (as_sum | untouched).all()
        waited += int(as_sum.sum())
    assert waited > 0, "somebody must have waited for the timeout"
    # with the sticky word set nothing waits any more: the next exchange returns at once and adds nothing
    h2 = h.copy()
    rc = emul.emul_p2p_exchange(T, m, H, ks, _ptr(partials), _ptr(h2), _ptr(w), _ptr(x), _ptr(mbox), lay["total"], lay["red_off"],
                                11, 1e-5, 10**12, 2, _ptr(err))
    assert rc == 0                                                  # no NEW timeout was raised
    assert np.array_equal(h2, h), "a dead exchange must leave the residual stream alone"


@pytest.mark.parametrize("T,nr,Vl,blocks,seq", [(2, 1, 128, 2, 0), (4, 3, 250, 3, 1), (8, 16, 64, 1, 0)])
def test_gather_kernel_source(emul, T, nr, Vl, blocks, seq):
    rng = np.random.default_rng(T * 100 + nr)
    H = 64
    lay = _layout(T, H, Vl, seq, 99)
    local = rng.integers(0, 1 << 16, size=(T, nr, Vl), dtype=np.uint16)
    out = np.zeros((T, nr, T * Vl), dtype=np.uint16)
    mbox = np.zeros((T, lay["total"]), dtype=np.uint64)
    err = np.zeros(T, dtype=np.uint32)
    rc = emul.emul_p2p_gather(T, nr, Vl, _ptr(local), _ptr(out), _ptr(mbox), lay["total"], lay["gat_off"], 100, 50_000_000, blocks,
                              _ptr(err))
    assert rc == 0 and not err.any()
    want = np.concatenate([local[r] for r in range(T)], axis=-1)    # [nr][V]: rank r's shard at columns r*Vl ..
    for r in range(T):
        assert np.array_equal(out[r], want)
    assert (mbox[:, :lay["gat_off"]] == 0).all() if seq == 0 else True     # the reduce region is not touched


def test_shim_cooperative_launch_runs_blocks_concurrently():
    """hipLaunchCooperativeKernel in the shim: one forked process per block over shared "device" memory — a hand-rolled grid
    barrier completes and every block reads what its neighbour published in the same round (tests/hip_emul/coop_selftest.cpp)."""
    cc = _clang()
    if cc is None:
        pytest.skip("no clang++")
    out_dir = os.path.join(EMUL, "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libcoop_selftest.so")
    r = subprocess.run([cc, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I", EMUL,
                        os.path.join(EMUL, "coop_selftest.cpp"), "-lrt", "-o", lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    L = C.CDLL(lib, mode=os.RTLD_LOCAL)
    blocks, threads, rounds = 6, 64, 5
    out = np.zeros(rounds * blocks, dtype=np.uint32)
    assert L.coop_selftest(blocks, threads, rounds, _ptr(out)) == 0
    want = np.array([[r * 1000 + (b + 1) % blocks for b in range(blocks)] for r in range(rounds)], dtype=np.uint32)
    assert np.array_equal(out.reshape(rounds, blocks), want)
