"""CPU-side checks of the drop-in boundary: libvlo.so loads and exports every symbol
include/vlo.h declares (no compute calls — no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vlo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vlo_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as G
    G.build()
    from videollm_online_amd import _C
    L = _C.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"libvlo.so does not export {n}"
    assert set(names) == set(_C.EXPORTS), set(names) ^ set(_C.EXPORTS)
    assert L.vlo_abi_version() == _C.VLO_ABI_VERSION


def test_error_convention_without_gpu():
    from videollm_online_amd import _C
    L = _C.lib()
    h = ctypes.c_void_p()
    cfg = _C.VloConfig()
    cfg.abi_version = 999
    rc = L.vlo_engine_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc == -1 and b"abi_version" in L.vlo_last_error()
    assert L.vlo_session_len(None) == -1


def test_product_path_has_no_oracle_or_cpu_fallback():
    pkg = os.path.join(ROOT, "videollm-online_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                s = open(os.path.join(dp, f)).read()
                assert "oracle" not in s.replace("no oracle", ""), f"{f} references the oracle"
                # the CPU emulation of the kernels (tests/hip_emul/) is test infrastructure: the product never builds, loads or mentions it
                assert "hip_emul" not in s and "libvlo_emul" not in s and "VLO_HIP_EMUL" not in s, f"{f} references the test-only emulation"


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from videollm_online_amd.engine import Engine, EngineConfig
    with pytest.raises(RuntimeError, match="no CPU path"):
        Engine(EngineConfig(64, 128, 1, 4, 2, 128))


def test_gemv_planner_covers_every_model_shape():
    """Host-side planner (no GPU): every reduction length of the supported models — incl. the tensor-parallel shards —
    decomposes as K/32 = waves x fragments x chunks x slices, and whole-K plans never split across blocks."""
    import ctypes as C
    from videollm_online_amd import _C
    L = _C.lib()
    out = (C.c_int * 4)()
    shapes = {  # K : where it occurs
        4096: "8B hidden", 14336: "8B intermediate", 2048: "TinyLlama hidden / 8B o_proj TP=2", 5632: "TinyLlama intermediate",
        8192: "70B hidden", 28672: "70B intermediate", 1024: "connector / SigLIP", 7168: "8B I/2", 3584: "8B I/4 / 70B I/8",
        1792: "8B I/8", 512: "8B o_proj TP=8", 1408: "TinyLlama I/4", 256: "toy", 704: "toy", 128: "toy vision"}
    for K, where in shapes.items():
        for allow in (0, 1):
            assert L.vlo_debug_gemv_plan(K, allow, out) == 0, (K, where)
            nw, kf, kc, ks = list(out)
            assert nw * kf * kc * ks * 32 == K, (K, where, list(out))
            assert nw in (1, 2, 4, 8) and 1 <= kf <= 16
            if not allow:
                assert ks == 1
    assert L.vlo_debug_gemv_plan(100, 0, out) < 0 and b"no GEMV plan" in L.vlo_last_error()    # K % 32 != 0


def test_block_path_geometry():
    """Host-side geometry of the 64-token block path (csrc/prefill.hip), no GPU: the K split of every model shape, and
    the packed-64 activation layout — a bijection onto [0, 64 K) in which the 8 consecutive k of one (row, k/8) stay a
    contiguous 16-byte unit and the 64 units of one MFMA B fragment (16 rows x 4 k-groups) are contiguous."""
    import ctypes as C
    import numpy as np
    from videollm_online_amd import _C
    L = _C.lib()
    out = (C.c_int * 3)()
    for K in (4096, 14336, 2048, 5632, 8192, 28672, 1024, 7168, 3584, 1792, 512, 1408, 256, 704, 128):
        assert L.vlo_debug_gemm64_plan(K, out) == 0, K
        nw, kf, kc = list(out)
        assert nw * kf * kc * 32 == K and nw in (1, 2, 4, 8) and kf in (1, 2, 4, 8), (K, list(out))
    assert L.vlo_debug_gemm64_plan(100, out) < 0
    K = 256
    idx = np.array([[L.vlo_debug_pack64_elem(r, k) for k in range(K)] for r in range(64)])
    assert sorted(idx.ravel().tolist()) == list(range(64 * K))
    assert (idx[:, 1:][:, np.arange(K - 1) % 8 != 7] - idx[:, :-1][:, np.arange(K - 1) % 8 != 7] == 1).all()
    for kf in range(K // 32):
        for mt in range(4):
            frag = idx[mt * 16:(mt + 1) * 16, kf * 32:(kf + 1) * 32]
            lo = (kf * 4 + mt) * 64 * 8
            assert frag.min() == lo and frag.max() == lo + 64 * 8 - 1
            # lane = (k % 32 / 8) * 16 + row % 16 owns unit `lane` of the fragment
            for r in (0, 5, 15):
                for q in range(4):
                    assert idx[mt * 16 + r, kf * 32 + q * 8] == lo + (q * 16 + r) * 8
    assert L.vlo_debug_pack64_elem(64, 0) == -1 and L.vlo_debug_pack64_elem(0, -1) == -1
