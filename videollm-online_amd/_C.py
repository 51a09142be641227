"""ctypes binding of libvlo.so (include/vlo.h).  Fails loudly when the library is missing:
there is no Python/CPU fallback for any entry point."""
import ctypes as C
import os

import torch  # noqa: F401  — loads torch's bundled libamdhip64 first so libvlo.so binds to the same HIP runtime

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvlo.so")

VLO_ABI_VERSION = 3
DT_F32, DT_BF16, DT_F16, DT_FP8_E4M3 = 0, 1, 2, 3


class VloConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("vocab_size", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float),
        ("vision_hidden_size", C.c_int32), ("frame_num_tokens", C.c_int32),
        ("has_vit", C.c_int32), ("vit_hidden_size", C.c_int32), ("vit_intermediate_size", C.c_int32),
        ("vit_num_layers", C.c_int32), ("vit_num_heads", C.c_int32), ("vit_image_size", C.c_int32),
        ("vit_patch_size", C.c_int32), ("vit_ln_eps", C.c_float), ("pool_h", C.c_int32), ("pool_w", C.c_int32),
        ("kv_pool_tokens", C.c_int64), ("tp_rank", C.c_int32), ("tp_size", C.c_int32), ("weight_dtype", C.c_int32),
        ("prefill_act_dtype", C.c_int32),
    ]


_lib = None

EXPORTS = [
    "vlo_abi_version", "vlo_last_error", "vlo_engine_create", "vlo_engine_load_weight", "vlo_engine_finalize",
    "vlo_engine_destroy", "vlo_engine_weight_bytes", "vlo_session_create", "vlo_session_reset", "vlo_session_len",
    "vlo_session_destroy", "vlo_visual_embed", "vlo_vision_tokens", "vlo_connector", "vlo_embed", "vlo_llm_step", "vlo_stream_sample",
    "vlo_greedy_generate", "vlo_session_read_kv", "vlo_step_algorithmic_bytes", "vlo_test_gemv",
    "vlo_profile_enable", "vlo_profile_read", "vlo_bench_gemv", "vlo_debug_read", "vlo_profile_calibrate", "vlo_debug_gemv_plan",
    "vlo_tp_unique_id", "vlo_tp_group_create", "vlo_tp_group_destroy", "vlo_tp_session_create", "vlo_tp_session_reset",
    "vlo_tp_session_len", "vlo_tp_session_destroy", "vlo_tp_llm_step", "vlo_tp_stream_sample", "vlo_tp_greedy_generate",
    "vlo_joint_embed", "vlo_logit_rows", "vlo_session_fork", "vlo_session_crop", "vlo_tp_selftest", "vlo_debug_gemm64_plan", "vlo_debug_pack64_elem",
    "vlo_step_input", "vlo_build_id", "vlo_frame_ingest", "vlo_frame_ingest_geometry", "vlo_test_gemv_fp8", "vlo_test_gemm_fp8", "vlo_tp_comm_info", "vlo_tp_allgather",
    "vlo_tp_p2p_export", "vlo_tp_p2p_enable", "vlo_tp_p2p_status", "vlo_debug_p2p_layout", "vlo_tp_bench_exchange",
    "vlo_tp_session_fork", "vlo_tp_session_crop",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:                                   # same HIP sources, built on the spot — not a fallback path
            from .build import build
            build()
        except Exception as ex:
            raise RuntimeError(f"{LIB_PATH} not found and could not be built ({ex}).  Build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950).  "
                               "The engine has no non-HIP path.") from ex
    from .build import library_build_id, source_hash
    want, got = source_hash(), library_build_id(LIB_PATH)
    if want is not None and got != want and os.environ.get("VLO_ALLOW_STALE_LIB") != "1":
        raise RuntimeError(f"{LIB_PATH} was built from other sources (library {str(got)[:12]}, tree {want[:12]}): rebuild it "
                           "(`python -c 'import __graft_entry__ as g; g.build()'`) or set VLO_ALLOW_STALE_LIB=1")
    L = bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    _lib = L
    return L


def bind(L):
    """Declare the argument / return types of every include/vlo.h entry point on a loaded library and check its ABI version."""
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.vlo_abi_version.restype = i32
    L.vlo_last_error.restype = C.c_char_p
    L.vlo_engine_create.argtypes = [C.POINTER(VloConfig), i32, C.POINTER(vp)]
    L.vlo_engine_load_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
    L.vlo_engine_finalize.argtypes = [vp]
    L.vlo_engine_destroy.argtypes = [vp]
    L.vlo_engine_destroy.restype = None
    L.vlo_engine_weight_bytes.argtypes = [vp]
    L.vlo_engine_weight_bytes.restype = i64
    L.vlo_session_create.argtypes = [vp, i64, C.POINTER(vp)]
    L.vlo_session_reset.argtypes = [vp]
    L.vlo_session_len.argtypes = [vp]
    L.vlo_session_len.restype = i64
    L.vlo_session_destroy.argtypes = [vp]
    L.vlo_session_destroy.restype = None
    L.vlo_visual_embed.argtypes = [vp, vp, i32, vp, vp]
    L.vlo_joint_embed.argtypes = [vp, vp, i32, i64, vp, i32, vp, vp]
    L.vlo_logit_rows.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.vlo_session_fork.argtypes = [vp, i64, C.POINTER(vp), vp]
    L.vlo_session_crop.argtypes = [vp, i64]
    L.vlo_vision_tokens.argtypes = [vp, vp, i32, vp, vp]
    L.vlo_connector.argtypes = [vp, vp, i32, vp, vp]
    L.vlo_embed.argtypes = [vp, vp, i32, vp, vp]
    L.vlo_step_input.argtypes = [vp, C.POINTER(i64), i32, vp, i32, vp, vp]
    L.vlo_build_id.restype = C.c_char_p
    L.vlo_frame_ingest.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_float, vp, vp]
    L.vlo_frame_ingest_geometry.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.vlo_llm_step.argtypes = [vp, vp, i32, vp, vp, vp]
    L.vlo_stream_sample.argtypes = [vp, C.c_float, i32, vp, vp, vp]
    L.vlo_greedy_generate.argtypes = [vp, vp, i32, i32, vp, i32, i32, C.POINTER(i32), vp]
    L.vlo_session_read_kv.argtypes = [vp, i32, i32, i32, i64, i64, vp, vp]
    L.vlo_step_algorithmic_bytes.argtypes = [vp, i64, i32]
    L.vlo_step_algorithmic_bytes.restype = C.c_double
    L.vlo_test_gemv.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.vlo_test_gemv_fp8.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.vlo_test_gemm_fp8.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(C.c_double), vp]
    L.vlo_bench_gemv.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]
    L.vlo_debug_read.argtypes = [vp, i32, vp, i64, vp]
    L.vlo_debug_gemv_plan.argtypes = [i32, i32, C.POINTER(i32)]
    L.vlo_profile_calibrate.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.vlo_tp_unique_id.argtypes = [vp]
    L.vlo_tp_selftest.argtypes = [i32]
    L.vlo_debug_gemm64_plan.argtypes = [i32, C.POINTER(i32)]
    L.vlo_debug_pack64_elem.argtypes = [i32, i32]
    L.vlo_debug_pack64_elem.restype = i64
    L.vlo_tp_group_create.argtypes = [C.POINTER(vp), i32, vp, C.POINTER(vp)]
    L.vlo_tp_group_destroy.argtypes = [vp]
    L.vlo_tp_group_destroy.restype = None
    L.vlo_tp_session_create.argtypes = [vp, i64, C.POINTER(vp)]
    L.vlo_tp_session_reset.argtypes = [vp]
    L.vlo_tp_session_len.argtypes = [vp]
    L.vlo_tp_session_len.restype = i64
    L.vlo_tp_session_destroy.argtypes = [vp]
    L.vlo_tp_session_destroy.restype = None
    L.vlo_tp_session_fork.argtypes = [vp, i64, C.POINTER(vp), vp]
    L.vlo_tp_session_crop.argtypes = [vp, i64]
    L.vlo_tp_llm_step.argtypes = [vp, vp, i32, vp, vp, vp]
    L.vlo_tp_stream_sample.argtypes = [vp, C.c_float, i32, vp, vp, vp]
    L.vlo_tp_greedy_generate.argtypes = [vp, vp, i32, i32, vp, i32, i32, C.POINTER(i32), vp]
    L.vlo_tp_p2p_export.argtypes = [vp, vp]
    L.vlo_tp_p2p_enable.argtypes = [vp, vp]
    L.vlo_tp_p2p_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.vlo_tp_bench_exchange.argtypes = [vp, i32, i32, C.POINTER(C.c_double), vp]
    L.vlo_tp_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.vlo_tp_allgather.argtypes = [vp, vp, vp, i64, vp]
    L.vlo_debug_p2p_layout.argtypes = [i32, i32, i32, C.c_uint32, C.c_uint32, C.POINTER(i64)]
    L.vlo_profile_enable.argtypes = [vp, i32]
    L.vlo_profile_read.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if L.vlo_abi_version() != VLO_ABI_VERSION:
        raise RuntimeError("libvlo.so ABI version mismatch — rebuild")
    return L


def check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libvlo error {rc}: {lib().vlo_last_error().decode()}")
