"""ctypes binding of libcslam_hip.so (the C ABI declared in include/cslam_hip.h; include/cslam_hip_experimental.h for the
A/B partners and diagnostics).

There is no CPU fallback: every product entry point raises CslamHipError when the
library is missing or no MI355X is visible.  Build with `python __graft_entry__.py`
(or `make -C cslam_amd/csrc`).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSLAM_HIP_LIB: another build of the same library (measurement builds: `make -C cslam_amd/csrc abl` = the timing-only
# ablation switches compiled in, libcslam_hip_abl.so; never the product default)
LIB_PATH = os.environ.get("CSLAM_HIP_LIB") or os.path.join(_HERE, "libcslam_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

F32, F64 = 0, 1
MODE_AUTO, MODE_SCAN, MODE_MFMA = 0, 1, 2


class CslamHipError(RuntimeError):
    pass


class CslamGraphError(CslamHipError):
    """CSLAM_E_GRAPH: the graph admits no Fiedler pair (not connected, singular junction Laplacian, TraceMIN breakdown) --
    the condition under which the reference's networkx call raises and acm.py:436-466 re-draws its start point.  Not a
    failure of the GPU path."""


class CslamUnsupportedError(CslamHipError):
    """CSLAM_E_UNSUPPORTED: an optional run-time dependency (RCCL, rocBLAS / rocSOLVER) was not found on this host."""


class CslamLimitError(CslamHipError):
    """CSLAM_E_LIMIT: valid input beyond a size limit of this entry point (cslam_fiedler: junctions of the dense factor)."""


_ERROR_CLASSES = {-5: CslamUnsupportedError, -6: CslamGraphError, -7: CslamLimitError}

_lib = None

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    "cslam_last_error": (C.c_char_p, []),
    "cslam_version": (_i, []),
    "cslam_device_count": (_i, [C.POINTER(_i)]),
    "cslam_device_info": (_i, [_i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i)]),
    "cslam_bank_create": (_i, [_i, _i, _i64, C.POINTER(_vp)]),
    "cslam_bank_destroy": (_i, [_vp]),
    "cslam_bank_size": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i)]),
    "cslam_bank_clear": (_i, [_vp]),
    "cslam_bank_add_host": (_i, [_vp, _vp, _i, _i64]),
    "cslam_bank_add_dev": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "cslam_bank_read_host": (_i, [_vp, _i64, _i64, _vp]),
    "cslam_bank_device_ptr": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "cslam_bank_search_host": (_i, [_vp, _vp, _i, _i64, _i, _vp, _i, _vp, _vp, _vp]),
    "cslam_bank_search_dev": (_i, [_vp, _vp, _i, _i64, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "cslam_bank_search_multi_dev": (_i, [_vp, _i, _vp, _i, _i64, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "cslam_bank_search_enqueue_dev": (_i, [_vp, _vp, _i, _i64, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "cslam_bank_search_finish": (_i, [_vp, C.POINTER(_i64)]),
    "cslam_bank_search_flag_copy_dev": (_i, [_vp, _vp, _vp]),
    "cslam_bank_search_multi_enqueue_dev": (_i, [_vp, _i, _vp, _i, _i64, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "cslam_bank_search_multi_finish": (_i, [_vp, _i, C.POINTER(_i64)]),
    "cslam_bank_last_stats": (_i, [_vp, C.POINTER(_i64 * 4)]),
    "cslam_bank_last_stage": (_i, [_vp, C.POINTER(C.c_int32 * 4)]),
    "cslam_bank_last_kernel_ms": (_i, [_vp, C.POINTER(_f)]),
    "cslam_topk_merge_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "cslam_l2_normalize_dev": (_i, [_vp, _i64, _i, _i64, _f, _i, _vp]),
    "cslam_vlad_aggregate_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp]),
    "cslam_vlad_aggregate_nhwc_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp]),
    "cslam_gem_fc_head_dev": (_i, [_vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "cslam_gem_fc_head_nhwc_dev": (_i, [_vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "cslam_pca_project_dev": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "cslam_pca_project_pairs_dev": (_i, [_vp, _i64, C.c_float, _vp, C.c_float, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "cslam_preprocess_dev": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_f * 3), C.POINTER(_f * 3), _vp, _vp]),
    "cslam_preprocess_nhwc_dev": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_f * 3), C.POINTER(_f * 3), _vp, _vp]),
    "cslam_mac_grad_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "cslam_csr_spmm_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "cslam_csr_spmm4_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "cslam_block4_gram_dev": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "cslam_block4_affine_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "cslam_block4_residual_dev": (_i, [_vp, _vp, _i64, _vp, C.c_double, _vp, _vp, _vp]),
    "cslam_block4_gram_sync": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "cslam_block4_affine_host": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "cslam_block4_residual_sync": (_i, [_vp, _vp, _i64, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "cslam_chain_forward_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cslam_chain_backward_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "cslam_chol_solve4_dev": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "cslam_fiedler": (_i, [_i64, _vp, _vp, _vp, _vp, C.c_uint32, C.c_double, _i, _vp, _vp, _vp, _vp]),
    "cslam_fiedler_start_block": (_i, [C.c_uint32, _i64, _vp]),
    "cslam_fiedler_release": (_i, []),
    "cslam_mac_fw_subset": (_i, [_i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "cslam_scbank_create": (_i, [_i, _i, _i, _i64, _vp]),
    "cslam_scbank_destroy": (_i, [_vp]),
    "cslam_scbank_size": (_i, [_vp, _vp, _vp, _vp]),
    "cslam_scbank_clear": (_i, [_vp]),
    "cslam_scbank_add_host": (_i, [_vp, _vp, _i64]),
    "cslam_scbank_add_dev": (_i, [_vp, _vp, _i64, _vp]),
    "cslam_scbank_read_host": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "cslam_scbank_search_host": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cslam_scbank_search_dev": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cslam_bias_act_pool_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_conv3x3_c3_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino_input_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino_output_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino4_input_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino4_output_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_absmax_dev": (_i, [_vp, _i64, _vp, _vp]),
    "cslam_wino4_input_h3_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "cslam_wino4_output_scaled_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_wino2_fused64_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino2_fused_c64_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_wino4_fused_c64_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cslam_scancontext_from_cloud_dev": (_i, [_vp, _vp, _i, _i, _i, C.c_double, _vp, _vp, _vp]),
    "cslam_wino4_fused_c64_h_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_h_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_r_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_r2_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv_stem_direct_h_dev": (_i, [_vp, _vp, _vp, _vp, C.c_float, _vp, _vp, C.c_float, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "cslam_wino4_stem_c64_h_dev": (_i, [_vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_debug_wfh_prof_dev": (_i, [_vp]),
    "cslam_conv3x3_c3_amax_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "cslam_wino4_input_h2_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "cslam_wino_gemm_h2_dev": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "cslam_wino_zgemm_h2_dev": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "cslam_wino4_output_z_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv_igemm_h2_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv_stem_pool_igemm_h2_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "cslam_conv_igemm_h2p_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float,
                                      C.c_float, C.c_float, _vp, _i, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_r_pairs_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_hp_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp]),
    "cslam_conv3x3_direct_p_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_float, C.c_float, C.c_float,
                                        _vp, _i, _vp, _vp, _vp]),
    "cslam_comm_unique_id": (_i, [_vp]),
    "cslam_comm_init": (_i, [_i, _i, _vp, _i, C.POINTER(_vp)]),
    "cslam_comm_destroy": (_i, [_vp]),
    "cslam_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "cslam_allgather_queries_dev": (_i, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "cslam_exchange_lists_dev": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "cslam_peak_copy_dev": (_i, [_vp, _vp, _i64, _i, _vp]),
    "cslam_peak_mfma_dev": (_i, [_i, _i, _i, _i, _vp, C.POINTER(C.c_double), _vp]),
    "cslam_debug_last_candidates": (_i, [_vp, _i64, C.POINTER(_i), _vp, _vp, C.POINTER(C.c_double)]),
    "cslam_ring_schedule_describe": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int32 * 6), _vp, _i64, _vp, _vp, _vp]),
    "cslam_trunk_timing": (_i, [_i]),
    "cslam_trunk_timing_read": (_i, [C.POINTER(C.c_double * 8)]),
}

# declared in include/cslam_hip_experimental.h (A/B partners, profiling hooks, peak micro-benchmarks), not in the stable ABI
EXPERIMENTAL_SYMBOLS = ("cslam_wino4_input_h3_dev", "cslam_wino2_fused64_dev", "cslam_wino2_fused_c64_dev",
                        "cslam_wino4_fused_c64_dev", "cslam_debug_wfh_prof_dev", "cslam_peak_copy_dev", "cslam_peak_mfma_dev",
                        "cslam_debug_last_candidates", "cslam_ring_schedule_describe", "cslam_trunk_timing", "cslam_trunk_timing_read")
EXPORTED_SYMBOLS = tuple(n for n in _SIGNATURES if n not in EXPERIMENTAL_SYMBOLS)


def build(verbose=False):
    """Compile libcslam_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC_DIR, "-j", "4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise CslamHipError("building libcslam_hip.so failed")
    return LIB_PATH


def load():
    """dlopen the library and bind every symbol of include/cslam_hip.h (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so ...).  Import torch
    # first so that this library and torch share ONE HIP/HSA runtime in the process; loading
    # /opt/rocm's copy first and torch's afterwards leaves the process without a visible GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise CslamHipError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python __graft_entry__.py` or `make -C cslam_amd/csrc`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().cslam_last_error()
        raise _ERROR_CLASSES.get(rc, CslamHipError)(f"libcslam_hip error {rc}: {msg.decode() if msg else ''}")


def require_gpu():
    """Fail loudly when no HIP device is visible (the product path never runs on the CPU)."""
    lib = load()
    n = C.c_int(0)
    rc = lib.cslam_device_count(C.byref(n))
    if rc != 0 or n.value < 1:
        msg = lib.cslam_last_error()
        raise CslamHipError("no HIP device visible: cslam_amd needs an MI355X (gfx950); "
                            f"no CPU fallback exists ({msg.decode() if msg else ''})")
    return n.value
