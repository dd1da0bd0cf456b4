"""Loader for the C-ABI library (classpro_amd/libclasspro_amd.so, built from csrc/ by build.py).

Fails loudly when the library is missing: there is no CPU or PyTorch fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLASSPRO_AMD_LIB") or os.path.join(_HERE, "libclasspro_amd.so")   # env: diagnostic builds only

# every symbol include/classpro_amd.h declares
SYMBOLS = [
    "cp_last_error", "cp_version", "cp_hist_covs", "cp_params_create", "cp_params_destroy",
    "cp_params_export", "cp_decode_profile", "cp_workspace_create", "cp_workspace_destroy",
    "cp_workspace_bytes", "cp_classify_batch", "cp_workspace_check", "cp_run_stages", "cp_get_counts",
    "cp_get_intervals", "cp_get_rel_asgn", "cp_get_bitmap", "cp_seq_context", "cp_scan_candidates",
    "cp_encode_profile", "cp_decode_profiles", "cp_params_create_model", "cp_params_create_pe", "cp_load_error_model", "cp_unpack_bases",
    "cp_find_seeds_batch", "cp_get_rep_masks", "cp_rep_masks_capacity", "cp_params_tables", "cp_pack_bases", "cp_pack_labels", "cp_unpack_labels", "cp_math_eval", "cp_pack_bases_batch",
    "cp_label_runs", "cp_label_runs_capacity", "cp_expand_label_runs",
]

_lib = None


class ClassProError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("classpro_amd error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "classpro_amd: %s not found. Build it with `python -m classpro_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # The Python layer shares device memory and streams with torch, so both must sit on ONE HIP runtime: torch's
    # (it ships its own libamdhip64) has to be in the process first -- loaded after ours, it comes up as a second
    # runtime and every later HIP call of this library answers "no ROCm-capable device".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.cp_last_error.restype = C.c_char_p
    L.cp_version.restype = C.c_char_p
    L.cp_hist_covs.argtypes = [vp, i32, i32, i64, i64, i32, C.POINTER(i32), C.POINTER(i32)]
    L.cp_params_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.cp_params_create_model.argtypes = [i32, i32, i32, i32, C.c_char_p, C.POINTER(vp)]
    L.cp_load_error_model.argtypes = [C.c_char_p, vp]
    L.cp_params_create_pe.argtypes = [i32, i32, i32, i32, vp, C.POINTER(vp)]
    L.cp_params_destroy.argtypes = [vp]
    L.cp_params_destroy.restype = None
    L.cp_params_export.argtypes = [vp] + [vp] * 7
    L.cp_params_tables.argtypes = [vp, vp, vp, vp]
    L.cp_math_eval.argtypes = [i32, vp, vp, vp, i64, vp]
    L.cp_label_runs.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.cp_label_runs_capacity.argtypes = [vp]
    L.cp_label_runs_capacity.restype = i64
    L.cp_expand_label_runs.argtypes = [vp, vp, i32, i32, i32, vp]
    L.cp_pack_bases.argtypes = [vp, i32, vp]
    L.cp_pack_bases_batch.argtypes = [vp, vp, i32, vp, vp, i32]
    L.cp_pack_labels.argtypes = [vp, vp, vp, i32, vp, vp]
    L.cp_unpack_labels.argtypes = [vp, i32, i32, vp]
    L.cp_decode_profile.argtypes = [vp, i64, vp, i32]
    L.cp_workspace_create.argtypes = [C.POINTER(vp)]
    L.cp_workspace_destroy.argtypes = [vp]
    L.cp_workspace_destroy.restype = None
    L.cp_workspace_bytes.argtypes = [vp]
    L.cp_workspace_bytes.restype = C.c_size_t
    L.cp_classify_batch.argtypes = [vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, vp]
    L.cp_workspace_check.argtypes = [vp]
    L.cp_run_stages.argtypes = [vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, i32, vp]
    L.cp_get_counts.argtypes = [vp, vp, vp, vp, vp]
    L.cp_get_intervals.argtypes = [vp, vp, vp, i64]
    L.cp_get_rel_asgn.argtypes = [vp, vp, vp, i64]
    L.cp_get_bitmap.argtypes = [vp, vp, i64]
    L.cp_seq_context.argtypes = [vp, vp, i32, i64, vp, vp, vp]
    L.cp_scan_candidates.argtypes = [vp, vp, i64, vp, vp]
    L.cp_encode_profile.argtypes = [vp, i32, vp, i64]
    L.cp_encode_profile.restype = i64
    L.cp_decode_profiles.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.cp_unpack_bases.argtypes = [vp, vp, vp, i32, vp, vp]
    L.cp_find_seeds_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, vp]
    L.cp_get_rep_masks.argtypes = [vp, vp, vp, vp, i64]
    L.cp_rep_masks_capacity.argtypes = [vp]
    L.cp_rep_masks_capacity.restype = i64
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise ClassProError(rc, lib().cp_last_error().decode(errors="replace"))
    return rc
