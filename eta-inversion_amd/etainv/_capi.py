"""ctypes binding of libetainv_hip.so (include/etainv.h).  No fallback: if the library is missing or a call
fails, an exception is raised -- the product path never computes on the CPU."""
import ctypes as C
import os
from pathlib import Path

F32, F16, BF16 = 0, 1, 2
ATTN_PLAIN, ATTN_STORE, ATTN_PTP, ATTN_MASA = 0, 1, 2, 3

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libetainv_hip.so"


class EtainvError(RuntimeError):
    pass


class EngineConfig(C.Structure):
    _fields_ = [("compute_dtype", C.c_int), ("max_unet_batch", C.c_int), ("latent_size", C.c_int), ("max_img", C.c_int),
                ("reserved", C.c_int * 4)]


class AttnCtrl(C.Structure):
    _fields_ = [("mode", C.c_int), ("n_img", C.c_int), ("store_maps", C.c_int),
                ("mapper", C.c_void_p), ("alphas", C.c_void_p), ("replace_mat", C.c_void_p), ("equalizer", C.c_void_p),
                ("cross_alpha", C.c_void_p),
                ("self_replace_active", C.c_int), ("self_max_tokens", C.c_int),
                ("masa_active", C.c_int), ("masa_first_block", C.c_int), ("first_row", C.c_int), ("src_exit_block", C.c_int), ("reserved", C.c_int * 2)]


_p, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/etainv.h one to one
SIGNATURES = {
    "etainv_abi_version": [],
    "etainv_last_error": [],
    "etainv_cfg_combine": [_p, _p, _f, _p, _i64, _i, _p],
    "etainv_ddim_step": [_p, _p, _f, _f, _p, _i64, _i, _p],
    "etainv_ddim_eta_step": [_p, _p, _f, _p, _i, _p, _f, _f, _f, _i, _i, _i, _p, _i, _p],
    "etainv_eta_backward_step": [_p, _p, _f, _p, _p, _i, _f, _p, _f, _i, _f, _f, _f, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "etainv_eta_backward_step_ex": [_p, _p, _f, _p, _p, _i, _f, _p, _f, _i, _f, _f, _f, _i, _i, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p],
    "etainv_lincomb3": [_p, _f, _p, _f, _p, _f, _p, _i64, _i, _p],
    "etainv_engine_create": [C.POINTER(EngineConfig), C.POINTER(_p)],
    "etainv_engine_destroy": [_p],
    "etainv_engine_num_weights": [_p],
    "etainv_engine_weight_info": [_p, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i)],
    "etainv_engine_set_weight": [_p, C.c_char_p, _p, _i64, _p],
    "etainv_engine_weights_ready": [_p],
    "etainv_unet_forward": [_p, _p, _i, C.POINTER(_i64), _p, _i, C.POINTER(AttnCtrl), _p, _i, _p],
    "etainv_maps_reset": [_p, _p],
    "etainv_maps_word_maps": [_p, _i, _p, _i, _i, _p, _i, _f, _p],
    "etainv_maps_word_maps_role": [_p, _i, _p, _i, _i, _i, _p, _i, _f, _p],
    "etainv_maps_configure": [_p, _i, _p],
    "etainv_maps_word_maps_ex": [_p, _i, _p, _i, _i, _i, C.c_uint, _p, _i, _f, _p],
    "etainv_local_blend": [_p, _p, _i, _p, _f, _p],
    "etainv_engine_graph_stats": [_p, C.POINTER(_i64), C.POINTER(_i64)],
    "etainv_engine_qkv_head_major_count": [_p, C.POINTER(C.c_longlong)],
    "etainv_engine_workspace_bytes": [_p],
    "etainv_engine_weight_bytes": [_p],
    "etainv_prof_enable": [_i],
    "etainv_prof_reset": [],
    "etainv_prof_read": [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)],
    "etainv_prof_records": [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), _i64, C.POINTER(_i64)],
    "etainv_prof_records_ex": [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _i64, C.POINTER(_i64)],
    "etainv_prof_split": [_i, C.c_double, C.POINTER(C.c_double), C.POINTER(_i64)],
    "etainv_op_gemm": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "etainv_op_gemm_ln": [_p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_i), _i, _i, _i, _i, _i, _p],
    "etainv_engine_cache_context": [_p, _i],
    "etainv_engine_context_generation": [_p, C.c_uint64],
    "etainv_op_gemm_gnstat": [_p, _p, _p, _p, _p, _p, C.POINTER(_i), _i, _i, _i, _i, _i, _p],
    "etainv_op_conv3x3_gnstat": [_p, _p, _p, _p, _p, _p, _p, C.POINTER(_i), _i, _i, _i, _i, _i, _i, _p],
    "etainv_op_groupnorm_pre": [_p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p],
    "etainv_op_ln_fold": [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _i, _p],
    "etainv_op_row_stats": [_p, _p, _i, _i, _f, _i, _p],
    "etainv_op_ln_finalize": [_p, _i, _i, _f, _p, _i, _p],
    "etainv_op_pack_ups4": [_p, _p, _i, _i, _i, _p],
    "etainv_op_conv3x3": [_p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "etainv_op_conv3x3_ex": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "etainv_op_im2col3x3": [_p, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    "etainv_op_row_softmax": [_p, _i, _i, _f, _i, _p],
    "etainv_op_quick_gelu": [_p, _p, _i64, _i, _p],
    "etainv_op_embed": [_p, _p, _p, _i, _i, _i, _p, _i, _p],
    "etainv_op_causal_attention": [_p, _p, _i, _i, _i, _i, _i, _p],
    "etainv_op_groupnorm": [_p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p],
    "etainv_op_layernorm": [_p, _p, _p, _p, _i, _i, _f, _i, _p],
    "etainv_op_self_attention": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "etainv_op_cross_attention": [_p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(AttnCtrl), _i, _i, _p, _i, _p],
    "etainv_op_word_maps": [_p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _i, _f, _p],
    "etainv_op_local_blend": [_p, _i, _i, _i, _i, _i, _p, _i, _p, _f, _p],
}
_RESTYPES = {"etainv_last_error": C.c_char_p, "etainv_engine_workspace_bytes": _i64, "etainv_engine_weight_bytes": _i64}

_lib = None


def lib_path() -> Path:
    return Path(os.environ.get("ETAINV_LIB", _LIB_PATH))


def load():
    """Load the shared library (once).  Raises EtainvError when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not path.exists():
            raise EtainvError(f"{path} not found: build it with eta-inversion_amd/csrc/build.sh (no CPU fallback exists)")
        lib = C.CDLL(str(path))
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def check(status: int):
    if status != 0:
        raise EtainvError(load().etainv_last_error().decode())


def dtype_code(dt) -> int:
    import torch
    return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[dt]


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "etainv needs contiguous device tensors"
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
