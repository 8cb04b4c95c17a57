"""ctypes binding of libcnerf_hip.so (C ABI: include/cnerf.h).  There is no fallback: if the library is missing or a
call fails this raises -- the product path never computes on the CPU."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcnerf_hip.so")

MAX_LAYERS = 16
MAX_LEVELS = 4
ABI_VERSION = 7
F_HIERARCHICAL, F_WHITE_BACK, F_LAST_BACK, F_SOFTPLUS, F_SIGMOID_RGB, F_INPUT_XYZ = 1, 2, 4, 8, 16, 32
PREC_FP32, PREC_FP16, PREC_FP16X3 = 0, 1, 2
PREC_CODE = {"fp32": PREC_FP32, "fp16": PREC_FP16, "fp16x3": PREC_FP16X3}
LAYER_FILM, LAYER_SINE, LAYER_RES, LAYER_PFILM = 0, 1, 2, 3
LAYER_CODE = {"film": LAYER_FILM, "sine": LAYER_SINE, "res": LAYER_RES, "pfilm": LAYER_PFILM}


class Cfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("V", C.c_int32), ("C", C.c_int32),
                ("H", C.c_int32), ("L", C.c_int32), ("layer_kind", C.c_int32 * MAX_LAYERS),
                ("ray_start", C.c_float), ("ray_end", C.c_float), ("voxel_length", C.c_float),
                ("noise_std", C.c_float), ("flags", C.c_uint32), ("fov_deg", C.c_double),
                ("n_levels", C.c_int32), ("level_V", C.c_int32 * MAX_LEVELS), ("level_C", C.c_int32 * MAX_LEVELS),
                ("precision", C.c_int32), ("philox", C.c_uint32), ("philox_offset", C.c_uint32), ("philox_seed", C.c_uint64),
                ("drop_p", C.c_float), ("reserved0", C.c_uint32)]


class Volumes(C.Structure):
    _fields_ = [("level", C.c_void_p * MAX_LEVELS)]


class FieldParams(C.Structure):
    _fields_ = [("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS), ("w2", C.c_void_p * MAX_LAYERS),
                ("b2", C.c_void_p * MAX_LAYERS), ("w_final", C.c_void_p), ("b_final", C.c_void_p),
                ("map_w1", C.c_void_p), ("map_b1", C.c_void_p), ("map_w2", C.c_void_p), ("map_b2", C.c_void_p)]


class FieldParamGrads(C.Structure):
    _fields_ = [("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS), ("w2", C.c_void_p * MAX_LAYERS),
                ("b2", C.c_void_p * MAX_LAYERS), ("w_final", C.c_void_p), ("b_final", C.c_void_p),
                ("map_w1", C.c_void_p), ("map_b1", C.c_void_p), ("map_w2", C.c_void_p), ("map_b2", C.c_void_p)]


class Saved(C.Structure):
    _fields_ = [("coarse_rgb_sigma", C.c_void_p), ("coarse_z", C.c_void_p), ("fine_rgb_sigma", C.c_void_p), ("fine_z", C.c_void_p)]


class Rng(C.Structure):
    _fields_ = [("u_strat", C.c_void_p), ("eps_coarse", C.c_void_p), ("u_fine", C.c_void_p), ("eps_final", C.c_void_p),
                ("fine_z", C.c_void_p), ("drop_coarse", C.c_void_p), ("drop_fine", C.c_void_p)]


AUX_FIELDS = ("coarse_points", "coarse_z", "coarse_rgb_sigma", "coarse_weights", "cdf", "inds", "fine_z",
              "fine_rgb_sigma", "sort_idx", "final_weights", "fine_points")


class Act16(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("h", C.c_void_p), ("c", C.c_void_p), ("amax", C.c_void_p)]


class Aux(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in AUX_FIELDS] + [("field_events", C.c_void_p * 4), ("act16", Act16 * 2)]


# name -> (restype, argtypes); every symbol include/cnerf.h declares
PROTOTYPES = {
    "cnerf_abi_version": (C.c_int, []),
    "cnerf_last_error": (C.c_char_p, []),
    "cnerf_philox_fill": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "cnerf_workspace_bytes": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cnerf_fvol_channel_last": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnerf_fvol_channel_first": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnerf_pack_field": (C.c_int, [C.POINTER(Cfg), C.POINTER(FieldParams), C.c_void_p, C.c_void_p]),
    "cnerf_gather_features": (C.c_int, [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cnerf_weight_grad": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "cnerf_scatter_features": (C.c_int, [C.POINTER(Cfg), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnerf_field_forward": (C.c_int, [C.POINTER(Cfg), C.POINTER(Volumes), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p]),
    "cnerf_composite": (C.c_int, [C.POINTER(Cfg), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnerf_resample": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "cnerf_render_forward": (C.c_int, [C.POINTER(Cfg), C.POINTER(Volumes), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(Rng), C.c_void_p, C.c_void_p, C.POINTER(Aux), C.c_void_p, C.c_void_p]),
    "cnerf_backward_bytes": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_size_t)]),
    "cnerf_pack_field_transposed": (C.c_int, [C.POINTER(Cfg), C.POINTER(FieldParams), C.c_void_p, C.c_void_p]),
    "cnerf_merge_composite_backward": (C.c_int, [C.POINTER(Cfg)] + [C.c_void_p] * 10),
    "cnerf_field_backward": (C.c_int, [C.POINTER(Cfg), C.c_int32, C.c_int32, C.c_int32, C.POINTER(Volumes)] + [C.c_void_p] * 14 +
                             [C.POINTER(Volumes), C.c_void_p, C.c_void_p]),
    "cnerf_weight_grad16": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6),
    "cnerf_backward16_bytes": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_size_t)]),
    "cnerf_pack_field_chain16": (C.c_int, [C.POINTER(Cfg), C.POINTER(FieldParams), C.c_void_p, C.c_void_p]),
    "cnerf_field_backward16": (C.c_int, [C.POINTER(Cfg), C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Volumes)] +
                               [C.c_void_p] * 16 + [C.POINTER(Volumes), C.c_void_p]),
    "cnerf_backward_workspace_bytes": (C.c_int, [C.POINTER(Cfg), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "cnerf_render_backward": (C.c_int, [C.POINTER(Cfg), C.c_int32, C.c_int32, C.POINTER(Volumes), C.POINTER(FieldParams), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Rng), C.POINTER(Saved), C.POINTER(Aux), C.c_void_p, C.c_void_p,
                                        C.POINTER(FieldParamGrads), C.c_void_p, C.c_void_p, C.POINTER(Volumes), C.c_void_p, C.c_void_p, C.c_void_p]),
}
B16_STORE, B16_DRY, B16_CHAIN = 1, 2, 4

_lib = None


class CnerfError(RuntimeError):
    pass


def lib():
    """The loaded library (cached).  Raises if it has not been built: run `python __graft_entry__.py build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CnerfError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                             f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.cnerf_abi_version() != ABI_VERSION:
            raise CnerfError("libcnerf_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cnerf_last_error().decode(errors="replace")
        raise CnerfError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device pointer of a CUDA tensor (None -> NULL).  Refuses anything that is not fp32/int32, contiguous, on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CnerfError("the HIP render path needs tensors on the GPU (there is no CPU fallback)")
    if not t.is_contiguous():
        raise CnerfError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())
