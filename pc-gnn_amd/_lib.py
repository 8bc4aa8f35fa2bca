"""ctypes binding of libpcgnn_hip.so (the C ABI in include/pcgnn.h).

There is NO fallback: if the library is missing or an entry point is absent the
import of the product path raises.  Build it with ``python -m pcgnn_amd.build``
(or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

from .build import lib_path

PCG_MAX_REL = 8
PCG_OK, PCG_E_ARG, PCG_E_UNSUPPORTED, PCG_E_LAUNCH = 0, -1, -2, -3
PCG_ST_SEL_OVERFLOW = 1
PCG_ST_LIST_ID_RANGE = 2
PCG_ST_SYNC_TIMEOUT = 4
PCG_ST_SORT_OVERFLOW = 8
PCG_NORM_COUNT, PCG_NORM_SQRT_COUNT = 0, 1
ABI_VERSION = 3


class GraphDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64),
        ("feat_dim", C.c_int32),
        ("feat_stride", C.c_int32),
        ("n_rel", C.c_int32),
        ("n_pos", C.c_int32),
        ("max_degree", C.c_int32),
        ("_pad", C.c_int32),
        ("X", C.c_void_p),
        ("train_pos", C.c_void_p),
        ("indptr", C.c_void_p * PCG_MAX_REL),
        ("indices", C.c_void_p * PCG_MAX_REL),
    ]


_P, _I32, _I64, _U64, _F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
_G = C.POINTER(GraphDesc)

# name -> (restype, argtypes); must list every symbol include/pcgnn.h declares
PROTOTYPES = {
    "pcg_version": (C.c_char_p, []),
    "pcg_abi_version": (C.c_int, []),
    "pcg_score_table": (C.c_int, [_G, _P, _P, _I64, _I64, _P, _P]),
    "pcg_score_rows": (C.c_int, [_G, _P, _P, _P, _I32, _P, _P]),
    "pcg_pos_sort_capacity": (_I64, [_I32]),
    "pcg_pos_sort": (C.c_int, [_G, _P, _P, _P]),
    "pcg_choose_workspace_bytes": (_I64, [_G, _I32, _I64]),
    "pcg_choose_workspace_offset": (_I64, [_G, _I32, _I64, _I32]),
    "pcg_choose_select": (C.c_int, [_G, _P, _P, _I32, _P, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                    _P, _P, _I64, _P, _P]),
    "pcg_aggregate_lists": (C.c_int, [_P, _I32, _I32, _I64, _I32, _P, _G, _I32, _P, _I64, _I32, _P, _I32, _P, _P]),
    "pcg_choose_aggregate": (C.c_int, [_G, _P, _P, _I32, _P, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                       _I32, _P, _I32, _P, _P, _I64, _P, _P]),
    "pcg_step_front": (C.c_int, [_G, _P, _P, _P, _P, _P, _P, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32, _P, _I64,
                                 _P, _P]),
    "pcg_step_front_a": (C.c_int, [_G, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                   _P, _I64, _P, _P]),
    "pcg_step_front_b": (C.c_int, [_G, _P, _P, _I32, _P, _P, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32, _P, _I64, _P,
                                   _P, _I64, _P]),
    "pcg_choose_select_planned": (C.c_int, [_G, _P, _P, _I32, _P, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                            _P, _P, _P, _I64, _P, _P, _I64, _P]),
    "pcg_step_scores": (C.c_int, [_G, _P, _P, _I64, _I64, _P, _P, _P, _I64, _P, _P, _P]),
    "pcg_choose_gather_train": (C.c_int, [_G, _P, _P, _I32, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _P, _I32, _P, _P, _P,
                                          _I64, _P, _P, _P, _P, _P, _I32, _P, _P, _P, C.c_float, C.c_float, _F64, _F64, _F64, _F64,
                                          _F64, _I32, _P, _P, _I32, _P, _I32, _P]),
    "pcg_choose_plan_bytes": (_I64, [_G, _I32, _I64]),
    "pcg_choose_data_bytes": (_I64, [_G, _I32, _I64]),
    "pcg_plan_batches": (C.c_int, [_G, _P, _P, _I32, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32, _P, _I64, _I64, _P, _P,
                                   _P]),
    "pcg_plan_epochs": (C.c_int, [_G, _P, _P, _I32, _I32, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32, _P, _I64, _I64, _P, _P,
                                  _P]),
    "pcg_pick_shuffled_epochs": (C.c_int, [_P, _P, _I32, _U64, _U64, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "pcg_pos_sort_in_select": (_I32, [_I32]),
    "pcg_pos_sort_one_launch": (_I32, [_I32]),
    "pcg_pos_sort_raw": (C.c_int, [_G, _P, _P, _P]),
    "pcg_sync_words_count": (_I32, []),
    "pcg_aggregate_lists_planned": (C.c_int, [_P, _I32, _I32, _I64, _I32, _P, _G, _I32, _P, _P, _I64, _I32, _P, _I32, _P, _P]),
    "pcg_gather_lists_planned": (C.c_int, [_P, _I32, _I32, _I64, _I32, _P, _G, _I32, _P, _P, _I64, _P, _I32, _P, _P]),
    "pcg_step_scores_train": (C.c_int, [_G, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _F64, _F64, _F64, _F64, _F64, _P, _P]),
    "pcg_touched_bytes": (_I64, [_I64]),
    "pcg_mark_touched": (C.c_int, [_G, _P, _I32, _I32, _P, _I64, _P, _P]),
    "pcg_mark_touched_planned": (C.c_int, [_G, _P, _I32, _I32, _P, _I64, _I64, _P, _I64, _P]),
    "pcg_choose_aggregate_planned": (C.c_int, [_G, _P, _P, _I32, _P, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                               _I32, _P, _I32, _P, _P, _I64, _P, _P]),
    "pcg_choose_gather_planned": (C.c_int, [_G, _P, _P, _I32, _P, _P, _P, C.POINTER(_F64), C.POINTER(_F64), _I32, _I32,
                                            _P, _I32, _P, _P, _P, _I64, _P, _P, _P]),
    "pcg_gather_lists": (C.c_int, [_P, _I32, _I32, _I64, _I32, _P, _G, _I32, _P, _I64, _P, _I32, _P, _P]),
    "pcg_train_dense": (C.c_int, [_G, _P, _P, _P, _I32, _P, _P, _I32, _P, _I32, _P, _P, _P, _I64, C.c_float, C.c_float, _P, _P, _P,
                                  _P, _P, _P, _P, _F64, _F64, _F64, _F64, _F64, _I32, _P, _I32, _P, _P]),
    "pcg_dense_sorts_keys": (_I32, [_I32, _I32]),
    "pcg_step_front_train": (C.c_int, [_G, _P, _P, _P, _I32, _P, _P, _P, _P, _I32, C.POINTER(_F64), C.POINTER(_F64), _I32, _P,
                                       _I64, _P, _P, _P, _P, _F64, _F64, _F64, _F64, _F64, _P]),
    "pcg_grad_reduce": (C.c_int, [_P, _I32, _I64, _P, _P, _P]),
    "pcg_adam_apply_pending": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I32, _F64, _F64, _F64, _F64, _F64, _P]),
    "pcg_adam_flush": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _I64, _P, _P, _F64, _F64, _F64, _F64, _F64, _P, _P, _I32, _I32, _I32,
                                 _I32, _P, _P]),
    "pcg_wgrad_act_rows": (_I64, [_I32, _I32, _I32]),
    "pcg_wgrad_scratch_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "pcg_wgrad": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _F64, _F64, _F64, _F64, _F64, _P, _I32, _I32, _P, _P, _P]),
    "pcg_step_scores_dist": (C.c_int, [_G, _P, _P, _P, _I32, _P, _P, _I64, _I64, _P, _P, _P, _I64, _P, _P, _F64, _F64, _F64, _F64, _F64,
                                       _P]),
    "pcg_gather_lists_dist": (C.c_int, [_P, _I32, _I32, _I64, _I32, _P, _G, _I32, _P, _P, _I64, _P, _I32, _P, _I32, _I32, _I32, _P, _P,
                                        _I32, _P, _I64, _P, _I32, _I32, _P, _P, _P, _I64, _I32, _P, _P]),
    "pcg_debug_set_stamps": (None, [_P]),
    "pcg_debug_set_dense_stamps": (None, [_P]),
    "pcg_sel_capacity_row": (_I64, [_I64, _F64, _F64, _I32, _I32, _I32]),
    "pcg_segment_mean": (C.c_int, [_G, _P, _P, _P, _I32, _I32, _P, _I32, _P]),
    "pcg_pick": (C.c_int, [_P, _P, _I32, _P, _U64, _U64, _I32, _P, _P]),
    "pcg_pick_shuffled": (C.c_int, [_P, _P, _I32, _U64, _U64, _P, _I32, _I32, _P, _P, _P, _P]),
    "pcg_gather_rows": (C.c_int, [_G, _P, _I32, _P, _I32, _P]),
    "pcg_halo_table_slots": (_I64, [_I32]),
    "pcg_halo_serve": (C.c_int, [_G, _P, _I32, _I32, _I32, _P, _I32, _P]),
    "pcg_halo_collect": (C.c_int, [_G, _P, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _I64, _P, _P, _I32, _I32, _I32, _I32,
                                   _P]),
    "pcg_halo_lookup": (C.c_int, [_G, _I32, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _I32, _P, _I64, _P, _I32, _I32, _P]),
    "pcg_dense_n_params": (_I64, [_I32, _I32, _I32]),
    "pcg_dense_param_offset": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "pcg_dense_n_tiles": (_I32, [_I32]),
    "pcg_dense_step": (C.c_int, [_G, _P, _I32, _P, _P, _I32, _P, _I32, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P,
                                 _P]),
    "pcg_adam_step": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _P, _F64, _F64, _F64, _F64, _F64, _P, _I32, _P]),
}

_lib = None


class PcgnnLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise (never fall back) if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise PcgnnLibraryError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc). There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PcgnnLibraryError(f"{path} does not export {name}: stale build? rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.pcg_abi_version() != ABI_VERSION:
        raise PcgnnLibraryError(f"{path}: ABI {lib.pcg_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


_ERR = {PCG_E_ARG: "bad argument", PCG_E_UNSUPPORTED: "unsupported shape", PCG_E_LAUNCH: "kernel launch failed"}


def check(rc, what):
    if rc != PCG_OK:
        raise PcgnnLibraryError(f"{what} failed: {_ERR.get(rc, rc)} ({rc})")
