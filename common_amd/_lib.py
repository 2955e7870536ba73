"""ctypes binding of include/microscopes_hip.h (the C-ABI of the HIP library).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C
common_amd/csrc``) as ``common_amd/lib/libmicroscopes_hip.so``.  There is no
Python / numpy / torch fallback for any of its entry points: if the shared
object is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MSC_LIB_PATH: an experiment build of the same library, common_amd/csrc/Makefile VARIANT=...; the product is the default)
LIB_PATH = os.environ.get("MSC_LIB_PATH") or os.path.join(_HERE, "lib", "libmicroscopes_hip.so")

# families / primitive types / flags, mirrored from the header
BB, GP, DD, NICH, NIW, NOOP, BBNC, BNB, DM = range(9)
(TYPE_B, TYPE_I8, TYPE_U8, TYPE_I16, TYPE_U16, TYPE_I32, TYPE_U32, TYPE_I64, TYPE_U64, TYPE_F32,
 TYPE_F64) = range(11)
SCORE_CRP_PRIOR = 0x1
SCORE_NIW_F32 = 0x2
ACC_RESET, ACC_SUBTRACT, ACC_NO_COMMIT = 0x1, 0x2, 0x4
OP_ADD, OP_REMOVE, OP_SCORE_VALUE, OP_SCORE_DATA = range(4)
ABI_VERSION = 1


class MicroscopesHipError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "microscopes_hip error %d: %s" % (code, message))
        self.code = code


class RuntimeType(C.Structure):
    _fields_ = [("type", C.c_int32), ("count", C.c_uint32)]


class FeatureSpec(C.Structure):
    _fields_ = [("family", C.c_int32), ("dim", C.c_uint32)]


_SIGS = {
    "msc_abi_version": (C.c_int, []),
    "msc_last_error": (C.c_char_p, []),
    "msc_build_info": (C.c_char_p, []),
    "msc_last_kernel": (C.c_char_p, [C.c_int]),
    "msc_context_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "msc_context_destroy": (C.c_int, [C.c_void_p]),
    "msc_context_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "msc_context_synchronize": (C.c_int, [C.c_void_p]),
    "msc_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "msc_device_alloc_probed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "msc_device_alloc_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32)]),
    "msc_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "msc_pinned_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "msc_pinned_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "msc_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "msc_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "msc_dataview_from_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.POINTER(RuntimeType), C.c_uint32, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_void_p)]),
    "msc_dataview_from_device_columns": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(RuntimeType),
                                                   C.c_uint32, C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "msc_dataview_destroy": (C.c_int, [C.c_void_p]),
    "msc_dataview_invalidate": (C.c_int, [C.c_void_p]),
    "msc_dataview_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "msc_dataview_column": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                      C.POINTER(RuntimeType)]),
    "msc_state_create": (C.c_int, [C.c_void_p, C.POINTER(FeatureSpec), C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_void_p)]),
    "msc_state_destroy": (C.c_int, [C.c_void_p]),
    "msc_state_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "msc_hp_floats": (C.c_size_t, [C.c_int, C.c_uint32]),
    "msc_state_set_hp": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "msc_state_get_hp": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "msc_ss_bytes": (C.c_size_t, [C.c_int, C.c_uint32]),
    "msc_state_set_ss": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "msc_state_get_ss": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "msc_state_set_alpha": (C.c_int, [C.c_void_p, C.c_float]),
    "msc_state_set_group_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "msc_state_get_group_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "msc_score_value": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                  C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "msc_score_tune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "msc_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                 C.c_void_p, C.c_uint32]),
    "msc_entity_op": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]),
    "msc_score_data": (C.c_int, [C.c_void_p, C.c_void_p]),
    "msc_sweep_assign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]),
    "msc_sweep_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                 C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]),
    "msc_sweep_step_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]),
    "msc_sweep_step_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "msc_state_reduce_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "msc_state_commit_reduce": (C.c_int, [C.c_void_p]),
    "msc_state_reduce_pack": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "msc_state_reduce_unpack": (C.c_int, [C.c_void_p]),
    "msc_state_set_sweep_rows": (C.c_int, [C.c_void_p, C.c_uint64]),
    "msc_comm_unique_id_bytes": (C.c_size_t, []),
    "msc_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "msc_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "msc_comm_adopt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "msc_comm_destroy": (C.c_int, [C.c_void_p]),
    "msc_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "msc_state_allreduce": (C.c_int, [C.c_void_p, C.c_void_p]),
    "msc_sweep_step_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.c_uint64, C.c_uint64, C.c_void_p]),
    "msc_accumulate_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "msc_value_op_single": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "msc_relation_slice_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64]),
    "msc_relation_blocks": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p]),
}

EXPORTS = tuple(sorted(_SIGS))
_lib = None


def load():
    """dlopen the HIP library (torch first, so both share one libamdhip64)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C common_amd/csrc).  common_amd has no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (loads the ROCm runtime the library binds to)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.msc_abi_version() != ABI_VERSION:
        raise ImportError("libmicroscopes_hip.so ABI %d != binding ABI %d" %
                          (lib.msc_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MicroscopesHipError(rc, load().msc_last_error().decode("utf-8", "replace"))
