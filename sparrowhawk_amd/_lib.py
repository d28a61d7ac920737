"""ctypes binding of libshk_hip.so — the C ABI declared in include/shk.h.

The library is built in-tree by `__graft_entry__.build()` (or `make -C sparrowhawk_amd/csrc`).
There is no fallback: if the shared object is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libshk_hip.so")

PROGRESS_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


class ShkPacked(C.Structure):
    _fields_ = [("bases", C.POINTER(C.c_uint32)), ("seg_off", C.POINTER(C.c_uint32)),
                ("n_seg", C.c_uint64), ("n_bases", C.c_uint64), ("n_reads", C.c_uint64),
                ("n_input_bases", C.c_uint64)]


# every symbol include/shk.h declares: name -> (restype, argtypes)
_u64, _u32, _vp, _cp, _int, _sz = C.c_uint64, C.c_uint32, C.c_void_p, C.c_char_p, C.c_int, C.c_size_t
SIGNATURES = {
    "shk_new": (_vp, [_u32, _int, _u32, _u32, _u64, _int, _int, _int, _int]),
    "shk_new_error": (_int, []),
    "shk_new_error_message": (_cp, []),
    "shk_free": (None, [_vp]),
    "shk_last_error": (_cp, [_vp]),
    "shk_set_progress_cb": (None, [_vp, PROGRESS_CB, _vp]),
    "shk_preprocess": (_int, [_vp, _cp, _sz, _cp, _sz]),
    "shk_push_reads": (_int, [_vp, _cp, _sz]),
    "shk_finish_reads": (_int, [_vp]),
    "shk_preprocess_packed_device": (_int, [_vp, _vp, _vp, _u64, _u64, _u64]),
    "shk_preprocess_packed_host": (_int, [_vp, _vp, _vp, _u64, _u64, _u64]),
    "shk_get_preprocessing_info": (_cp, [_vp]),
    "shk_assemble": (_int, [_vp]),
    "shk_get_assembly": (_cp, [_vp]),
    "shk_shard_partition": (_int, [_vp, _vp, _vp, _u64, _u64, _u64, _u32, _vp]),
    "shk_shard_record_bytes": (_u32, [_vp]),
    "shk_shard_pack": (_int, [_vp, _vp, _vp, _u32]),
    "shk_shard_count": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp]),
    "shk_shard_rows": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "shk_shard_set_solid": (_int, [_vp, _vp, _vp, _u64, _u64]),
    "shk_comm_unique_id": (_int, [_vp]),
    "shk_comm_init": (_vp, [_vp, _int, _int]),
    "shk_comm_error": (_cp, []),
    "shk_comm_rank": (_int, [_vp]),
    "shk_comm_world": (_int, [_vp]),
    "shk_comm_free": (None, [_vp]),
    "shk_shard_preprocess": (_int, [_vp, _vp, _vp, _vp, _u64, _u64, _u64, _u32]),
    "shk_plan_exchange": (_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "shk_choose_partitions": (_u32, [_u64, _u32, _u32]),
    "shk_pack_fastq": (_int, [_cp, _sz, _u32, _u32, C.POINTER(ShkPacked), C.POINTER(_cp)]),
    "shk_packed_free": (None, [C.POINTER(ShkPacked)]),
    "shk_key_words": (_u32, [_vp]),
    "shk_total_instances": (_u64, [_vp]),
    "shk_n_distinct": (_u64, [_vp]),
    "shk_get_distinct": (_int, [_vp, _vp, _vp, _u64]),
    "shk_n_solid": (_u64, [_vp]),
    "shk_get_solid": (_int, [_vp, _vp, _vp, _u64]),
    "shk_get_histo": (_int, [_vp, _vp]),
    "shk_used_min_count": (_u32, [_vp]),
    "shk_get_adjacency": (_int, [_vp, _vp, _vp, _vp, _u64]),
    "shk_get_timings": (_cp, [_vp]),
    "shk_peak_device_bytes": (_u64, [_vp]),
    "shk_host_mem_counter": (None, [_vp, _sz, C.POINTER(_u64), C.POINTER(_u64)]),
    "shk_host_canonical": (_int, [_cp, _u32, _vp, C.POINTER(_int)]),
    "shk_host_nthash": (_u64, [_cp, _u32]),
    "shk_host_fit": (_int, [_vp, C.POINTER(_u32)]),
    "shk_host_assembly_json": (_vp, [_cp, _vp, _vp, _u64, _u32]),
    "shk_host_assembly_json_arriving": (_vp, [_cp, _vp, _vp, _u64, _u32, _u64, _u32]),
    "shk_host_free": (None, [_vp]),
    "shk_host_gunzip": (_int, [_cp, _sz, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_u64), C.POINTER(C.c_double)]),
    "shk_device_gunzip": (_int, [_cp, _sz, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
    "shk_host_unitig_assemble": (_vp, [_u32, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int]),
    "shk_release_cached_memory": (None, []),
    "shk_measure_stream_read": (_int, [_sz, _int, C.POINTER(C.c_double)]),
    "shk_version": (_cp, []),
}

_lib = None


def load():
    """Load libshk_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C sparrowhawk_amd/csrc` (there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib
