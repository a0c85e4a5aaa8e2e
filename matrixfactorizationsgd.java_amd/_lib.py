"""ctypes binding of include/mfsgd.h -- the stub a reference-side maintainer
would write (INTEGRATION.md shows the JNI equivalent)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    """The in-tree build; MFSGD_LIBRARY names another build of the same ABI (kernel experiments)."""
    return os.environ.get("MFSGD_LIBRARY") or os.path.join(_HERE, "lib", "libmfsgd.so")


def rehearsal_library_path():
    """libmfsgd_rehearsal.so: libmfsgd.so plus the DSGD ring's shared-memory rehearsal transport (several ranks on one
    GPU).  Tests and bench.py --rehearse-on-one-gpu point MFSGD_LIBRARY at it; the product library does not contain it."""
    return os.path.join(_HERE, "lib", "libmfsgd_rehearsal.so")


class Config(C.Structure):
    _fields_ = [
        ("n_users", C.c_int32),
        ("n_items", C.c_int32),
        ("k", C.c_int32),
        ("lr", C.c_float),
        ("lambda_", C.c_float),
        ("device", C.c_int32),
        ("blocks", C.c_int32),
        ("waves", C.c_int32),
        ("n_parts", C.c_int32),
        ("host_threads", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class ScheduleInfo(C.Structure):
    _fields_ = [
        ("nnz", C.c_int64),
        ("part", C.c_int32),
        ("blocks", C.c_int32),
        ("waves", C.c_int32),
        ("group_lanes", C.c_int32),
        ("slots", C.c_int32),
        ("kp", C.c_int32),
        ("rounds", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("total_steps", C.c_int64),
        ("total_rows", C.c_int64),
        ("max_cell_nnz", C.c_int64),
        ("max_cell_rows", C.c_int64),
        ("max_cell_steps", C.c_int64),
        ("sum_round_steps", C.c_int64),
        ("build_seconds", C.c_double),
        ("swapped", C.c_int32),
        ("device_ingest", C.c_int32),
        ("chunks", C.c_int64),
        ("split_cells", C.c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


FLAG_NO_GRAPH = 1
FLAG_ROUND_LAUNCH = 2
FLAG_HOST_INGEST = 4
FLAG_DEVICE_INGEST = 8
FLAG_NO_SOLO = 16
FLAG_HOST_PACK = 32

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_H = C.c_void_p

# every symbol include/mfsgd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mfsgd_abi_version": (C.c_int, []),
    "mfsgd_device_count": (C.c_int, [_i32p]),
    "mfsgd_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "mfsgd_destroy": (None, [_H]),
    "mfsgd_last_error": (C.c_char_p, [_H]),
    "mfsgd_set_ratings": (C.c_int, [_H, _i32p, _i32p, _f32p, C.c_int64]),
    "mfsgd_init_factors": (C.c_int, [_H, C.c_int64]),
    "mfsgd_set_factors": (C.c_int, [_H, _f32p, _f32p]),
    "mfsgd_get_factors": (C.c_int, [_H, _f32p, _f32p]),
    "mfsgd_train": (C.c_int, [_H, C.c_int32, _f64p]),
    "mfsgd_rmse": (C.c_int, [_H, _f64p]),
    "mfsgd_predict": (C.c_int, [_H, _i32p, _i32p, _f32p, C.c_int64]),
    "mfsgd_recommend": (C.c_int, [_H, _i32p, C.c_int32, C.c_int32, _i32p, _f32p]),
    "mfsgd_train_timed": (C.c_int, [_H, C.c_int32, _f64p, _i64p]),
    "mfsgd_ratings_file_open": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_H)]),
    "mfsgd_ratings_file_info": (C.c_int, [_H, _i64p, _i32p, _i32p]),
    "mfsgd_ratings_file_read": (C.c_int, [_H, _i32p, _i32p, _f32p, _i64p, _i64p]),
    "mfsgd_ratings_file_close": (None, [_H]),
    "mfsgd_io_last_error": (C.c_char_p, []),
    "mfsgd_get_dims": (C.c_int, [_H, _i32p, _i32p, _i32p]),
    "mfsgd_save_factors": (C.c_int, [_H, C.c_char_p]),
    "mfsgd_load_factors": (C.c_int, [_H, C.c_char_p]),
    "mfsgd_get_schedule_info": (C.c_int, [_H, C.c_int32, C.POINTER(ScheduleInfo)]),
    "mfsgd_get_order": (C.c_int, [_H, C.c_int32, _i64p, _i64p]),
    "mfsgd_debug_schedule_sizes": (C.c_int, [_H, C.c_int32, _i64p, _i64p, _i64p, _i64p]),
    "mfsgd_debug_get_schedule": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mfsgd_debug_epoch_profile": (C.c_int, [_H, C.POINTER(C.c_uint64), _i32p]),
    "mfsgd_debug_counters": (C.c_int, [_H, _i64p]),
    "mfsgd_debug_occupy": (C.c_int, [_H, C.c_int32]),
    "mfsgd_debug_round_stamps": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "mfsgd_dsgd_plan": (C.c_int, [_i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]),
    "mfsgd_dsgd_plan_ex": (C.c_int, [_i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _i32p, _i32p, _i64p]),
    "mfsgd_set_item_partition": (C.c_int, [_H, _i32p]),
    "mfsgd_get_item_partition": (C.c_int, [_H, _i32p, _i32p]),
    "mfsgd_part_rows": (C.c_int, [_H, C.c_int32, _i32p]),
    "mfsgd_part_init_q": (C.c_int, [_H, C.c_int32, C.c_int64, C.c_int64, _f32p]),
    "mfsgd_part_train": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p]),
    "mfsgd_part_sse": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, _f64p]),
    "mfsgd_init_p_offset": (C.c_int, [_H, C.c_int64, C.c_int64]),
    "mfsgd_part_sync": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "mfsgd_part_settle": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, _i32p]),
    "mfsgd_get_parts": (C.c_int, [_H, _i32p, _i32p, _i32p]),
    "mfsgd_dsgd_unique_id": (C.c_int, [C.c_void_p]),
    "mfsgd_dsgd_create": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(_H)]),
    "mfsgd_dsgd_destroy": (None, [_H]),
    "mfsgd_dsgd_last_error": (C.c_char_p, [_H]),
    "mfsgd_dsgd_init_q": (C.c_int, [_H, C.c_int64, C.c_int64]),
    "mfsgd_dsgd_set_q": (C.c_int, [_H, C.c_int32, _f32p]),
    "mfsgd_dsgd_get_q": (C.c_int, [_H, C.c_int32, _i32p, _i32p, _f32p]),
    "mfsgd_dsgd_train": (C.c_int, [_H, C.c_int32, _f64p]),
    "mfsgd_dsgd_rmse": (C.c_int, [_H, _f64p]),
    "mfsgd_dsgd_train_timed": (C.c_int, [_H, C.c_int32, _f64p]),
    "mfsgd_dsgd_allreduce": (C.c_int, [_H, _f64p, C.c_int32]),
    "mfsgd_dsgd_stats": (C.c_int, [_H, _i64p]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so /
    libhsa-runtime64.so; libmfsgd.so is linked against the system ROCm's.  Both have the SONAME
    libamdhip64.so.7, but torch asks for "libamdhip64.so", so when libmfsgd.so comes first the
    process ends up with two runtimes -- and on some hosts whichever initialises second then sees
    no GPU (measured here: hipGetDeviceCount() == 0 / "No HIP GPUs are available").  Loading
    torch's copy first makes the dynamic loader resolve libmfsgd.so's dependency to it (SONAME
    match), which is also what happens whenever `import torch` precedes this module.  torch itself
    is not imported; without a torch installation nothing is done."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # a broken torch installation must not keep the library from loading
        pass


def share_torch_rccl():
    """One RCCL per process where possible: libmfsgd binds librccl.so.1 at run time (dlopen); loading
    PyTorch's bundled copy first makes that resolve to the copy torch itself uses (SONAME match)."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """Loads libmfsgd.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C matrixfactorizationsgd.java_amd/csrc` (there is no CPU fallback)"
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
