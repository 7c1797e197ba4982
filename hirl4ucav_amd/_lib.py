"""ctypes binding of libhx_mi355.so (the C ABI declared in include/hirl4ucav.h).

There is NO fallback: if the HIP extension is missing or a call fails, this raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C hirl4ucav_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HX_LIBRARY selects another build of the same library (A/B timing of two builds on one box, the phase-stamp build); never a fallback
SO_PATH = os.environ.get("HX_LIBRARY") or os.path.join(_HERE, "libhx_mi355.so")

ENV_WORDS, OBS_DIM, ACT_DIM, ROW_WORDS = 37, 13, 4, 32
STAT_NAMES = ("episodes", "kills", "fire_success_episodes", "time_limit", "fires", "good_fires", "locked_steps", "env_steps", "nonfinite_actions")
STAT_WAYS, STAT_PITCH = 32, 16  # HX_STAT_WAYS, HX_STAT_PITCH: the counters are kept 32 times (way = workgroup % 32, one 128-byte line each); a statistic = the sum

F_LOCKED_PREV, F_LOCKED, F_SLOT_PREV, F_SLOT, F_FIRED, F_FIRE_SUCCESS, F_EPISODE_SUCCESS, F_DONE = (1 << i for i in range(8))
F_SCEN_SHIFT = 8
F_SERP_POS, F_SERP_LONG, F_M_ACTIVE, F_M_GUIDED, F_SIM_SLOT = (1 << i for i in range(10, 15))

_vp, _i32, _i64, _u32, _u64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float


class HxStepOpts(ctypes.Structure):
    _fields_ = [("max_step", _i32), ("auto_reset", _i32), ("randomize", _i32), ("env_id0", _u32), ("seed", _u64),
                ("episode_ctr", _vp), ("ring", _vp), ("ring_success", _vp), ("cap", _i64), ("total", _vp), ("stats", _vp),
                ("ev_start", _vp), ("ev_stop", _vp), ("layout", _i32)]


def layout(pair, envs_per_block):
    """HX_LAYOUT(pair, envs_per_block) of include/hirl4ucav.h: force the env-step launch shape (0 = library's choice)."""
    return ((1 if pair else 0) << 8) | (int(envs_per_block) // 4)


class HxError(RuntimeError):
    pass


_SIGNATURES = {
    "hx_env_reset": [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _u64, _u32, _vp, _vp, _vp],
    "hx_env_step": [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(HxStepOpts), _vp],
    "hx_env_rearm": [_vp, _i64, _i64, _vp, _vp],
    "hx_label_transitions": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "hx_event_destroy": [_vp],
    "hx_event_elapsed_us": [_vp, _vp, ctypes.POINTER(ctypes.c_float)],
}

_lib = None
ABI_VERSION = 114  # HX_ABI_VERSION of include/hirl4ucav.h
_ABI_STRUCTS = {}  # name -> (index in hx_abi_sizes, ctypes class): filled by the binding modules (check_struct)


def check_struct(index, cls):
    """Compare ctypes.sizeof(cls) with the library's sizeof of the same struct (hx_abi_sizes): a binding that has drifted from the header
    would hand the kernels short buffers (a 9-word statistics buffer under a library that adds into 512 words)."""
    sizes = (ctypes.c_int32 * 8)()
    L = load()
    L.hx_abi_sizes.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    if L.hx_abi_sizes(sizes) != 0 or sizes[index] != ctypes.sizeof(cls):
        raise HxError(f"ABI mismatch: {cls.__name__} is {ctypes.sizeof(cls)} bytes here, {sizes[index]} in {SO_PATH}")
    if sizes[7] != STAT_WAYS * STAT_PITCH:
        raise HxError(f"ABI mismatch: a statistics buffer is {STAT_WAYS * STAT_PITCH} words here, {sizes[7]} in {SO_PATH}")


def load():
    """Load the shared library (once).  Raises HxError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HxError(f"{SO_PATH} not found: the HIP extension is not built (run __graft_entry__.build() or "
                      f"`make -C hirl4ucav_amd/csrc`). There is no CPU fallback.")
    # torch first: libhx_mi355.so must bind to the SAME HIP runtime (libamdhip64) torch has loaded, because
    # device pointers and streams are shared with torch.  Loaded the other way round the process would hold two
    # runtimes and ours would see no device.
    import torch  # noqa: F401

    L = ctypes.CDLL(SO_PATH)
    L.hx_last_error.restype = ctypes.c_char_p
    L.hx_version.restype = ctypes.c_int
    if L.hx_version() != ABI_VERSION:
        raise HxError(f"{SO_PATH} implements ABI version {L.hx_version()}, this binding was written for {ABI_VERSION} (include/hirl4ucav.h "
                      f"HX_ABI_VERSION): rebuild the library (make -C hirl4ucav_amd/csrc)")
    L.hx_event_create.restype = ctypes.c_void_p
    L.hx_event_create.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    _lib = L
    return L


def register(name, argtypes):
    """Used by the other binding modules to add their entry points."""
    _SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int


def call(name, *args):
    L = load()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise HxError(f"{name} failed ({rc}): {L.hx_last_error().decode()}")


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream
