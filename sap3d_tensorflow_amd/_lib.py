"""ctypes binding of libp3dhip.so (include/p3d_hip.h).  There is no CPU fallback: a missing or
unloadable library, or a box without a HIP device, raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libp3dhip.so")

P3D_COMM_ID_BYTES = 128


class P3dConfig(C.Structure):
    _fields_ = [("structure", C.c_int), ("batch", C.c_int), ("frames", C.c_int), ("height", C.c_int),
                ("width", C.c_int), ("base", C.c_int), ("blocks", C.c_int * 3), ("device", C.c_int),
                ("world_size", C.c_int), ("rank", C.c_int)]


class P3dOpTime(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kernel", C.c_char * 48), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("phase", C.c_int)]


class P3dError(RuntimeError):
    pass


_lib = None
_fp = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int)

# every symbol include/p3d_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "p3d_default_config": (None, [C.POINTER(P3dConfig)]),
    "p3d_create": (C.c_int, [C.POINTER(P3dConfig), C.POINTER(C.c_void_p)]),
    "p3d_destroy": (None, [C.c_void_p]),
    "p3d_last_error": (C.c_char_p, []),
    "p3d_num_params": (C.c_int, [C.c_void_p]),
    "p3d_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), _ip, _i64p, _ip]),
    "p3d_set_param": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.c_int64]),
    "p3d_get_param": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.c_int64]),
    "p3d_get_grad": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.c_int64]),
    "p3d_init_params": (C.c_int, [C.c_void_p, C.c_uint64]),
    "p3d_forward": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_float, C.c_uint64, _fp]),
    "p3d_predict_windows": (C.c_int, [C.c_void_p, _fp, _fp]),
    "p3d_block_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "p3d_block_forward": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int64, _fp, C.c_int64]),
    "p3d_set_pointwise_fp16": (C.c_int, [C.c_void_p, C.c_int]),
    "p3d_train_step": (C.c_int, [C.c_void_p, _fp, _fp, C.c_float, C.c_uint64, _fp]),
    "p3d_backward": (C.c_int, [C.c_void_p, _fp, _fp, C.c_float, C.c_uint64, _fp, _fp]),
    "p3d_set_adam": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]),
    "p3d_activation_info": (C.c_int, [C.c_void_p, C.c_char_p, _i64p]),
    "p3d_get_activation": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.c_int64]),
    "p3d_upload_inputs": (C.c_int, [C.c_void_p, _fp, _fp]),
    "p3d_train_step_device": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    "p3d_forward_device": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_uint64]),
    "p3d_last_loss": (C.c_int, [C.c_void_p, _fp]),
    "p3d_synchronize": (C.c_int, [C.c_void_p]),
    "p3d_profile_step": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.POINTER(P3dOpTime), C.c_int]),
    "p3d_comm_unique_id": (C.c_int, [C.c_void_p]),
    "p3d_comm_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "p3d_op_conv3d": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_conv3d_backprop_input": (C.c_int, [C.c_int, _fp, _fp, _i64p, _ip, _i64p, _fp]),
    "p3d_op_conv3d_backprop_filter": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_conv3d_transpose": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_max_pool3d": (C.c_int, [C.c_int, _fp, _i64p, _ip, _ip, _fp]),
    "p3d_op_max_pool3d_grad": (C.c_int, [C.c_int, _fp, _i64p, _ip, _ip, _fp, _fp]),
}


def _one_rocm_runtime_per_process():
    """Load order guard.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl (torch/lib).
    Loading libp3dhip.so BEFORE torch ends in `double free or corruption` at interpreter exit (tools/dupe_probe.py):
    first because two HIP runtimes get loaded (system + bundled), and -- even with a single runtime -- because
    libp3dhip's HIP fat-binary un-registration (an atexit handler hipcc emits) then runs after torch has torn the
    runtime down.  With torch loaded first both problems vanish: libp3dhip binds to the already loaded runtime by
    soname and un-registers before torch's teardown.  So if a torch installation is present it is imported here,
    before the library -- the order bench.py has at N > 1 anyway (torch.distributed comes first there)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("P3D_NO_TORCH_PRELOAD") == "1":
        return          # (the opt-out is for processes that are known never to import torch: bench.py at N = 1, smoke())
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:       # a broken torch installation must not take the HIP path down with it
        pass


def lib():
    """Load libp3dhip.so once.  Raises if it has not been built (python -m sap3d_tensorflow_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P3dError("libp3dhip.so is not built (%s); run `python sap3d_tensorflow_amd/build.py`. "
                           "There is no CPU fallback." % LIB_PATH)
        _one_rocm_runtime_per_process()
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise P3dError(lib().p3d_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(_fp) if a is not None else None
