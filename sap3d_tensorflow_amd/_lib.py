"""ctypes binding of libp3dhip.so (include/p3d_hip.h).  There is no CPU fallback: a missing or
unloadable library, or a box without a HIP device, raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# P3D_LIB: another build of the same library (same-box A/Bs against an older round's, tools/ab/); every declared symbol must
# still bind, so it cannot be anything else
LIB_PATH = os.environ.get("P3D_LIB") or os.path.join(_HERE, "libp3dhip.so")

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
_dp = C.POINTER(C.c_double)

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
    "p3d_block_backward": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int64, _fp, C.c_int64, _fp]),
    "p3d_set_pointwise_fp16": (C.c_int, [C.c_void_p, C.c_int]),
    "p3d_set_bn_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "p3d_set_attention_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "p3d_debug_dirty_counters": (C.c_int64, []),
    "p3d_debug_force_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "p3d_debug_schedule": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_char_p, C.c_int64, _i64p]),
    "p3d_debug_decision_count": (C.c_int, [C.c_void_p]),
    "p3d_debug_decision_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _i64p]),
    "p3d_debug_decision_get": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, C.c_int64]),
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
    "p3d_debug_stem_wgrad_through_bn": (C.c_int, [C.c_int, _fp, C.POINTER(C.c_int64), _fp, _fp, _fp, _fp, C.c_int, _fp, _fp]),
    "p3d_debug_install_abort_trace": (C.c_int, []),
    "p3d_device_count": (C.c_int, []),
    "p3d_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "p3d_debug_bucket_audit": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_int64, _i64p, _i64p, C.POINTER(C.c_int32), C.c_int,
                                         _i64p, _i64p]),
    "p3d_op_conv3d": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_conv3d_backprop_input": (C.c_int, [C.c_int, _fp, _fp, _i64p, _ip, _i64p, _fp]),
    "p3d_op_conv3d_backprop_filter": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_conv3d_transpose": (C.c_int, [C.c_int, _fp, _i64p, _fp, _i64p, _ip, _fp, _fp]),
    "p3d_op_max_pool3d": (C.c_int, [C.c_int, _fp, _i64p, _ip, _ip, _fp]),
    "p3d_op_max_pool3d_grad": (C.c_int, [C.c_int, _fp, _i64p, _ip, _ip, _fp, _fp]),
    "p3d_op_bias_add_grad": (C.c_int, [C.c_int, _fp, C.c_int64, C.c_int, _fp]),
    "p3d_op_attention_core": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "p3d_metric_cc": (C.c_int, [C.c_int, _fp, _fp, C.c_int, C.c_int, _dp]),
    "p3d_metric_sim": (C.c_int, [C.c_int, _fp, _fp, C.c_int, C.c_int, _dp]),
    "p3d_metric_nss": (C.c_int, [C.c_int, _fp, _fp, C.c_int, C.c_int, _dp]),
    "p3d_metric_auc_judd": (C.c_int, [C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, _dp]),
    "p3d_metric_auc_borji": (C.c_int, [C.c_int, _fp, _fp, _ip, C.c_int, C.c_int, C.c_int, C.c_double, _dp]),
    "p3d_mapf_frames": (C.c_int, [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp]),
    "p3d_mapf_density": (C.c_int, [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "p3d_crc32c": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "p3d_shutdown": (C.c_int, []),
}


def _torch_lib_dir():
    """Directory of the ROCm libraries a PyTorch-ROCm wheel bundles (torch/lib), or None.  Does not import torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        return None
    if spec is None or not spec.origin:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def _one_rocm_runtime_per_process():
    """One HIP runtime per process, whatever the import order.  PyTorch-ROCm wheels bundle their own libamdhip64 /
    libhsa-runtime64 / librccl (torch/lib) under the same sonames as /opt/rocm's.  If libp3dhip.so were loaded first it
    would bind to /opt/rocm's copies, a later `import torch` would map the bundled ones next to them: two HIP runtimes
    in one process (tools/dupe_probe.py).  So when such a wheel is installed and torch is not loaded yet, its bundled runtime is mapped
    here BY PATH, without importing torch (a first `import torch` costs a minute or two on a fresh box): libp3dhip then
    binds to it by soname, and so does torch whenever it is imported.  With torch already imported nothing is needed."""
    import sys
    if "torch" in sys.modules:
        return
    if mapped_rocm_runtimes():
        return          # a runtime is already mapped (e.g. rocprofv3 preloads /opt/rocm's): bind to that one
    d = _torch_lib_dir()
    if d is None:
        return
    for name in ("libhsa-runtime64.so", "libamdhip64.so", "librccl.so"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            C.CDLL(p)        # RTLD_LOCAL: see lib()


def mapped_rocm_runtimes():
    """{library stem: sorted list of distinct files mapped into this process} for the HIP / HSA / RCCL runtimes."""
    import re
    found = {}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*/(libamdhip64|libhsa-runtime64|librccl)\.so[^\s/]*)", line)
                if m:
                    found.setdefault(m.group(2), set()).add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return dict((k, sorted(v)) for k, v in found.items())


def _refuse_two_runtimes():
    dup = dict((k, v) for k, v in mapped_rocm_runtimes().items() if len(v) > 1)
    if dup:
        raise P3dError("two copies of a ROCm runtime library are mapped into this process (%s): it would crash at exit. "
                       "Load sap3d_tensorflow_amd (or torch) before anything else that pulls in a HIP runtime; see "
                       "INTEGRATION.md, 'One HIP runtime per process'." % dup)


_sessions = None


def register_session(s):
    """Sessions are closed, and the library's process-wide device resources released, from a Python atexit hook -- i.e.
    before interpreter finalisation and before any C++ static destructor -- so teardown never depends on the order in
    which libraries were loaded."""
    global _sessions
    if _sessions is None:
        import weakref
        _sessions = weakref.WeakSet()
    _sessions.add(s)


def _shutdown():
    if _sessions is not None:
        for s in list(_sessions):
            try:
                s.close()
            except Exception:
                pass
    if _lib is not None:
        try:
            _lib.p3d_shutdown()
        except Exception:
            pass


def lib():
    """Load libp3dhip.so once.  Raises if it has not been built (python -m sap3d_tensorflow_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P3dError("libp3dhip.so is not built (%s); run `python sap3d_tensorflow_amd/build.py`. "
                           "There is no CPU fallback." % LIB_PATH)
        _one_rocm_runtime_per_process()
        # RTLD_LOCAL (ctypes' default), never RTLD_GLOBAL: with its symbols in the global scope a later `import torch`
        # binds some of torch's C++ runtime symbols to this library's copies and the process dies at interpreter exit
        # (`double free or corruption`), even with a single HIP runtime mapped (tests/test_load_order.py)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _refuse_two_runtimes()
        _lib = l
        import atexit
        atexit.register(_shutdown)
    return _lib


def check(rc):
    if rc != 0:
        raise P3dError(lib().p3d_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(_fp) if a is not None else None
