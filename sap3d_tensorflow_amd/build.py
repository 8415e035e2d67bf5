"""Builds libp3dhip.so (the HIP kernels + host runtime + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU.  Objects are cached by source mtime so that an unchanged
tree rebuilds in a fraction of a second."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libp3dhip.so")
SOURCES = ["conv_igemm2.hip", "conv_pointwise.hip", "conv_wgrad2.hip", "stem_wgrad.hip", "elementwise.hip", "bn_small.hip", "gn.hip", "cbam.hip", "head.hip", "attention.hip", "attention_flash.hip", "metrics.hip", "net.hip"]
HEADERS = [os.path.join(CSRC, n) for n in ("p3d_kernels.h", "det_reduce.h", "igemm_epilogue.h", "net_ops.inc", "net_gn.inc", "net_graphs.inc", "net_plan.inc", "net_sched.inc", "net_abi.inc")] + [ os.path.join(os.path.dirname(HERE), "include", "p3d_hip.h")]
# P3D_EXTRA_HIPCC_FLAGS: extra compile flags for A/B builds of the tuning macros (e.g. -DP3D_PF64=0), see tools/README.md
FLAGS = (["-DP3D_SETPRIO"] if os.environ.get("P3D_SETPRIO") else []) + os.environ.get("P3D_EXTRA_HIPCC_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libp3dhip cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_key():
    """The compile flags are part of an object's identity: an A/B build with -DSOMETHING must never be reused by the next
    plain build (objects used to be cached by source mtime alone)."""
    import hashlib
    return hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:16]


def build(verbose=False, force=False):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    keyfile = os.path.join(objdir, "flags.key")
    key = _flags_key()
    if not os.path.exists(keyfile) or open(keyfile).read().strip() != key:
        force = True            # objects built with other flags (or before the key existed) are not ours
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for msg in ex.map(run, jobs):
            if verbose and msg:
                print(msg)
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs +
            ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    with open(keyfile, "w") as f:      # written last: an interrupted build leaves the old key and is redone
        f.write(key + "\n")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
