"""Build libcapdec.so (gfx950 HIP kernels + C ABI) in-tree with hipcc, and libcapdec_tune.so, the same sources with
the kernel-selection overrides of include/capdec_tuning.h compiled in (tests and benchmark tools only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libcapdec.so")
LIB_TUNE = os.path.join(HERE, "libcapdec_tune.so")
SOURCES = ["gemm.hip", "ops.hip", "lstm.hip", "engine.hip", "train_ops.hip", "decode_fused.hip"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def _build_one(lib, objdir, defines, verbose):
    objs = []
    procs = []
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function"] + \
            defines + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp"] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(lib + ".tmp", lib)
    return lib


def build(force=False, verbose=False):
    """Compile every kernel source for gfx950 and link the product library and the tuning build."""
    if force or stale(LIB):
        _build_one(LIB, os.path.join(CSRC, "build"), [], verbose)
    if force or stale(LIB_TUNE):
        _build_one(LIB_TUNE, os.path.join(CSRC, "build", "tune"), ["-DCAPDEC_TUNING_HOOKS"], verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
