"""Build libmri3d_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain C-ABI .so.

    python -m mri_epilepsy_diagnosis_amd.build [--force] [--jobs N]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libmri3d_hip.so")
SOURCES = ["api.hip", "conv_generic.hip", "conv_mfma.hip", "conv_march.hip", "conv_pointwise.hip", "upconv.hip", "sepconv.hip", "norm.hip", "resample.hip", "loss.hip", "elementwise.hip", "preprocess.hip", "surface.hip", "patches.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file additions.  conv_march.hip keeps its accumulators in a[0:95] BY NAME (inline asm): hipcc must never park a VGPR in an
# AGPR there (it would pick a0.. and be overwritten by the MFMAs) — its own VGPR -> AGPR spilling is switched off, and
# tests/test_march_codegen.py checks the generated code for stray AGPR writes and scratch use.
EXTRA_FLAGS = {"conv_march.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}


def _digest(path):
    h = hashlib.sha256()
    for dep in [path, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mfma_util.h"), os.path.join(HERE, "..", "include", "mri3d.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date", LIB)
    return LIB


def build_variant(name, extra_flags, verbose=True):
    """Tuning builds for tools/ only (A/B kernels, ablations): libmri3d_hip_<name>.so next to the product library, compiled with
    -DMRI3D_TUNING plus `extra_flags`.  The product package never loads these; tools/conv_bench.py --lib does."""
    obj_dir = os.path.join(CSRC, "build_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    lib = os.path.join(HERE, "libmri3d_hip_%s.so" % name)

    def one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        r = subprocess.run([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-DMRI3D_TUNING"] + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-3000:]))
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-3000:])
    if verbose:
        print("built", lib)
    return lib


def build_host_sanitizer(verbose=True):
    """Host-only AddressSanitizer + UBSan build of the library's host code (argument validation, plan / dispatch selection,
    workspace sizing) linked with tests/native/host_sanitizer_driver.cpp into ONE executable; the sanitizers instrument the host
    side only (-fno-gpu-sanitize) and the driver only takes paths that return before a kernel launch.  GPU ASan is unavailable on this
    pool.  Returns the executable's path (cached by a digest of every input)."""
    root = os.path.join(HERE, "..")
    driver = os.path.join(root, "tests", "native", "host_sanitizer_driver.cpp")
    out_dir = os.path.join(root, "tests", "native", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "host_sanitizer_driver")
    # the device side is compiled normally (-fno-gpu-sanitize: the fat binary must exist for the module constructor) and never runs
    flags = ["--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
             "-O1", "-g", "-std=c++17", "-w", "-I", os.path.join(root, "include")]
    h = hashlib.sha256(" ".join(flags).encode())
    inputs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mfma_util.h"), os.path.join(root, "include", "mri3d.h"), driver]
    for f in inputs:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = exe + ".sha"
    if os.path.exists(exe) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return exe

    def one(src):
        obj = os.path.join(out_dir, os.path.basename(src).rsplit(".", 1)[0] + ".host.o")
        cmd = [HIPCC] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("host sanitizer build failed for %s:\n%s" % (src, r.stderr[-3000:]))
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, [os.path.join(CSRC, s) for s in SOURCES] + [driver]))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-o", exe] + objs, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("host sanitizer link failed:\n%s" % r.stderr[-3000:])
    with open(stamp, "w") as f:
        f.write(h.hexdigest())
    if verbose:
        print("built", exe)
    return exe


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    elif "--host-sanitizer" in sys.argv:
        print(build_host_sanitizer())
    else:
        build(force="--force" in sys.argv)
