"""Build libyacht_hip.so (and the drop-in train executable) in-tree with hipcc for gfx950.

The shared library is the product: HIP kernels + the C ABI of include/yacht_hip.h.
Everything lands in yacht_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libyacht_hip.so")
EXE_PATH = os.path.join(LIB_DIR, "run_yacht_train_core")

LIB_SOURCES = ["yh_api.hip", "yh_build.hip", "yh_sort.hip", "yh_query.hip", "yh_batch.hip", "yh_pairwise.hip", "yh_sketch.hip", "yh_sigread.hip", "yh_pack.hip", "yh_hyp.cpp"]
EXE_SOURCES = ["train_core_main.cpp"]
HEADERS = ["yh_common.h", "yh_sigread.h", "yh_sort.h", os.path.join(REPO_DIR, "include", "yacht_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libyacht_hip.so cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd))


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    hipcc = _hipcc()
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
              "-Wno-unused-result", f"-I{os.path.join(REPO_DIR, 'include')}"]
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            if verbose:
                print("hipcc -c", os.path.basename(s), flush=True)
            _run([hipcc, *common, "-c", s, "-o", o])
    if force or _stale(LIB_PATH, objs):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH, "-lz", "-ldl"])
    return LIB_PATH


def build_variant(name: str, defines: dict, verbose: bool = False) -> str:
    """Tuning aid: build lib/libyacht_hip_<name>.so with -D overrides of the YH_* macros in
    csrc/yh_common.h.  Select it at run time with YACHT_HIP_LIB=<path>."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    out = os.path.join(LIB_DIR, f"libyacht_hip_{name}.so")
    flags = [f"-D{k}={v}" for k, v in defines.items()]
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    if verbose:
        print("hipcc variant", name, " ".join(flags), flush=True)
    _run([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
          f"-I{os.path.join(REPO_DIR, 'include')}", *flags, *srcs, "-o", out, "-lz", "-ldl"])
    return out


def build_exe(force: bool = False, verbose: bool = False) -> str:
    """run_yacht_train_core: same argv / files contract as the reference executable."""
    build_lib(force=force, verbose=verbose)
    srcs = [os.path.join(CSRC, s) for s in EXE_SOURCES]
    if force or _stale(EXE_PATH, srcs + [LIB_PATH, os.path.join(CSRC, "yh_sigread.h")]):
        _run([_hipcc(), "-O2", "-std=c++17", f"-I{os.path.join(REPO_DIR, 'include')}", f"-I{CSRC}", *srcs, "-o", EXE_PATH,
              f"-L{LIB_DIR}", "-lyacht_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    return EXE_PATH


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True)
    if os.path.exists(os.path.join(CSRC, EXE_SOURCES[0])):
        build_exe(force="--force" in sys.argv, verbose=True)
    print(LIB_PATH)
