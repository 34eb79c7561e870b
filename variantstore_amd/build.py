"""Build recipe for the native parts (hipcc, gfx950 only).

`python -m variantstore_amd.build` compiles

  variantstore_amd/lib/libvariantstore_hip.so   C-ABI engine + HIP kernels (include/variantstore_hip.h)
  variantstore_amd/bin/variantstore             the drop-in CLI (query / construct), linked against the engine

The shared objects are built in-tree so they travel with the source snapshot.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "variantstore_amd")
LIB = os.path.join(PKG, "lib", "libvariantstore_hip.so")
CLI = os.path.join(PKG, "bin", "variantstore")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources():
    """Every file under csrc/ (recursively: csrc/host/formats/*.hpp are included by the engine too) + the ABI header."""
    out = []
    for base, _dirs, files in os.walk(os.path.join(PKG, "csrc")):
        out += [os.path.join(base, f) for f in sorted(files)]
    out.append(os.path.join(ROOT, "include", "variantstore_hip.h"))
    return out


def build_engine(force=False, verbose=True):
    srcs = _sources()
    if not force and _newer(LIB, srcs):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", LIB, os.path.join(PKG, "csrc", "hip", "engine.hip"), "-lz"]
    if os.environ.get("VS_BUILD_TUNING") == "1":   # ablation / occupancy switches of k_fill_carriers (tools/exp_fill.py)
        cmd.insert(1, "-DVS_TUNING")
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=ROOT)
    return LIB


def build_cli(force=False, verbose=True):
    src = os.path.join(PKG, "csrc", "cli", "variantstore.cpp")
    if not os.path.exists(src):
        return None
    if not force and _newer(CLI, _sources() + [LIB]):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI, src, "-I", os.path.join(ROOT, "include"),
           "-L", os.path.dirname(LIB), "-lvariantstore_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=ROOT)
    return CLI


def build_all(force=False, verbose=True):
    build_engine(force, verbose)
    build_cli(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
