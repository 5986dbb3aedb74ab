"""Build recipe for libcamli_hip.so (hipcc cross-compiles gfx950 without a GPU present)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_DIR = os.path.join(HERE, "hip")
INCLUDE_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB_PATH = os.path.join(HERE, "libcamli_hip.so")

# -ffp-contract=off: KNN / FPS distances are specified UNFUSED (oracle/camli_oracle.c); dot products
# that may fuse say so explicitly with __builtin_fmaf.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def sources():
    return sorted(os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source into camliflow_amd/csrc/libcamli_hip.so.  Returns the path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcamli_hip.so")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", HIP_DIR, "-I", INCLUDE_DIR] + sources() + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libcamli_hip.so")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
