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
# per-source additions.  corr3dmlp: the SLP vectoriser pairs the fmaf chains into v_pk_fma_f32, whose operands must be
# vector registers -- the 1024 wave-uniform weights then leave the scalar registers (512 VGPRs + scratch instead of 168)
EXTRA_FLAGS = {"corr3dmlp.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source into camliflow_amd/csrc/libcamli_hip.so.  Returns the path.
    One object per source (compiled in parallel, rebuilt only when the source or a header is newer), then one
    link: a kernel edit costs one file's compile time."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcamli_hip.so")
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR) if f.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > newest_header):
            return obj, None
        cmd = [hipcc] + compile_flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", "-I", HIP_DIR, "-I", INCLUDE_DIR, src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (res.stdout + res.stderr if res.returncode != 0 else None)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, sources()))
    errors = [err for _, err in results if err]
    if errors:
        sys.stderr.write("\n".join(errors))
        raise RuntimeError("hipcc failed building libcamli_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for obj, _ in results] + ["-o", LIB_PATH + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libcamli_hip.so")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
