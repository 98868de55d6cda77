"""Build libproxgrad_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch extension machinery:
the library is a plain C-ABI shared object that also serves non-Python hosts (Julia `ccall`)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libproxgrad_hip.so")
SOURCES = ["pg_core.hip", "pg_gemv.hip", "pg_gemv_tn2.hip", "pg_gemv_tn3.hip", "pg_gemv_tn4.hip", "pg_gemv_tn4d.hip", "pg_gemv_tn5.hip", "pg_gemv_dys.hip", "pg_vec.hip", "pg_iter.hip", "pg_persist.hip", "pg_lbfgs.hip", "pg_comm.hip"]
# every header under csrc/ and include/ (pg_gemv_tn4.hip is also compiled a second time, as pg_gemv_tn4d.hip's body)
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(CSRC, "pg_gemv_tn4.hip"), os.path.join(INCLUDE, "proxgrad_hip.h"), os.path.join(INCLUDE, "proxgrad_hip_ext.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path."""
    hipcc = _hipcc()
    objs = []
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE, "-I", CSRC,
             "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"] + os.environ.get("PG_EXTRA_HIPCC_FLAGS", "").split()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in jobs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"[build] {src} FAILED", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
