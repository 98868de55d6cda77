"""TEST INFRASTRUCTURE: loader of oracle/csrc/cpu_twin.c, the C / OpenMP restatement of the FastForwardBackward fixed-step
iteration (benchmark/benchmarks.jl:11-17 + fast_forward_backward.jl:131-142, unfused, GEMVs threaded over all cores) that
bench.py times as its CPU leg.  Only tests/, __graft_entry__ and bench.py's cpu_baseline may use it.

The shared object is compiled with -march=native, so it is keyed by the host's CPU (model + flags): a library built in
one container is not loaded on a different host -- it is rebuilt there (gcc is part of the image)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cpu_twin.c")
OUT_DIR = os.path.join(HERE, "_build")


def _host_key():
    model, flags = "", ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            if ln.startswith("flags") and not flags:
                flags = ln.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return hashlib.sha256((model + "|" + flags + "|" + open(SRC).read()).encode()).hexdigest()[:12]


def lib_path():
    return os.path.join(OUT_DIR, "libcpu_twin_%s.so" % _host_key())


def build(force=False):
    """gcc -O3 -march=native -fopenmp -> oracle/_build/libcpu_twin_<host key>.so; returns the path"""
    path = lib_path()
    if force or not os.path.exists(path):
        os.makedirs(OUT_DIR, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", SRC, "-o", tmp, "-lm"], check=True)
        os.replace(tmp, path)
    # builds of other hosts / of earlier versions of the source are dead weight (git-ignored, but pushed to the GPU box every run)
    for name in os.listdir(OUT_DIR):
        if name.startswith("libcpu_twin_") and name.endswith(".so") and os.path.join(OUT_DIR, name) != path:
            try:
                os.remove(os.path.join(OUT_DIR, name))
            except OSError:
                pass
    return path


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.cpu_twin_ffb.restype = C.c_float
        lib.cpu_twin_ffb.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_double)]
        lib.cpu_twin_read_pass.restype = C.c_double
        lib.cpu_twin_read_pass.argtypes = [C.c_void_p, C.c_long, C.c_int, C.POINTER(C.c_double)]
        lib.cpu_twin_read_pass_seq.restype = C.c_double
        lib.cpu_twin_read_pass_seq.argtypes = [C.c_void_p, C.c_long, C.c_int, C.POINTER(C.c_double)]
        lib.cpu_twin_first_touch.restype = None
        lib.cpu_twin_first_touch.argtypes = [C.c_void_p, C.c_long]
        lib.cpu_twin_threads.restype = C.c_int
        lib.cpu_twin_set_threads.argtypes = [C.c_int]
        _lib = lib
    return _lib


def ffb(A, b, lam, Lf, steps, threads=None):
    """init + `steps` fixed-step FastForwardBackward iterations (x0 = 0) on a column-major Float32 matrix.
    Returns (z, f_x, seconds of the `steps` iterations, threads used)."""
    lib = load()
    assert A.dtype == np.float32 and A.flags.f_contiguous and b.dtype == np.float32
    m, n = A.shape
    if threads is not None:
        lib.cpu_twin_set_threads(int(threads))
    z = np.empty(n, np.float32)
    sec = C.c_double()
    fx = lib.cpu_twin_ffb(A.ctypes.data, m, n, m, b.ctypes.data, float(lam), float(Lf), int(steps), z.ctypes.data, C.byref(sec))
    return z, float(fx), float(sec.value), int(lib.cpu_twin_threads())


def read_gbps(A, threads=None, min_seconds=1.0):
    """GB/s this host reads the array `A` at with `threads` OpenMP threads: the better of two access patterns (eight interleaved
    streams per thread; one contiguous stream per thread, gemv_t's own), each repeated until `min_seconds` / 2 have gone by"""
    lib = load()
    if threads is not None:
        lib.cpu_twin_set_threads(int(threads))
    best = 0.0
    for fn in (lib.cpu_twin_read_pass, lib.cpu_twin_read_pass_seq):
        sec = C.c_double()
        fn(A.ctypes.data, A.size, 1, C.byref(sec))  # sizes the run (and touches the pages)
        reps = int(max(1, min(50, 0.5 * min_seconds / max(sec.value, 1e-4))))
        fn(A.ctypes.data, A.size, reps, C.byref(sec))
        best = max(best, A.nbytes * reps / sec.value / 1e9)
    return best


def first_touch(A, threads=None):
    """touch the pages of the (uninitialised) array `A` from the threads that will later stream them (cpu_twin_first_touch)"""
    lib = load()
    if threads is not None:
        lib.cpu_twin_set_threads(int(threads))
    lib.cpu_twin_first_touch(A.ctypes.data, A.size)
    return A
