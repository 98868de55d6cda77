#!/usr/bin/env python3
"""Export the reference's shipped LASSO benchmark instances to .npz fixtures.

Source data (DATA files, not source code):
  /root/reference/benchmark/data/lasso_{tiny,small,medium}.jld2
loaded by the reference at benchmark/benchmarks.jl:30-45.  JLD2 is HDF5; the
HDF5 dataset dims are the reverse of Julia's (column-major) dims, so a raw
little-endian dump of /A with HDF5 shape {n, m} *is* the Julia m-by-n matrix
in column-major order.

Run once in the build container (needs /opt/conda/bin/h5dump and the reference
checkout); the resulting tests/golden/lasso_*.npz are committed so that nothing
at test time reads /root/reference.

    python tests/golden/make_benchmark_fixtures.py
"""
import os
import subprocess
import tempfile

import numpy as np

REF = "/root/reference/benchmark/data"
H5DUMP = "/opt/conda/bin/h5dump"
OUT = os.path.dirname(os.path.abspath(__file__))

SHAPES = {"lasso_tiny": (5, 10), "lasso_small": (50, 100), "lasso_medium": (500, 1000)}


def dump(path, dset, dtype):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run(
            [H5DUMP, "-d", "/" + dset, "-b", "LE", "-o", tmp.name, path],
            check=True,
            stdout=subprocess.DEVNULL,
        )
        return np.fromfile(tmp.name, dtype=dtype)


def main():
    for name, (m, n) in SHAPES.items():
        path = os.path.join(REF, name + ".jld2")
        A = dump(path, "A", "<f8")
        assert A.size == m * n
        A = A.reshape((m, n), order="F")  # Julia column-major m x n
        b = dump(path, "b", "<f8")
        xstar = dump(path, "xstar", "<f8")
        ystar = dump(path, "ystar", "<f8")
        lam = dump(path, "lambda", "<i8")
        assert b.shape == (m,) and xstar.shape == (n,) and ystar.shape == (m,) and lam.shape == (1,)
        # sanity: stored optimum satisfies the LASSO KKT conditions
        g = A.T @ (A @ xstar - b)
        assert np.max(np.abs(g)) <= float(lam[0]) * (1 + 1e-9), np.max(np.abs(g))
        assert np.allclose(ystar, b - A @ xstar, atol=1e-9)
        np.savez(
            os.path.join(OUT, name + ".npz"),
            A=np.asfortranarray(A),
            b=b,
            xstar=xstar,
            ystar=ystar,
            lam=np.float64(lam[0]),
        )
        print(name, A.shape, "lambda", int(lam[0]), "max|A'(Ax*-b)| =", np.max(np.abs(g)))


if __name__ == "__main__":
    main()
