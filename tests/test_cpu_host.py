"""CPU-only tests: the C-ABI library loads and exports every symbol include/proxgrad_hip.h declares, host-side
logic (sequences, sharding partition), loud failure without a GPU, and the world_size-2 gloo path."""
import itertools
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import proxgrad_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build()
    from proximalalgorithms.jl_amd import _lib

    return _lib


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "proxgrad_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:pg_status|int32_t|const char\*)\s+(pg_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 45
    handle = lib.load()
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} is declared in proxgrad_hip.h but not exported"
    assert declared == set(lib.exported_symbols()), declared ^ set(lib.exported_symbols())
    assert handle.pg_abi_version() == 1


def test_no_cpu_fallback(lib):
    import torch

    import proximalalgorithms.jl_amd as pa

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pa.ProxGradError):
        pa.LeastSquares(np.eye(3), np.ones(3))
    with pytest.raises(pa.ProxGradError):
        pa.HIPVector.zeros(4, np.float32)
    # argument validation happens before any device work
    import ctypes as C

    assert lib.load().pg_mat_create(None, 0, 1, 1, C.byref(C.c_void_p())) != 0
    assert b"null" in lib.load().pg_last_error()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "proximalalgorithms.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".jl")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_sequences_match_oracle(lib, dtype):
    import proximalalgorithms.jl_amd as pa

    for a, b in itertools.islice(zip(pa.FixedNesterovSequence(dtype), o.fixed_nesterov_sequence(dtype)), 50):
        assert a == b and a.dtype == dtype
    for a, b in itertools.islice(zip(pa.SimpleNesterovSequence(dtype), o.simple_nesterov_sequence(dtype)), 50):
        assert a == b
    m, s = dtype(1.0), dtype(0.1)
    assert next(iter(pa.ConstantNesterovSequence(m, s))) == next(o.constant_nesterov_sequence(m, s))
    pa_seq, o_seq = pa.AdaptiveNesterovSequence(dtype(0.3)), o.AdaptiveNesterovSequence(dtype(0.3))
    for k in range(30):
        g = dtype(0.5 + 0.01 * k)
        assert pa.next_(pa_seq, g) == o_seq.next(g)


def test_shard_rows_partition():
    import proximalalgorithms.jl_amd as pa

    for m, w in [(16384, 8), (131072, 8), (10, 3), (7, 8), (0, 2)]:
        parts = [pa.shard_rows(m, w, r) for r in range(w)]
        assert parts[0][0] == 0
        for (o0, c0), (o1, _) in zip(parts, parts[1:]):
            assert o0 + c0 == o1
        assert parts[-1][0] + parts[-1][1] == m
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
        assert parts == [pa.shard_cols(m, w, r) for r in range(w)]  # column shards use the same balanced partition


def test_world_size_2_gloo_sharded_path():
    """Two CPU processes over gloo: row-shard A, all-reduce [grad ; f] with the package's collective helper,
    drive a full FFB solve with the sharded operator, compare with the unsharded oracle."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29617", script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_SHARDED_OK" in out.stdout


def test_bench_cli_parses_without_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_bench_self_launches_ranks_as_a_child(monkeypatch, capsys):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE): before anything touches the GPU the script starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (subprocess, never exec), relays the
    child's single JSON line and returns its exit code."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_run(cmd, stdout=None, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done(seen["rc"], seen["out"])

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4})
    seen["rc"], seen["out"] = 0, ("NCCL banner\n" + line + "\n").encode()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert capsys.readouterr().out.strip() == line  # exactly the child's JSON line
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a failing child: its exit code comes back, nothing is printed
    seen["rc"], seen["out"] = 3, b"Traceback ...\n"
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3 and capsys.readouterr().out == ""
    # a child that exits 0 without a line is an error, not a silent success
    seen["rc"], seen["out"] = 0, b""
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 1


def test_pmc_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic is only quoted while the sweep-kernel sources hash to what the rocprofv3 --pmc passes were taken
    on; otherwise bench.py reports traffic = null with traffic_stale = true."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h = bench.kernel_source_hash()
    assert len(h) == 64 and h == bench.kernel_source_hash()
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert "kernel_source_sha256" in committed["headline"]
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    rec = {"headline": {"source": "profiles/x.md", "kernel_source_sha256": h, "kernels": {"gemv_tn": {"hbm_bytes": 123.0}}}}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    assert bench.pmc_traffic("headline", "gemv_tn") == (123.0, "profiles/x.md", False)
    assert bench.pmc_traffic("config2", "gemv_tn") == (None, None, False)  # never measured there
    rec["headline"]["kernel_source_sha256"] = "0" * 64
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    assert bench.pmc_traffic("headline", "gemv_tn") == (None, "profiles/x.md", True)


def test_iteration_tools():
    """test/utilities/test_iteration_tools.jl (host logic, no device)"""
    import itertools
    import time

    from proximalalgorithms.jl_amd import iteration_tools as IterationTools

    rng = np.random.default_rng(0)
    seq = list(rng.random(10))
    assert IterationTools.loop(seq) == seq[-1]  # :17-21
    fib = [0, 1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765]
    truncated = IterationTools.halt(fib, lambda x: x >= 1000)  # :23-52
    assert len(truncated) == len(fib) and IterationTools.loop(truncated) == 1597

    def fibonacci(s0, s1):
        while True:
            yield s0
            s0, s1 = s1, s0 + s1

    seen = []
    teed = IterationTools.tee(fibonacci(0, 1), seen.append)  # :54-64
    assert list(itertools.islice(teed, 10)) == fib[:10] == seen
    data = list(rng.standard_normal(147))
    sampled = IterationTools.sample(data, 10)  # :66-77
    assert len(sampled) == 15
    for k, x in enumerate(sampled):
        assert x == data[min(147, (k + 1) * 10) - 1]
    assert k == 14
    timed = IterationTools.stopwatch(seq[:4])  # :79-93
    assert len(timed) == 4
    for k, (t, x) in enumerate(timed):
        assert x == seq[k] and t >= k * 2e7 * 0.9
        time.sleep(0.02)
    with pytest.raises(TypeError):
        IterationTools.loop([])


def test_design_table_matches_the_committed_bench_lines():
    """DESIGN.md's "Measured (round 2)" block is generated from the bench lines under profiles/ (scripts/design_table.py):
    the committed block must be what the committed lines produce."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_table.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    b, e = "<!-- measured:begin -->\n", "<!-- measured:end -->\n"
    block = text[text.index(b) + len(b):text.index(e)]
    assert block.strip() == out.stdout.strip()
