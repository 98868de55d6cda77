import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the suite drives kernel geometry through the PG_* tuning variables in places: the library only looks at them in a process
# started with PG_TUNE set (csrc/pg_internal.h::pg_tuning_enabled)
os.environ.setdefault("PG_TUNE", "1")
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "rates: measured rates against soft thresholds (reported, not asserted, except hard floors); always also `gpu`")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def bench_default_line():
    """The driver's N = 1 command (`python bench.py --gpus 1 --steps 6 --warmup 2`), run ONCE per session: the parity suite checks
    the STRUCTURE of its line (tests/test_gpu_parity.py), tests/test_gpu_rates.py reads the measured rates off the same line.
    Returns a function: line() -> dict (cached), line(fresh=True) -> a new run."""
    import json
    import subprocess

    cache = {}

    def run():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                              "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        lines = out.stdout.splitlines()
        assert len(lines) == 1, lines
        return json.loads(lines[0])

    def line(fresh=False):
        if fresh or "d" not in cache:
            cache["d"] = run()
        return cache["d"]

    return line
