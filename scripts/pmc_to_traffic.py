#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, rocpd .db files) into the record bench.py reads
from profiles/pmc_traffic.json:   pmc_to_traffic.py <workload key> <fetch.db> <write.db> <source note> [json path]
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): read bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 tallies the
128-byte requests of wide coalesced streams as 64 bytes), write bytes = WRITE_SIZE[KiB] * 1024.  The record is tied to the
sha256 of the sweep-kernel sources, so bench.py stops quoting it as soon as they change."""
import importlib.util
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {"gemv_tn": ("gemv_tn_kernel", "gemv_tnm_kernel", "gemv_tnw_kernel", "gemv_tnc_kernel", "gemv_tnt_kernel"), "gemv_n_partial": ("gemv_n_partial_kernel",),
          "gemv_t": ("gemv_t_kernel",)}


def counter(db, cname):
    con = sqlite3.connect(db)
    rows = con.execute("select name, avg(counter_value), count(*) from pmc_events where counter_name = ? group by name",
                       (cname,)).fetchall()
    out = {}
    for name, avg, cnt in rows:
        for key, prefixes in GROUPS.items():
            if any(p + "<" in name for p in prefixes):
                tot, n = out.get(key, (0.0, 0))
                out[key] = (tot + avg * cnt, n + cnt)
    return {k: t / n for k, (t, n) in out.items()}


def main():
    key, fetch_db, write_db, note = sys.argv[1:5]
    path = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fetch, write = counter(fetch_db, "FETCH_SIZE"), counter(write_db, "WRITE_SIZE")
    rec = {"workload": key, "source": note, "units": "bytes per launch",
           "correction": "read = 2 * FETCH_SIZE[KiB] * 1024 (gfx950: 128-B requests tallied as 64 B for wide coalesced streams, "
                         "MI355X_MICROARCH.md HBM section), write = WRITE_SIZE[KiB] * 1024",
           "kernel_source_sha256": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        rd, wr = 2.0 * fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        rec["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = rec
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
