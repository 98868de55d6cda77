#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5v; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panocplus or newton_family or image_slab" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for sp in 1 0; do for r in 1 2; do timeout 600 python scripts/bench_panoc.py --algo panocplus --speculate $sp --steps 23 --warmup 0 > $O/panocplus_sp${sp}_r$r.json 2>/dev/null; done; done
timeout 600 python scripts/bench_panoc.py --algo panocplus --steps 60 --warmup 3 > $O/panocplus_60.json 2>/dev/null
timeout 600 python tests/tools/fuzz_newton.py 600 900000 2>&1 | grep -v amdgpu.ids | tail -3 > $O/fuzz_newton.log
timeout 900 python tests/tools/fuzz_newton.py 400 910000 tall 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/fuzz_newton.log
tail -12 $O/pytest.log; for f in $O/panocplus*.json; do echo $f; cut -c1-330 $f; echo; done; cat $O/fuzz_newton.log
