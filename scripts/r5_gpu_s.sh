#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zerofpr or panoc or newton or image_slab or lbfgs" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for r in 1 2; do timeout 600 python scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 > $O/zerofpr_r$r.json 2>/dev/null; done
timeout 600 python scripts/bench_panoc.py --algo zerofpr --steps 60 --warmup 0 > $O/zerofpr_60.json 2>/dev/null
timeout 600 python scripts/bench_panoc.py --algo panocplus --steps 23 --warmup 0 > $O/panocplus.json 2>/dev/null
timeout 600 python scripts/bench_panoc.py --algo panoc --steps 23 --warmup 0 > $O/panoc_23.json 2>/dev/null
timeout 600 python tests/tools/fuzz_newton.py 600 880000 2>&1 | grep -v amdgpu.ids | tail -3 > $O/fuzz_newton.log
timeout 900 python tests/tools/fuzz_newton.py 300 890000 tall 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/fuzz_newton.log
tail -4 $O/pytest.log; for f in $O/*.json; do echo $f; cut -c1-330 $f; echo; done; cat $O/fuzz_newton.log
