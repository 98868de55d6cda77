# Collect the round's measurements on a GPU box (run through gpurun); outputs land in gpurun_out/ and are copied
# into profiles/ by hand afterwards: bench lines (headline one / two sweeps, adaptive, config 2, column-shard shapes),
# rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, no other trace domains).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
python bench.py --mode adaptive --no-cpu-baseline > gpurun_out/bench_headline_adaptive.json 2>/dev/null
python bench.py --sweeps two --no-cpu-baseline > gpurun_out/bench_headline_two_sweeps.json 2>/dev/null
python bench.py --workload config2 --no-cpu-baseline --steps 50 > gpurun_out/bench_config2.json 2>/dev/null
for n in 524288 262144 131072; do python bench.py --m 16384 --n $n --force-comm --sharding cols --no-cpu-baseline > gpurun_out/bench_colshard_n$n.json 2>/dev/null; done
for m in 8192 4096 2048; do python bench.py --m $m --sweeps two --no-cpu-baseline > gpurun_out/bench_shard$m.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_headline -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
for d in prof_headline prof_fetch prof_write; do python scripts/rocpd_summary.py gpurun_out/$d/*/*_results.db > gpurun_out/$d.md 2>&1; done
python scripts/bench_panoc.py > gpurun_out/bench_panoc.json 2>/dev/null
python tests/tools/bench_dr.py > gpurun_out/bench_dr.json 2>/dev/null
for a in ffb ffb-generic; do python scripts/bench_panoc.py --algo $a 2>/dev/null; done > gpurun_out/bench_logistic_ffb.json
python scripts/bench_primal_dual.py > gpurun_out/bench_primal_dual.json 2>/dev/null
python tests/tools/bench_suite.py > gpurun_out/bench_suite.log 2>/dev/null
