# Round-end check on a GPU box (run through gpurun): smoke, headline / adaptive / config-2 bench lines, rocprofv3 kernel stats.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err; cat gpurun_out/bench_headline.json | cut -c1-600
python bench.py --mode adaptive --no-cpu-baseline > gpurun_out/bench_headline_adaptive.json 2>/dev/null
python bench.py --workload config2 --no-cpu-baseline --steps 50 > gpurun_out/bench_config2.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_headline -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_headline/*/*_results.db > gpurun_out/prof_headline_summary.md 2>&1; head -30 gpurun_out/prof_headline_summary.md
