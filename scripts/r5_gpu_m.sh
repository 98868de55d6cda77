#!/bin/bash
# rocprofv3 kernel stats of the row-team sweep (two ranks as contexts of ONE process: one program after `--`, no launcher)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5m; mkdir -p $O
for s in "2048 4096 1048576" "16384 32768 131072"; do set -- $s
  python3 tests/tools/row_team.py --bench --m $2 --n $3 --steps 20 --max-wgs -2 2>/dev/null | tail -1 > $O/row_team_bench_$1.json
  rocprofv3 --kernel-trace --stats -d $O/prof_rt_$1 -- python3 tests/tools/row_team.py --bench --m $2 --n $3 --steps 20 --max-wgs -2 > $O/prof_rt_$1.log 2>&1
  python scripts/rocpd_summary.py $O/prof_rt_$1/*/*_results.db > $O/prof_rt_$1.md 2>&1
done
head -12 $O/prof_rt_2048.md | cut -c1-250; cut -c1-400 $O/row_team_bench_2048.json
