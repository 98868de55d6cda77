#!/bin/bash
# round 5 differential campaigns on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5k; mkdir -p $O
L=$O/fuzz_campaigns.log
{
echo "# Round 5 differential campaigns on the final code (row-team sweep: one / two / four waves per column, lag tiles in registers; ZeroFPR with two trial points per sweep; resume fixes)"
echo "## python tests/tools/fuzz_row_team.py 400 800000    (row teams on one GPU, ranks as contexts of one process: 2..8 ranks, blocks of 1..16384 rows incl. ragged, f32 / f64, FB / FFB, fixed / adaptive, batched loop, second problem)"
timeout 1500 python tests/tools/fuzz_row_team.py 400 800000 2>&1 | grep -v amdgpu.ids | tail -8
echo "## python tests/tools/fuzz_bench_ranks.py 60 810000    (the production path: one PROCESS per rank through bench.py, 2..4 ranks, rows / row teams / columns, against one rank)"
timeout 1500 python tests/tools/fuzz_bench_ranks.py 60 810000 2>&1 | grep -v amdgpu.ids | tail -8
echo "## python tests/tools/fuzz_newton.py 1500 820000    (PANOC / ZeroFPR / PANOCplus / DouglasRachford against the CPU restatement)"
timeout 900 python tests/tools/fuzz_newton.py 1500 820000 2>&1 | grep -v amdgpu.ids | tail -5
echo "## python tests/tools/fuzz_newton.py 600 830000 tall    (column lengths 600 .. 140000: ZeroFPR's two-point sweep is on for 8193 .. 16384 Float32 / 4097 .. 8192 Float64 rows)"
timeout 1500 python tests/tools/fuzz_newton.py 600 830000 tall 2>&1 | grep -v amdgpu.ids | tail -5
echo "## python tests/tools/fuzz_resume.py 1500 840000    (checkpoint / resume, bit for bit)"
timeout 900 python tests/tools/fuzz_resume.py 1500 840000 2>&1 | grep -v amdgpu.ids | tail -5
echo "## python tests/tools/fuzz_parity.py 2000 850000 ; 500 860000 tall"
timeout 900 python tests/tools/fuzz_parity.py 2000 850000 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python tests/tools/fuzz_parity.py 500 860000 tall 2>&1 | grep -v amdgpu.ids | tail -4
} > $L 2>&1
cat $L
