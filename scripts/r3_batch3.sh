set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b3
mkdir -p $O
python scripts/r3_mid_sweep.py ab > $O/mid_ab.log 2>&1
python scripts/r3_mid_sweep.py check > $O/mid_check.log 2>&1
tail -3 $O/mid_check.log
