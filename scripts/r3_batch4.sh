set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b4
mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/tile_pattern.hip -o /tmp/tile_pattern && /tmp/tile_pattern > $O/tile_pattern_x.log 2>&1
