#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['config']['also'])
r=[a for a in d['also'] if a['label']=='headline_row_block_n8'][0]
print(r['value'], r['ms_per_step'], r['roofline'])"
done
python bench.py --m 2048 --n 1048576 --steps 6 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | cut -c1-300
