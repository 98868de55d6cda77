#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
if 'error' in d: print('ERROR', str(d['error'])[:500]); sys.exit()
r0=d['steps'][0]
print('$1', 'agree', d['ranks_agree_bitwise'], [(r['k'], round(r['gamma'],6), round(r['gamma_oracle'],6), '%.2e'%r['dz'], r['a_passes'], r['flags']) for r in r0])"; }
A="--m 23891 --n 3 --ranks 8 --steps 8 --adaptive --g box"
python tests/tools/row_team.py $A 2>/dev/null | show default
PG_TUNE=1 PG_TNP_W=4 PG_TNP_C=2 PG_TNP_LAG=2 PG_TNP_LAGR=0 PG_TNP_PF=2 PG_TNP_WGS=3 python tests/tools/row_team.py $A 2>/dev/null | show r4geom
python tests/tools/row_team.py $A --no-team 2>/dev/null | show noteam
python tests/tools/row_team.py --m 23891 --n 3 --ranks 8 --steps 8 --g box 2>/dev/null | show fixed_default
python tests/tools/row_team.py $A --dtype f64 2>/dev/null | show f64_default
bash scripts/r5_gpu_m.sh
