#!/bin/bash
# A/B of the chunked two-stream pipeline on the headline workload
cd ${GRAFT_REPO_ROOT:-.}
for c in 1 2 4 8; do
  echo "== EVOGP_TC_CHUNKS=$c"
  EVOGP_TC_CHUNKS=$c timeout 300 python bench.py --steps 20 --warmup 3 --headline-only | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['stage_ms'])"
done
echo "== EVOGP_TC_CHUNKS=1 EVOGP_TC_WG=512"
EVOGP_TC_CHUNKS=1 EVOGP_TC_WG=512 timeout 300 python bench.py --steps 20 --warmup 3 --headline-only | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['stage_ms'])"
