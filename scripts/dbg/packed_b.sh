#!/bin/bash
# trees per wave of the packed program compiler (EVOGP_TC_PACKED = 16 / 32 / 64) at 1 M and 250 k trees, per call and by stage (scripts/pool_sweep.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do for b in 16 32 64; do echo "B=$b $(EVOGP_TC_PACKED=$b python scripts/pool_sweep.py 2>&1 | grep '250000 trees\| 1000000 trees' | tr '\n' ';' | cut -c1-400)"; done; done
