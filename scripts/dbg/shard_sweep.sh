#!/bin/bash
# Work-distribution sweep of the interpreter at shard sizes: EVOGP_TC_DYNSHIFT x EVOGP_TC_STATIC, call time on N trees (shard_timeline.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in ${SIZES:-125000 100000 250000}; do
  for ds in "" 0 1 2 3; do
    for st in "" 50 70; do
      r=$(env ${ds:+EVOGP_TC_DYNSHIFT=$ds} ${st:+EVOGP_TC_STATIC=$st} python scripts/dbg/shard_timeline.py $n 2>/dev/null | grep "ms per call")
      echo "dynshift=${ds:-default} static=${st:-default}: $r"
    done
  done
done
