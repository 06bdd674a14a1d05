#!/bin/bash
# repeated A/B of launch parameters given as words "VAR=val[,VAR=val]"; three runs each, launch_ms of the default division
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "$@"; do
  printf "%s " "$cfg"
  for i in 1 2 3; do env $(echo $cfg | tr ',' ' ') python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['launch_ms'], end=' ')"; done
  echo
done
