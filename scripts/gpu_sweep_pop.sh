#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "$@"; do
  printf "%s " "$cfg"
  for p in 125000 1000000; do env $(echo $cfg | tr ',' ' ') python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pop-per-gpu $p | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%d: %.4f ms %.3e;' % (d['config']['pop_per_gpu'], d['roofline']['launch_ms'], d['value']), end=' ')"; done
  echo
done
