# usage: tail_exp.sh <global pop> "<dynshift static>" ...
P=$1; shift
for cfg in "$@"; do set -- $cfg
  r=$(EVOGP_TC_DYNSHIFT=$1 EVOGP_TC_STATIC=$2 timeout 120 python bench.py --steps 30 --warmup 5 --headline-only --no-cpu-baseline --global-pop $P 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['stage_ms'].items()})")
  echo "pop $P dynshift $1 static $2: $r"
done
