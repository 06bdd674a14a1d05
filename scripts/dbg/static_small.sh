cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do for st in - 100 95; do for n in 64 1024 4096 16384 32768; do
  if [ $st = - ]; then echo "STATIC=default $(python scripts/dbg/shard_timeline.py $n 2>&1 | grep 'ms per call')"; else echo "STATIC=$st $(EVOGP_TC_STATIC=$st python scripts/dbg/shard_timeline.py $n 2>&1 | grep 'ms per call')"; fi
done; done; done | sort -s -k1,1 -k2,2n
