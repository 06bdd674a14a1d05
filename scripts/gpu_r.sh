R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_api.py tests/test_gpu_breed.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/r03r_pytest.log 2>&1; tail -4 $OUT/r03r_pytest.log | cut -c1-250
timeout 900 python scripts/shard_model.py 2>&1 | grep trees > $OUT/r03r_shard_model.log; cat $OUT/r03r_shard_model.log
EVOGP_TC_FUNC_MASK=0 timeout 900 python scripts/shard_model.py 2>&1 | grep trees | sed 's/^/mask ignored: /' > $OUT/r03r_shard_model_nomask.log; cat $OUT/r03r_shard_model_nomask.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/r03r_bench.json 2> $OUT/r03r_bench.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03r_bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print(j['value'], j['ms_per_step'], j['roofline']['traffic'], j['roofline']['valu_issue']['valu_pipe_busy'])
print(j['shard_model']['ms'], j['configs1']['ms_per_step'], j['configs1']['generation_ms']['median'])
print([(x['selection'][:10], x['median']) for x in j['generation_ms_sharded']['runs']])
PY
