R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_breed.py tests/test_gpu_api.py tests/test_gpu_dropin.py tests/test_gpu_rollout.py -m gpu -q -x > $OUT/r03o_pytest.log 2>&1; tail -4 $OUT/r03o_pytest.log | cut -c1-250
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/r03o_bench.json 2> $OUT/r03o_bench.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03o_bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print([(x['selection'][:10], x['median']) for x in j['generation_ms_sharded']['runs']])
print(j['configs1']['generation_ms'], j['configs3']['generation_ms'], j['vis_ipynb_config']['generation_ms'])
PY
