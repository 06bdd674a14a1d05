R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_breed.py tests/test_gpu_parity.py -m gpu -q -x -k "tournament or sharded_native or trigonometric or sqrt_exp_log or golden_battery or each_function" > $OUT/r03n_pytest.log 2>&1; tail -4 $OUT/r03n_pytest.log | cut -c1-250
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/r03n_bench.json 2> $OUT/r03n_bench.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03n_bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print([(x['selection'][:10], x['median']) for x in j['generation_ms_sharded']['runs']])
PY
