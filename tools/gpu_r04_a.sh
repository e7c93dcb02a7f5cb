O=gpurun_out/r04_a; mkdir -p $O
{
echo "## window on"; timeout 120 tools/sweep_microbench
echo "## window off (S360_SWEEP_DBG=256)"; S360_SWEEP_DBG=256 timeout 120 tools/sweep_microbench
echo "## ts window on"; timeout 60 tools/sweep_microbench_ts ts 5040 1052 4
echo "## ts window off"; S360_SWEEP_DBG=256 timeout 60 tools/sweep_microbench_ts ts 5040 1052 4
} > $O/microbench.txt 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python -m pytest tests/test_gpu_fullsize.py -k "config1 or config2" -q > $O/pytest.log 2>&1
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_zz_variants.py -m gpu -x -q >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
cat $O/microbench.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_a/bench.json'))
print(d['value'], d.get('single_frame',{}).get('ms'), d.get('single_frame',{}).get('sweep'))
PY
