cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for m in 0 1; do
  TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py ntu 128 2>/dev/null | tail -1 > $O/r04c_config3_ntu128_mode$m.json
  TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py syn 128 2>/dev/null | tail -1 > $O/r04c_config4_syn128_mode$m.json
  TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py ucla52 2>/dev/null | tail -1 > $O/r04c_config1_ucla52_mode$m.json
done
timeout -k 10 300 python tools/config_bench.py syn 256 2>/dev/null | tail -1 > $O/r04c_config4_syn256_mode0.json
timeout -k 10 300 python bench.py --config 4stream --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04c_config2_4stream.json
for f in $O/r04c_config*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print({k:d[k] for k in ('ms_per_step','value','clips_per_s','ms_per_step_split_bf16') if k in d})")"; done
