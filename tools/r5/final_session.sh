# round 5's closing GPU session: the whole -m gpu suite, the profiles of the workloads whose kernels changed, the bench lines
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/lines
S=$SECONDS
timeout 1100 python -m pytest tests -m gpu -q -x > $R/gpurun_out/gpu_suite.log 2>&1
tail -3 $R/gpurun_out/gpu_suite.log; echo "suite seconds $((SECONDS-S))"
bash tools/prof_all.sh config3 config3_r8 config5_share config4 config2 2>&1 | grep -v simple_timer | tail -5
python tools/collect_profiles.py r05 2>&1 | tail -6
cd $R
for W in config3 config3_r8 config5_share config4 config3_most config3_balanced; do
  timeout 200 python bench.py --workload $W 2>/dev/null | tail -1 > gpurun_out/lines/${W}_sweep_bench_line.json
  python -c "import json; d=json.load(open('gpurun_out/lines/${W}_sweep_bench_line.json')); print('$W', round(d['ms_per_step'],4), d['roofline'].get('frac'), d['roofline'].get('traffic'), (d.get('every_row') or {}).get('ms'))"
done
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/lines/config2_default_bench_line.json
python -c "import json; d=json.load(open('gpurun_out/lines/config2_default_bench_line.json')); print('default', d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['config5_leg'].items() if k.endswith('_ms') or k.endswith('per_pod')})"
timeout 400 python bench.py --workload config5 --gpus 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/lines/config5_full_bench_line.json
python -c "import json; d=json.load(open('gpurun_out/lines/config5_full_bench_line.json')); print('config5 whole', d['ms_per_step'], d['value'])"
mkdir -p gpurun_out/r05new; cp profiles/r05/* gpurun_out/r05new/
