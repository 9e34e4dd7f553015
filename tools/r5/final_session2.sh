# after the narrow rank layout: the test files that touch the NRT Filter outside tests/test_gpu_nrt.py and tests/test_gpu_exhaustive.py (run
# before), the profiles whose stamp covers kernels_nrt_rank.hip, the bench lines
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/lines
timeout 500 python -m pytest tests/test_gpu_delta.py tests/test_gpu_profile.py tests/test_gpu_commit.py tests/test_gpu_commit_full.py tests/test_gpu_multi.py tests/test_gpu_bench.py tests/test_gpu_load_c.py tests/test_gpu_decide.py tests/test_gpu_host_mirror.py -m gpu -q -x > $R/gpurun_out/gpu_suite2.log 2>&1
tail -2 $R/gpurun_out/gpu_suite2.log
bash tools/prof_all.sh config3 config3_r8 config5_share 2>&1 | grep -v simple_timer | tail -3
python tools/collect_profiles.py r05 2>&1 | tail -4
cd $R
for W in config3 config3_r8 config5_share; do
  timeout 200 python bench.py --workload $W 2>/dev/null | tail -1 > gpurun_out/lines/${W}_sweep_bench_line.json
  python -c "import json; d=json.load(open('gpurun_out/lines/${W}_sweep_bench_line.json')); print('$W', round(d['ms_per_step'],4), d['roofline'].get('frac'), d['roofline'].get('traffic'), (d.get('every_row') or {}).get('kernel_ms'))"
done
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/lines/config2_default_bench_line.json
python -c "import json; d=json.load(open('gpurun_out/lines/config2_default_bench_line.json')); print('default', d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['config5_leg'].items() if k in ('sweep_ms','sweep_every_row_ms','sweep_plus_argmax_ms','decide_ms','sequential_us_per_pod','load_c_ms')})"
timeout 400 python bench.py --workload config5 --gpus 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/lines/config5_full_bench_line.json
python -c "import json; d=json.load(open('gpurun_out/lines/config5_full_bench_line.json')); print('config5 whole', d['ms_per_step'], d['value'])"
mkdir -p gpurun_out/r05new; cp profiles/r05/* gpurun_out/r05new/
