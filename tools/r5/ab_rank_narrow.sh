# A/B of SPX_OPT_NRT_RANK_NARROW on one box (the option is read at pod upload: bench.py sets it before loading)
timeout 300 python -m pytest tests/test_gpu_nrt.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_exhaustive.py -q -m gpu -x -k "config3_every_cell or config5_share or six_slots" 2>&1 | tail -3
one() {
  timeout 300 python bench.py --workload $1 --steps $2 --warmup 3 --opt NRT_RANK_NARROW=$3 2>/dev/null | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('$1 NARROW=$3 ms_per_step', round(d['ms_per_step'], 4))"
}
for r in 1 2; do for o in 1 0; do one config3 30 $o; done; done
for o in 1 0; do one config5_share 10 $o; done
for o in 1 0; do one config3_r8 10 $o; done
for o in 1 0; do one config3_most 10 $o; done
