# A/B of SPX_OPT_NRT_PACKED_SCORE on one box: the NRT tests, then config #3 and config #5's share with the option on / off, alternating
timeout 600 python -m pytest tests/test_gpu_nrt.py -q -m gpu 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_exhaustive.py -q -m gpu -k "config3_every_cell and LeastAllocated" -s 2>&1 | grep -E "packed Score|passed|failed"
one() {
  timeout 300 python bench.py --workload $1 --steps $2 --warmup 3 --opt NRT_PACKED_SCORE=$3 2>/dev/null | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('$1 PACKED=$3 ms_per_step', round(d['ms_per_step'], 4))"
}
for r in 1 2; do for o in 1 0; do one config3 20 $o; done; done
for o in 1 0; do one config5_share 10 $o; done
for o in 1 0; do one config3_r8 10 $o; done
