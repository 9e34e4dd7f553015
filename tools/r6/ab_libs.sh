# A/B of library builds on ONE box: tools/r6/ab_libs.sh <workload> <variant letters...>  (scheduler-plugins_amd/libspx_<X>.so, built beforehand)
W=$1; shift
for rep in 1 2; do for v in "$@"; do
  cp scheduler-plugins_amd/libspx_$v.so scheduler-plugins_amd/libspx.so
  python bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$W', round(d['ms_per_step'],4), round(d['roofline'].get('kernel_ms'),4))"
done; done
