#!/bin/bash
# Headline A/B on ONE lease: the config #2 sweep with HEAD's kernels_trimaran.hip, with round 1's final
# kernels_trimaran.hip (tools/_var/kernels_trimaran_r01.hip = git show <round-1 verdict commit>^:...), and with HEAD's minus the
# DecideArgs kernel argument of the table-mode instantiations, three rounds of 200 steps each, alternating.
# build first:  python tools/r4/make_tlp_variants.py; python tools/variant.py build nodec kernels_trimaran.hip --from tools/_var/kernels_trimaran_nodec.hip
#               python tools/variant.py build r01tri kernels_trimaran.hip --from tools/_var/kernels_trimaran_r01.hip
out=${1:-gpurun_out/ab_headline.txt}
: > $out
for round in 1 2 3; do
  for v in base r01tri nodec; do
    python tools/variant.py run $v bench.py --sweep-only --steps 200 --warmup 20 --cpu-budget 0 --no-config5-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('round $round $v ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" | tee -a $out
  done
done
