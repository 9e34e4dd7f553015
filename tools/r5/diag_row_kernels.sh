# where do the per-row kernels of the full profile (k_alloc_masked, k_net_cls) lose their time?  diagnostic variants built with
# tools/variant.py (tools/_var/*.hip: table reads wrapped into 1 KB / status reads removed — wrong results, timing only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for V in "$@"; do
  rm -rf /tmp/dv_$V
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dv_$V -o t -- python $R/tools/variant.py run $V $R/bench.py --workload config5_share --cpu-budget 0 --sweep-only --no-every-row --steps 5 --warmup 1 > /tmp/dv_$V.log 2>&1
  echo "== $V"; grep -E "k_alloc_masked|k_net_cls" /tmp/dv_$V/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
