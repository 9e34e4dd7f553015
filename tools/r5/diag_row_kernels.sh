# where do the per-row kernels of the full profile (k_net_cls, k_alloc_masked) lose their time?  diagnostic variants built with
# tools/variant.py from scratch copies (table reads wrapped into 1 KB / status reads removed — wrong results, timing only), timed by
# difference of evaluations (tools/r5/time_row_kernels.py; rocprofv3 around tools/variant.py hung on this pool — not used here)
for V in "$@"; do
  echo "== $V"
  timeout 60 python tools/variant.py run $V tools/r5/time_row_kernels.py 2>&1 | tail -1
done
