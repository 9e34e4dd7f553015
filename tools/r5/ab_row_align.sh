# same-lease A/B of the result tables' row padding on the headline sweep (config #2): SPX_OPT_ROW_ALIGN 128 (row stride 10112 B for 10 000
# nodes), 256 / 2048 (both 10240 B) and 4096 (12288 B), interleaved, 300 steps each
for rep in 1 2 3 4; do for v in 128 2048 256 4096; do python bench.py --steps 300 --warmup 30 --cpu-budget 0 --no-config5-leg --sweep-only --opt ROW_ALIGN=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('row_align=$v kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))"; done; done
