J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,1),"Gevals/s kern_ms",round(r["kernel_ms"],3),"GB/s",round(r["achieved"]),"frac",round(r["frac"],3))'
echo "== default (round 0)"; python bench.py --cpu-budget 0 2>&1 | python -c "$J"
echo "== round 0.1"; python bench.py --cpu-budget 0 --round-frac 0.1 2>&1 | python -c "$J"
echo "== alloc only"; python bench.py --cpu-budget 0 --plugins alloc 2>&1 | python -c "$J"
echo "== tlp only"; python bench.py --cpu-budget 0 --plugins tlp 2>&1 | python -c "$J"
echo "== align16 default"; SPX_ROW_ALIGN=16 python bench.py --cpu-budget 0 2>&1 | python -c "$J"
echo "== align16 alloc only"; SPX_ROW_ALIGN=16 python bench.py --cpu-budget 0 --plugins alloc 2>&1 | python -c "$J"
echo "== lvrb (exact)"; python bench.py --cpu-budget 0 --plugins lvrb 2>&1 | python -c "$J"
