J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,1),"Gevals/s kern_ms",round(r["kernel_ms"],3),"GB/s",round(r["achieved"]),"frac",round(r["frac"],3))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== lvrb only"; python bench.py --cpu-budget 0 --plugins lvrb 2>&1 | python -c "$J"
echo "== lvrb only round .1"; python bench.py --cpu-budget 0 --plugins lvrb --round-frac 0.1 2>&1 | python -c "$J"
echo "== config2_lvrb"; python bench.py --cpu-budget 0 --workload config2_lvrb 2>&1 | python -c "$J"
echo "== config2"; python bench.py --cpu-budget 0 2>&1 | python -c "$J"
