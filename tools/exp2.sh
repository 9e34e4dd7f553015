J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,1),"Gevals/s kern_ms",round(r["kernel_ms"],3),"GB/s",round(r["achieved"]),"frac",round(r["frac"],3))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== default (round 0)"; python bench.py --cpu-budget 0 2>&1 | python -c "$J"
echo "== round 0.1"; python bench.py --cpu-budget 0 --round-frac 0.1 2>&1 | python -c "$J"
echo "== tlp only"; python bench.py --cpu-budget 0 --plugins tlp 2>&1 | python -c "$J"
echo "== fast1 default"; SPX_TLP_FAST1=1 python bench.py --cpu-budget 0 2>&1 | python -c "$J"
