#!/bin/bash
# sequential commit loop of the Filter-less profile: library variants (tools/variant.py) alternated on one box
#   tools/r4/ab_commit.sh <variant> [<variant> ...]
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
mkdir -p gpurun_out/ab_commit
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v rep$rep"
    timeout 300 python tools/variant.py run $v tools/r4/time_commit_trimaran.py 10000 100000 0,1 0 3 2>&1 | grep -v "^memory" | tail -3
  done
done
for v in "$@"; do
  echo "== $v: ties / LVRB / 4k nodes"
  timeout 300 python tools/variant.py run $v tools/r4/time_commit_trimaran.py 10000 100000 0,1 1 2 2>&1 | grep "registers" | tail -2
  timeout 300 python tools/variant.py run $v tools/r4/time_commit_trimaran.py 10000 20000 0,1,2 0 2 2>&1 | grep "registers" | tail -2
  timeout 300 python tools/variant.py run $v tools/r4/time_commit_trimaran.py 4000 100000 0,1 0 2 2>&1 | grep "registers" | tail -2
done
