#!/usr/bin/env python3
"""Instruction histogram of the hottest basic blocks of a gfx950 kernel (from hipcc -save-temps .s).
usage: isa_hist.py file.s <substring of mangled kernel name> [n_blocks]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
funcs = re.findall(r'^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M)
for name, body in funcs:
    if want not in name:
        continue
    blocks = re.split(r'^\.LBB\d+_\d+:.*$', body, flags=re.M)
    def instrs(b):
        return [l.split()[0] for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    sized = sorted(((len(instrs(b)), i) for i, b in enumerate(blocks)), reverse=True)
    print(f"== {name}: {sum(n for n, _ in sized)} instrs, {len(blocks)} blocks; largest: {sized[:6]}")
    for n, i in sized[:nb]:
        ops = collections.Counter(instrs(blocks[i]))
        print(f"-- block {i}: {n} instrs")
        for k, v in ops.most_common(60):
            print(f"{v:5d} {k}")
