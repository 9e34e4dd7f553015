#!/usr/bin/env python3
"""Scratch copies of kernels_trimaran.hip for the headline A/B: is the table-mode sweep slower than round 1's
because of what round 2 added to it?  Suspects: the by-value DecideArgs kernel argument (+88 bytes of kernarg) and the D template
branch.  Writes tools/_var/kernels_trimaran_<name>.hip; build each with tools/variant.py build <name> kernels_trimaran.hip --from <file>.
  nodec   the table-mode instantiations take no DecideArgs (a separate kernel body is instantiated for them through a wrapper type)
"""
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
src = (ROOT / "scheduler-plugins_amd/csrc/kernels_trimaran.hip").read_text()
out = ROOT / "tools/_var"
out.mkdir(exist_ok=True)
# nodec: DecideArgs becomes an empty struct for the table-mode kernels — pass a zero-size tag type instead
s = src
s = s.replace("void k_tlp_fast2(TrimaranArgs a, int n_tiles, double c1, double c2, DecideArgs dec) {",
              "void k_tlp_fast2(TrimaranArgs a, int n_tiles, double c1, double c2, std::conditional_t<D, DecideArgs, DecideNone> dec_in) {\n"
              "  DecideArgs dec_store{};\n  if constexpr (D) dec_store = dec_in;\n  const DecideArgs& dec = dec_store;")
s = s.replace("struct DecideArgs {", "struct DecideNone {};\nstruct DecideArgs {", 1)
s = s.replace("n_tiles, c1, c2, DecideArgs{});", "n_tiles, c1, c2, DecideNone{});")
s = s.replace('#include <cstdlib>\n', '#include <cstdlib>\n#include <type_traits>\n', 1)
(out / "kernels_trimaran_nodec.hip").write_text(s)
print("wrote", out / "kernels_trimaran_nodec.hip")
