#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S` listing, and where its scratch (register-spill)
accesses sit.  usage: isa_stats.py <listing.s> <substring of the kernel's mangled name> [--spills]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel"))
body = lines[start:end]
c = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m:
        c[m.group(1)] += 1
def total(prefix):
    return sum(v for k, v in c.items() if k.startswith(prefix))
print(lines[start].split(":")[0])
print(f"  static instructions: vector {total('v_')}, scalar {total('s_')}, LDS {total('ds_')}, global/flat {total('global_') + total('flat_')}, "
      f"scratch loads {total('scratch_load')}, scratch stores {total('scratch_store')}, v_readlane {c['v_readlane_b32']}, v_writelane {c['v_writelane_b32']}")
if "--spills" in sys.argv:
    lab = None
    for i, l in enumerate(body):
        if re.match(r"\.LBB\d+_\d+:", l):
            lab = l.split(":")[0]
        if "scratch_" in l:
            print(f"  line {i:6d} after {lab}: {l.strip()}")
