#!/usr/bin/env python3
"""Derived figures from a pmc_summary listing: lanes active, vector units busy, waiting, L2 hit rate,
HBM fetch.  usage: pmc_digest.py <summary.txt> [segments of the counted dispatches]"""
import re
import sys

vals, kernel = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"== (.*)", line)
    if m:
        if kernel is None or "k_paths" in m.group(1):
            kernel, vals = m.group(1), {}
        continue
    m = re.match(r"\s+(\S+)\s+(\d+)\s+dispatches (\d+)", line)
    if m and kernel and "k_paths" in kernel:
        vals[m.group(1)] = float(m.group(2)) / (int(m.group(3)) / 2 if m.group(1) == "GRBM_GUI_ACTIVE" and int(m.group(3)) == 4 else 1)
g = vals.get
print(kernel)
if g("SQ_ACTIVE_INST_VALU"):
    print(f"  lanes active        {g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.1%}   (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU)")
    print(f"  vector units busy   {g('SQ_ACTIVE_INST_VALU') / (g('SQ_WAVE_CYCLES') / 4):.1%}   (SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 4 waves per SIMD), both in 4-cycle units)")
if g("SQ_WAIT_ANY"):
    print(f"  waves waiting       {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.1%}   (SQ_WAIT_ANY / SQ_WAVE_CYCLES)")
if g("TCC_REQ_sum"):
    print(f"  L2 hit rate         {g('TCC_HIT_sum') / g('TCC_REQ_sum'):.1%}")
if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
    cyc = g("GRBM_GUI_ACTIVE") / 8  # summed over the 8 XCDs
    print(f"  TA busy / TD busy   {g('TA_TA_BUSY_sum') / 256 / cyc:.1%} / {g('TD_TD_BUSY_sum') / 256 / cyc:.1%} of the kernel's cycles (per CU)")
if len(sys.argv) > 2:
    seg = float(sys.argv[2])
    print(f"  per segment         {g('SQ_INSTS_VALU') / seg:.1f} vector + {g('SQ_INSTS_SALU') / seg:.1f} scalar + {g('SQ_INSTS_VMEM') / seg:.2f} vector-memory wave-instructions"
          + (f", HBM fetch {g('FETCH_SIZE') * 1024 / seg:.0f} B" if g("FETCH_SIZE") else ""))
