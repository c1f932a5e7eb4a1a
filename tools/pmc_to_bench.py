#!/usr/bin/env python3
"""Makes bench_pmc.json, the counter digests bench.py puts into its roofline object, from the
listings tools/pmc_summary.py writes (one per workload, from tools/pmc_quick.sh).

    pmc_to_bench.py c3=profiles/r3_pmc_c3_spp64.txt:861323066 c4=profiles/r3_pmc_c4.txt:294235320 ...

name=listing:segments -- segments traced by the dispatches of the path kernel that ONE counter pass
saw (every pass runs the same two frames).  HBM bytes per segment = (2 x FETCH_SIZE + WRITE_SIZE) x
1024 / segments: both counters are in KiB, and on gfx950 FETCH_SIZE tallies a 128-byte request as 64."""
import json
import re
import sys
from pathlib import Path

out = {}
for arg in sys.argv[1:]:
    name, rest = arg.split("=", 1)
    path, seg = rest.rsplit(":", 1)
    seg = float(seg)
    vals, kernel = {}, None
    for line in open(path):
        m = re.match(r"== (.*)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+(\S+)\s+(\d+)\s+dispatches (\d+)", line)
        if m and "k_paths" in cur:
            kernel = cur
            vals[m.group(1)] = float(m.group(2))
    if kernel is None:
        raise SystemExit(f"{path}: no path kernel in the listing")
    g = vals.get
    entry = {"kernel": kernel.replace("rt::", "").split("<")[0], "source": path, "segments_counted": int(seg)}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        entry["hbm_bytes_per_segment"] = round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / seg, 3)
        entry["fetch_kib"], entry["write_kib"] = int(g("FETCH_SIZE")), int(g("WRITE_SIZE"))
    if g("SQ_ACTIVE_INST_VALU"):
        entry["lanes_active"] = round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 4)
        entry["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") / (g("SQ_WAVE_CYCLES") / 4), 4)
        entry["valu_per_64_segments"] = round(64 * g("SQ_INSTS_VALU") / seg, 1)
        entry["salu_per_64_segments"] = round(64 * g("SQ_INSTS_SALU") / seg, 1)
    out[name] = entry
Path(__file__).resolve().parents[1].joinpath("bench_pmc.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
