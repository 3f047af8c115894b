#!/usr/bin/env python3
"""Union of the kernel intervals of a rocprofv3 kernel trace (csv) against the wall time they span: how busy the GPU was.
   python3 scripts/gpu_busy.py <kernel_trace.csv> [skip_first_seconds]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0 = iv[0][0] + int(skip * 1e9)
iv = [(a, b) for a, b in iv if a >= t0]
busy = 0; cur_a, cur_b = iv[0]
gaps = []
for a, b in iv[1:]:
    if a > cur_b:
        busy += cur_b - cur_a; gaps.append((a - cur_b, cur_b)); cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
wall = iv[-1][1] - iv[0][0]
print(f"kernels {len(iv)}, wall {wall / 1e9:.3f}s, some kernel running {busy / 1e9:.3f}s = {busy / wall:.3f}; idle gaps: {len(gaps)}, longest {max(g for g, _ in gaps) / 1e6:.2f} ms, gaps > 1 ms: {sum(1 for g, _ in gaps if g > 1e6)} summing {sum(g for g, _ in gaps if g > 1e6) / 1e9:.3f}s")
