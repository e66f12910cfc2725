#!/usr/bin/env python3
"""Per-shape table from an MI_PROF_TRACE file: groups launches by (kernel, bytes, flops)."""
import collections
import sys

rows = collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    k, b, f, ms = line.split()
    key = (k, float(b), float(f))
    rows[key][0] += 1
    rows[key][1] += float(ms)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = sys.argv[3] if len(sys.argv) > 3 else None
tot = sum(v[1] for v in rows.values()) / steps
print(f"total {tot:.2f} ms/step")
for (k, b, f), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if only and k != only:
        continue
    us = ms / n * 1e3
    print(f"{k:22s} n/step={n // steps:4d} {ms / steps:7.3f} ms/step  {us:8.1f} us  {b / 1e6:8.1f} MB {b / us / 1e3 if us else 0:7.0f} GB/s "
          f"{f / 1e9:8.2f} GF {f / us / 1e6 if us else 0:7.1f} TF/s")
