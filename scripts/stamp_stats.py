#!/usr/bin/env python3
"""Summarise EDIGPU_TILE_STAMPS output: per-workgroup phases of the tiled panel sweep (100 MHz ticks -> us)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.uint64)
d = d[d[:, 1] > 0]
t0 = d[:, 1].min()
s = (d[:, 1:] - t0).astype(float) / 100.0
print("workgroups", len(s), "kernel span us", s[:, 3].max())
print("load phase (start->loads issued) med %.2f" % np.median(s[:, 1] - s[:, 0]))
print("wait phase (issued->barrier passed) med %.2f p90 %.2f" % (np.median(s[:, 2] - s[:, 1]), np.percentile(s[:, 2] - s[:, 1], 90)))
print("compute phase (barrier->stores done) med %.2f p90 %.2f" % (np.median(s[:, 3] - s[:, 2]), np.percentile(s[:, 3] - s[:, 2], 90)))
print("total per WG med %.2f p90 %.2f" % (np.median(s[:, 3] - s[:, 0]), np.percentile(s[:, 3] - s[:, 0], 90)))
st = np.sort(s[:, 0])
print("start times: first 1024 by %.2f us; quartiles of all starts" % st[min(1023, len(st) - 1)], np.percentile(st, [25, 50, 75, 100]))
# concurrency profile
ev = np.concatenate([np.stack([s[:, 0], np.ones(len(s))], 1), np.stack([s[:, 3], -np.ones(len(s))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for q in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = int(q * len(ev))
    print("  t=%.1f us concurrent WGs %d" % (ev[i, 0], conc[i]))
