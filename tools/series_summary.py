"""Summarise `membw series` JSON lines: per shape min / median / p90 time, TB/s at each, even/odd launch medians."""
import json, sys
import numpy as np
for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    s = json.loads(line)
    d = np.array(s["dur_us"])[len(s["dur_us"]) // 5:]        # settled part
    tb = s["bytes_per_launch"] / 1e6
    q = lambda v: tb / v
    print("%-52s med %6.1f us  TB/s min-time %.2f  med %.2f  p90 %.2f  | std %4.1f us  even/odd med %.1f / %.1f" % (
        s["shape"], np.median(d), q(d.min()), q(np.median(d)), q(np.percentile(d, 90)), d.std(),
        np.median(d[0::2]), np.median(d[1::2])))
