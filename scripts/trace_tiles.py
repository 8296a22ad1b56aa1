"""Phase trace of the tile kernel (FA_TRACE build): per tile, waves 0-3: cycle stamps
[start, staged, points done, pair lanes done (stores issued), stores retired], then pair lanes, elements, nodes."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
T = (a[:, :5] - a[:, :1]).astype(np.float64)
print("waves", len(T), " pair lanes / elements / nodes per tile: %.0f / %.0f / %.0f" % (a[:, 5].mean(), a[:, 6].mean(), a[:, 7].mean()))
for i, n in enumerate(["stage", "Gauss points", "pair lanes", "store drain"]):
    d = T[:, i + 1] - T[:, i]
    print(f"{n:14s} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p50 {np.percentile(d,50):9.0f}  p90 {np.percentile(d,90):9.0f}  cycles")
print("wave life mean %.0f cycles (100 MHz counter: x24 for shader cycles at 2.4 GHz)" % T[:, 4].mean())
