"""Phase trace of the pair kernel (FA_TRACE build): per-wave cycle stamps
[start, gathers done, compute done, barrier 1 passed, LDS adds done, barrier 2 passed, stores issued, stores retired]."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
T = (a - a[:, :1]).astype(np.float64)
ok = (T[:, 1] > 0) & (T[:, 7] < 1e7)
T = T[ok]
print("waves", len(T))
names = ["record + gathers", "compute", "barrier 1 wait", "LDS adds", "barrier 2 wait", "write-out issue", "store drain"]
for i, n in enumerate(names):
    d = T[:, i + 1] - T[:, i]
    print(f"{n:20s} mean {d.mean():8.0f}  p10 {np.percentile(d,10):8.0f}  p50 {np.percentile(d,50):8.0f}  p90 {np.percentile(d,90):8.0f}")
print("wave life mean", T[:, 7].mean())
