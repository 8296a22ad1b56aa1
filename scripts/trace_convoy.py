"""Are the write-out phases of the workgroups of one XCD bunched in time?  (FA_TRACE build; slot 3 = XCC_ID<<32 | HW_ID)"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
xcc = (a[:, 3] >> 32) & 0xF
hw = a[:, 3] & 0xFFFFFFFF
print("xcc ids", np.unique(xcc), " cu/se bits sample", [hex(int(h)) for h in hw[:4]])
for x in np.unique(xcc)[:2]:
    b = a[xcc == x]
    t0 = b[:, 0].min()
    T = (b[:, [0, 1, 2, 4, 5, 6, 7]] - t0).astype(np.float64)      # start, gathered, computed, added, barrier, issued, retired
    span = T[:, 6].max()
    print(f"xcc {x}: waves {len(b)}, span {span:.0f} cycles, mean wave life {np.mean(T[:,6]-T[:,0]):.0f}")
    lo = span * 0.5
    edges = lo + np.arange(0, 120001, 1000)
    cent = 0.5 * (edges[1:] + edges[:-1])
    def occ(i, j):
        return np.array([np.count_nonzero((T[:, i] <= c) & (c < T[:, j])) for c in cent])
    for name, i, j in (("gather ", 0, 1), ("compute", 1, 2), ("ldsadd ", 2, 3), ("write  ", 4, 6)):
        o = occ(i, j)
        print(name, "mean %6.1f std %6.1f  min %4d max %4d  " % (o.mean(), o.std(), o.min(), o.max()), o[:40])
