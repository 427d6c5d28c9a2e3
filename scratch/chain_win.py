import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tools import gabgen
cb = gabgen.chain(5, 10000, 0)
order = np.argsort(-cb.hdr["n"], kind="stable")[:300]
rows = []
for c in order:
    o = int(cb.call_off[c]); n = int(cb.hdr["n"][c]); x = cb.x[o:o+n].astype(np.int64)
    st = np.searchsorted(x, x - int(cb.hdr["max_dist_x"][c]), side="left")
    win = np.arange(n) - st
    win = np.minimum(win, 5000)
    rows.append((n, win.mean(), (win > 256).mean(), (win > 512).mean(), win.max(), (x[-1]-x[0])))
rows = np.array(rows)
print("top-300 calls by n: mean window", rows[:,1].mean(), "frac>256", rows[:,2].mean(), "frac>512", rows[:,3].mean())
idx = np.argsort(-rows[:,0]*rows[:,1])[:10]
for i in idx: print("n=%d meanwin=%.0f f256=%.2f f512=%.2f maxwin=%d span=%d" % tuple(rows[i]))
