# -*- coding: utf-8 -*-
"""Print the shader-clock stamps a tile-kernel workgroup left (SSQ_TILE_TRACE=<file>, see
csrc/ssq_cwt_tiles.hip): per wavefront and step -- top of step, loads issued, arithmetic done,
turn taken, update done (cycles relative to the workgroup's first stamp)."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
t0 = int(h[16 * 16 * 8])
st = h[:16 * 16 * 8].reshape(16, 16, 8).astype(np.int64)
print("wave: loop end, kernel end (cycles after workgroup start)")
for w in range(16):
    a, b = int(h[16 * 16 * 8 + 1 + w]), int(h[16 * 16 * 8 + 20 + w])
    if a:
        print(w, a - t0, b - t0)
print("wave step: top, loads issued, math done, turn taken, update done | step time")
for w in range(16):
    prev = None
    for j in range(16):
        if st[w, j, 0]:
            r = [int(v) - t0 for v in st[w, j, :5]]
            print(w, j, r, '' if prev is None else r[0] - prev)
            prev = r[0]
