# -*- coding: utf-8 -*-
"""Print the shader-clock stamps one workgroup of the tile kernel left for its third tile
(SSQ_TILE_TRACE=<file>, see csrc/ssq_cwt_tiles.hip): per step (in ticket order) the wavefront
that took it, start of the step, prefetch point, arithmetic done, ticket passed."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
S = 128
st = h[:16 * S * 4].reshape(16, S, 4)
t0 = int(st[st > 0].min())
ev = []
for w in range(16):
    for j in range(S):
        if st[w, j, 0]:
            ev.append((j, w, *[int(v - t0) if v else -1 for v in st[w, j]]))
ev.sort()
prev = None
print("step wave   start  prefetch  computed    passed   (compute, wait+update, gap to the previous pass)")
for j, w, a, b, c_, d in ev:
    print("%4d  w%2d %8d %8d %8d %8d   (%6d, %6d, %s)" % (j, w, a, b, c_, d, c_ - a, d - c_, '' if prev is None else d - prev))
    prev = d
