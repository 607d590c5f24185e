# -*- coding: utf-8 -*-
"""Print the shader-clock stamps one workgroup of the tile kernel left for its third tile
(SSQ_TILE_TRACE=<file>, a -DSSQ_TILE_TRACE_BUILD library; see csrc/ssq_cwt_tiles.hip): per step (in
ticket order) the wavefront that took it and, relative to the step's start: samples there / gather
issued (4), taps of row 0 there (5), row 0 done (6), row 1 done (7), next step's loads issued (1),
arithmetic done (2), ticket passed (3)."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
S, K = 128, 8
st = h[:16 * S * K].reshape(16, S, K)
t0 = int(st[st > 0].min())
ev = []
for w in range(16):
    for j in range(S):
        if st[w, j, 0]:
            ev.append((j, w, [int(v - t0) if v else -1 for v in st[w, j]]))
ev.sort()
prev = None
print("step wave    start | row0: gather  taps   done | row1 done | loads  | computed |  passed | gap to previous pass")
for j, w, v in ev:
    a = v[0]
    rel = lambda k: (v[k] - a) if v[k] >= 0 else -1
    print("%4d  w%2d %8d | %6d %6d %6d | %6d | %6d | %6d | %6d | %s" % (
        j, w, a, rel(4), rel(5), rel(6), rel(7), rel(1), rel(2), rel(3), '' if prev is None else v[3] - prev))
    prev = v[3]
