# -*- coding: utf-8 -*-
"""Print the shader-clock stamps one workgroup of the tile kernel left for its third tile
(SSQ_TILE_TRACE=<file>, see csrc/ssq_cwt_tiles.hip): per wavefront and pair of steps -- start of
the arithmetic, end of it, turn taken, update done; and the write-out of the previous tile."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64)
st = h[:16 * 16 * 8].reshape(16, 16, 8).astype(np.int64)
t0 = int(st[st > 0].min())
ev = []
for w in range(16):
    if st[w, 0, 5]:
        print("wave %d: write-out of the previous tile %d .. %d" % (w, st[w, 0, 5] - t0, st[w, 0, 6] - t0))
    for j in range(0, 16, 2):
        if st[w, j, 0]:
            ev.append((int(st[w, j, 3]) - t0, int(st[w, j, 4]) - t0, w, j // 2, int(st[w, j, 0]) - t0,
                       int(st[w, j + 1, 2]) - t0))
ev.sort()
prev = None
for tt, ud, w, p, top, md in ev:
    print("wave %d pair %d: arithmetic %6d..%6d  turn %6d  done %6d  (update %4d, waited %5d, chain gap %s)"
          % (w, p, top, md, tt, ud, ud - tt, tt - md, '' if prev is None else tt - prev))
    prev = ud
