# -*- coding: utf-8 -*-
"""Print the shader-clock stamps one workgroup of the tile kernel left for its third tile
(SSQ_TILE_TRACE=<file>, see csrc/ssq_cwt_tiles.hip).
Updater (wavefront 0), per step: start of the update, end of it, flag seen (loads of the step
4 ahead are issued next), loads issued. Producers, per step: start, prefetch point, arithmetic
done, flag set."""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
S = 128
st = h[:16 * S * 4].reshape(16, S, 4)
extra = h[16 * S * 4:]
t0 = int(st[st > 0].min())
u = st[0]
print("updater: step  update_start  update_end  flag_seen  loads_issued   (dt to next update_start)")
rows = [j for j in range(S) if u[j, 0] or u[j, 2]]
for i, j in enumerate(rows):
    nxt = u[rows[i + 1], 0] - u[j, 0] if i + 1 < len(rows) and u[rows[i + 1], 0] and u[j, 0] else 0
    print("  %3d  %8d  %8d  %8d  %8d   %6d   upd %5d  load %5d" % (
        j, *[int(v - t0) if v else -1 for v in u[j]], nxt, u[j, 1] - u[j, 0], u[j, 3] - u[j, 2]))
if extra[1]:
    print("write-out %d .. %d (%d)" % (extra[1] - t0, extra[2] - t0, extra[2] - extra[1]))
print("producers: wave step  start  prefetch  done  flagged  (len)")
ev = []
for w in range(1, 16):
    for j in range(S):
        if st[w, j, 0]:
            ev.append((int(st[w, j, 0] - t0), w, j, *[int(v - t0) for v in st[w, j, 1:]]))
ev.sort()
for e in ev:
    print("  w%2d s%3d  %8d %8d %8d %8d  (%d, wait %d)" % (e[1], e[2], e[0], e[3], e[4], e[5], e[5] - e[0], e[5] - e[4]))
