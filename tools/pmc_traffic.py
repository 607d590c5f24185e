# -*- coding: utf-8 -*-
"""HBM traffic per transform from rocprofv3 PMC passes (tools/pmc_collect.sh).

    python tools/pmc_traffic.py <pmc_dir> <transforms_in_the_run> [out.json]

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch. On gfx950 FETCH_SIZE reports half of
the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section) -- checked here on
the reassignment kernel, whose read volume is known (Wx 8 B + bin map 2 B per point):
the ratio printed below must be ~0.5 -- so reads are doubled; WRITE_SIZE matches the
known write volume of the same kernel (Tx, 8 B per point) and is used as is.
"""
import csv, glob, json, os, sys, collections


def load(d, counter):
    tot = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if (row.get('Counter_Name') or row.get('Counter Name')) == counter:
                    name = row.get('Kernel_Name') or row.get('Kernel Name')
                    tot[name] += float(row.get('Counter_Value') or row.get('Counter Value'))
    return tot


def main(d, n_transforms, out=None):
    n_transforms = int(n_transforms)
    fetch, write = load(d, 'FETCH_SIZE'), load(d, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
        rd = 2 * fetch.get(k, 0) * 1024 / n_transforms
        wr = write.get(k, 0) * 1024 / n_transforms
        rows.append((k, rd, wr))
    total = sum(r[1] + r[2] for r in rows)
    for k, rd, wr in rows[:14]:
        print("%-90s read %9.1f MB  write %9.1f MB" % (k[:90], rd / 1e6, wr / 1e6))
    acc = [r for r in rows if 'accumulate' in r[0]]
    if acc:
        known_rd, known_wr = 300 * 160000 * 10, 300 * 160000 * 8
        print("calibration on %s: FETCH_SIZE/known = %.3f, WRITE_SIZE/known = %.3f"
              % (acc[0][0][:40], acc[0][1] / 2 / known_rd, acc[0][2] / known_wr))
    print("total per transform: %.1f MB" % (total / 1e6))
    if out:
        # the device code measured: the library's own stamp (ssq_build_sha: the last commit that touched csrc/ or
        # include/), which bench.py compares with the library it runs (`roofline.traffic_stale`)
        import ctypes
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            lib = ctypes.CDLL(os.environ.get('SSQ_HIP_LIB') or os.path.join(root, 'ssqueezepy_amd', 'libssq_hip.so'))
            lib.ssq_build_sha.restype = ctypes.c_char_p
            sha = lib.ssq_build_sha().decode()
        except Exception as e:
            sha = 'unknown (%r)' % (e,)
        json.dump({"git_sha": sha, "bytes_per_transform": total,
                   "per_kernel": {k[:120]: {"read": rd, "write": wr} for k, rd, wr in rows[:20]},
                   "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                             "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, summed over "
                             "all kernels of the run, divided by transforms"},
                  open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:4])
