# -*- coding: utf-8 -*-
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name:
mean counter value per dispatch. Usage: python tools/pmc_summary.py <dir> [...]"""
import csv, glob, os, sys, collections


def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get('Kernel_Name') or row.get('Kernel Name')
                    cn = row.get('Counter_Name') or row.get('Counter Name')
                    cv = float(row.get('Counter_Value') or row.get('Counter Value') or 0)
                    a = agg[name][cn]
                    a[0] += cv; a[1] += 1
    for name, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        print(name[:100])
        for cn, (tot, cnt) in sorted(cs.items()):
            print("    %-26s mean/dispatch = %16.1f   (n=%d)" % (cn, tot / cnt, cnt))


if __name__ == '__main__':
    main(sys.argv[1:])
