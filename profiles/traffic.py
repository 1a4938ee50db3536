#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE
runs, each with --kernel-trace only), corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: counter unit = KiB, and FETCH_SIZE reports one half of the bytes of a wide
coalesced streaming read (16 B per lane) -> doubled.  WRITE_SIZE is exact for 16-B-per-lane stores.

    python profiles/traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [kernel regex] [out.json]
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        per_dispatch[r['Dispatch_Id']] += float(r['Counter_Value'])
        names[r['Dispatch_Id']] = r['Kernel_Name']
    for d, v in per_dispatch.items():
        m = re.search(r'(\w+_kernel)', names[d])
        agg[m.group(1) if m else names[d][:40]].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    f = per_kernel(sys.argv[1], 'FETCH_SIZE')
    w = per_kernel(sys.argv[2], 'WRITE_SIZE')
    pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {}
    for k in sorted(set(f) | set(w)):
        if pat and not pat.search(k): continue
        fetch_raw = f.get(k, 0.0) * 1024
        write = w.get(k, 0.0) * 1024
        out[k] = {'fetch_bytes_raw': int(fetch_raw), 'fetch_bytes_corrected_x2': int(2 * fetch_raw), 'write_bytes': int(write),
                  'hbm_bytes_per_launch': int(2 * fetch_raw + write)}
        print('%-28s fetch(raw) %8.3f GB  fetch(x2) %8.3f GB  write %8.3f GB  total %8.3f GB' %
              (k, fetch_raw / 1e9, 2 * fetch_raw / 1e9, write / 1e9, (2 * fetch_raw + write) / 1e9))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
