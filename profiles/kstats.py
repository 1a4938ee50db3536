#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (kernel, calls, average us, share)."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(\w+_kernel(<[^>]*>)?|__amd\w+)', r['Name'])
    print('%-44s calls %4s  avg %10.1f us  %6s%%' % (m.group(1) if m else r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
