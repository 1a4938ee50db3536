#!/usr/bin/env python3
"""profiles/pack_stats_traffic.json from PMC passes (tools/pmc_traffic.sh): one entry per (workload, reads, kernel form), each with the HBM bytes per launch of the
step's pack kernel, the command it came from and the sha256 of the kernel's source -- bench.py quotes an entry only while uq_amd/csrc/pack.hip still hashes to it.
    python tools/update_traffic.py <workload: cfg2|cfg5-ntrick|cfg5-notricks> <reads> <length> <form: qname|plain> <gpurun_out/..._traffic.json> <kernel name> <command...>"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, reads, length, form, src, kernel = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
cmd = ' '.join(sys.argv[7:])
t = json.load(open(src))
k = t[next(iter(t))] if len(t) == 1 else t['pack_tile_kernel']
path = os.path.join(HERE, 'profiles', 'pack_stats_traffic.json')
doc = json.load(open(path)) if os.path.exists(path) else {}
if 'entries' not in doc: doc = {'entries': []}
sha = hashlib.sha256(open(os.path.join(HERE, 'uq_amd', 'csrc', 'pack.hip'), 'rb').read()).hexdigest()
doc['kernel_source'], doc['kernel_source_sha256'] = 'uq_amd/csrc/pack.hip', sha
doc['entries'] = [e for e in doc['entries'] if e.get('kernel_source_sha256') == sha and not (e['workload'] == workload and e['reads'] == reads and e['kernel_form'] == form)]
doc['entries'].append({'workload': workload, 'reads': reads, 'length': int(length) if length.isdigit() else length, 'kernel_form': form, 'kernel': kernel,
                       'hbm_bytes_per_launch': k['hbm_bytes_per_launch'], 'fetch_bytes_corrected_x2': k['fetch_bytes_corrected_x2'], 'write_bytes': k['write_bytes'],
                       'kernel_source_sha256': sha,
                       'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only): tools/pmc_traffic.sh ... ' + cmd + '; profiles/traffic.py; FETCH_SIZE x2 per MI355X_MICROARCH.md'})
json.dump(doc, open(path, 'w'), indent=1)
print('wrote', path, len(doc['entries']), 'entries')
