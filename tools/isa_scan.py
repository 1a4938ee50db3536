#!/usr/bin/env python3
"""What the compiler made of the kernels' memory operations (no GPU needed): every csrc/*.hip is compiled to gfx950 assembly and, per kernel, the
script reports the patterns that cost round 4 its easy milliseconds:
  lwl    a vector load followed by `s_waitcnt vmcnt(0)` before the next load: the lane has ONE item in flight (a load written behind a bounds check)
  sws    a store, a `vmcnt(0)`, another store: stores waiting for each other's acknowledgement (a load left pending over a loop that stores)
  flat   flat_load / flat_store inside a loop: an address whose address space the compiler lost (integer casts, a select between two pointers,
         pointers read out of a struct) -- counted in lgkmcnt as well and returning out of order, every later wait becomes a full one
  sload  scalar loads inside a loop (followed by an immediate wait when the kernel is short of SGPRs)
  spill  v_readlane / v_writelane (SGPR spills) inside loops, scratch accesses, VGPR / SGPR counts
    python tools/isa_scan.py [file.hip ...] [--all] [--pattern NAME]     (default: kernels with any finding; --pattern prints the L/W/S/B string of kernels matching NAME)"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

def assemble(src, out):
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-result', '-S', '--cuda-device-only', src, '-o', out],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    return out

def classify(line):
    if re.search(r'\b(global|flat|buffer)_load', line): return 'F' if 'flat_load' in line else 'L'
    if re.search(r'\b(global|flat|buffer)_store', line): return 'S'
    if re.search(r's_waitcnt.*vmcnt\(0\)', line): return 'W'
    if 's_barrier' in line: return 'B'
    if 's_swappc' in line: return 'C'
    return ''

def scan(path, want_all, pattern):
    text = open(path).read()
    meta = {}
    for b in re.findall(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', text, re.S):
        nm = re.search(r'\.name:\s+(\S+)', b).group(1)
        meta[nm] = tuple(int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1)) for k in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'private_segment_fixed_size'))
    parts = re.split(r'\n(_Z\w+):\s*;?[^\n]*\n', text)
    rows = []
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
        if name not in meta: continue
        depth, seq, flat_loop, sload_loop, spill_loop = 0, [], 0, 0, 0
        for l in body.split('\n'):
            m = re.search(r'Depth=(\d)', l)
            if l.startswith('.LBB') or l.startswith('; %bb'): depth = int(m.group(1)) if m else 0
            c = classify(l)
            if c: seq.append(c)
            if depth and re.search(r'\bflat_(load|store)', l): flat_loop += 1
            if depth and re.search(r'\bs_(buffer_)?load_dword', l): sload_loop += 1
            if depth and re.search(r'v_(read|write)lane_b32', l): spill_loop += 1
        seq = ''.join(seq)
        lwl = len(re.findall(r'[LF]W(?=[LF])', seq)); sws = len(re.findall(r'SW(?=S)', seq))
        short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').replace('void ', '', 1).split('(')[0]
        if pattern is not None:
            if pattern in short: print('%s\n    %s' % (short, seq))
            continue
        v, sg, vs, scr = meta[name]
        if want_all or lwl >= 2 or sws >= 2 or flat_loop or sload_loop or vs or scr:
            rows.append('%-12s %-64s lwl %3d  sws %3d  flat-in-loop %2d  sload-in-loop %2d  sgpr-spill-ops-in-loops %3d  vgpr %3d sgpr %3d vgpr-spills %d scratch %d'
                        % (os.path.basename(path)[:-2], short[:64], lwl, sws, flat_loop, sload_loop, spill_loop, v, sg, vs, scr))
    return rows

def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    want_all = '--all' in sys.argv
    pattern = sys.argv[sys.argv.index('--pattern') + 1] if '--pattern' in sys.argv else None
    if pattern in args: args.remove(pattern)
    srcs = [os.path.join(HERE, 'uq_amd', 'csrc', a) if not os.path.exists(a) else a for a in args] or sorted(glob.glob(os.path.join(HERE, 'uq_amd', 'csrc', '*.hip')))
    tmp = tempfile.mkdtemp(prefix='isa_scan_')
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(lambda s: assemble(s, os.path.join(tmp, os.path.basename(s)[:-4] + '.s')), srcs))
    for o in outs:
        for r in scan(o, want_all, pattern): print(r)

if __name__ == '__main__':
    main()
