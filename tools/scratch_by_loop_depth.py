"""Where a kernel's scratch (spill) accesses sit: per kernel of the gfx950 code object, the number of scratch_load / scratch_store
instructions by the loop depth the compiler's own block comments give them (no GPU needed).  Spills outside the innermost
loops -- set up once per kernel, reloaded once per strip -- cost nothing measurable; spills inside an item loop do.
    python tools/scratch_by_loop_depth.py [name filter] [unit]   (compiles one translation unit of the library to assembly; unit =
    tu_scan_sorted (default), tu_morph, tu_scan, tu_grad, blueice_hip: see blueice_amd/csrc/bi_common.h)"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
flt = sys.argv[1] if len(sys.argv) > 1 else ''
unit = sys.argv[2] if len(sys.argv) > 2 else 'tu_scan_sorted'
out = os.path.join(tempfile.gettempdir(), 'blueice_hip_gfx950.s')
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-ffp-contract=off',
                '-mllvm', '--amdgpu-mfma-vgpr-form', '-Wno-unused-function', '-o', out,
                os.path.join(root, 'blueice_amd', 'csrc', unit + '.hip')], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
FUNC = re.compile(r'^(_Z\w+):')
mangled = [FUNC.match(l).group(1) for l in lines if FUNC.match(l)]
names = subprocess.run(['c++filt'], input='\n'.join(mangled), capture_output=True, text=True).stdout.split('\n')
demangled = dict(zip(mangled, names))
cur, depth, i, stats = None, 0, 0, {}
while i < len(lines):
    l = lines[i]
    if FUNC.match(l):
        cur, depth = demangled[FUNC.match(l).group(1)], 0
        stats[cur] = {}
    elif cur and re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l):
        txt, j = l, i + 1
        while j < len(lines) and re.match(r'^\s+;', lines[j]):
            txt += lines[j]; j += 1
        m = re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)', txt) or re.search(r'in Loop: Header=\S+ Depth=(\d+)', txt)
        depth = int(m.group(1)) if m else 0
    elif cur and 's_endpgm' in l:
        cur = None
    elif cur and re.search(r'\bscratch_(load|store)', l):
        stats[cur][depth] = stats[cur].get(depth, 0) + 1
    i += 1
for name, by in stats.items():
    if by and flt in name:
        print('%-90s %s' % (name[:90], ', '.join('depth %d: %d' % (d, n) for d, n in sorted(by.items()))))
