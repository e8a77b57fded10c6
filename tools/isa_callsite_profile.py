#!/usr/bin/env python3
"""Diagnostic: static instruction count of pdb_step_kernel per call site in stepBody (which phase of the tick owns how much of the
instruction stream).  Builds the device code with -g, disassembles the kernel, resolves every instruction's inline stack with
llvm-symbolizer and buckets it by the line of stepBody it was inlined at.  The kernel is almost entirely straight-line code
(unrolled loops, lane-predicated branches that every wave walks), so static counts are close to what a wave issues; exceptions:
branches on the car model (suspension types, heave springs), the cold teleport block, the optional CarState output.
With `scratch` as second argument: where the kernel's scratch (spill) instructions sit instead -- by stepBody line and innermost
function -- so that a spill on the path every car takes cannot hide behind the cold blocks' ones.
With `lds`: the kernel's LDS instructions (ds_*) by call site instead, with the widths used.
usage: python3 tools/isa_callsite_profile.py [kernel-name] [scratch|scratchlist|lds]   (build container; no GPU needed)"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
kern = sys.argv[1] if len(sys.argv) > 1 else 'pdb_step_kernel'
only_scratch = len(sys.argv) > 2 and sys.argv[2] in ('scratch', 'scratchlist')
list_scratch = len(sys.argv) > 2 and sys.argv[2] == 'scratchlist'
only_lds = len(sys.argv) > 2 and sys.argv[2] == 'lds'   # every scratch instruction with its address and inline stack
tmp = os.environ.get('PDB_ISA_TMP') or tempfile.mkdtemp(prefix='pdb_isa_')   # PDB_ISA_TMP: keep / reuse the -g object between runs
csrc = os.path.join(ROOT, 'projectd-core_amd', 'csrc')
obj, co = os.path.join(tmp, 'k.o'), os.path.join(tmp, 'k.co')
if not os.path.exists(obj):
  subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-O3', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-g', '-DPDB_FAST_BUILD', '-mllvm', '-disable-machine-licm', '-fno-slp-vectorize',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(csrc, 'host'), '-I' + os.path.join(csrc, 'device'),
                       '--cuda-device-only', '-c', os.path.join(csrc, 'device', 'batch.hip'), '-o', obj])
subprocess.check_call([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + obj, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
dis = subprocess.run([LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', co], capture_output=True, text=True).stdout
ins, inside = [], False
for l in dis.split('\n'):
    m = re.match(r'^[0-9a-f]+ <(\w+)>:', l)
    if m:
        inside = m.group(1) == kern
        continue
    if inside:
        m = re.match(r'^\s+([a-z_0-9]+)\s.*// ([0-9A-F]+):', l)
        if m and (not only_scratch or m.group(1).startswith('scratch_')) and (not only_lds or m.group(1).startswith('ds_')):
            ins.append((int(m.group(2), 16), m.group(1) + (' ' + l.split('//')[0].split(None, 1)[1].strip() if only_scratch else '')))
sym = subprocess.run([LLVM + '/llvm-symbolizer', '--obj=' + co, '--inlines', '--output-style=JSON'], input='\n'.join('0x%x' % a for a, _ in ins),
                     capture_output=True, text=True).stdout
recs = [json.loads(l) for l in sym.strip().split('\n')]
src = open(os.path.join(csrc, 'device', 'step_kernel.hip.inc')).read().split('\n')
short = lambda fn: (re.search(r'(?:k\d+::)?(\w+)(?:<[^>]*>)?\(', fn) or re.search(r'(\w+)', fn)).group(1)
tot, valu, callee = collections.Counter(), collections.Counter(), {}
if list_scratch:
    for (a, op), r in zip(ins, recs):
        print('%6x %-22s %s' % (a, op, ' <- '.join('%s:%d' % (short(f['FunctionName']), f['Line']) for f in r['Symbol'])))
for (a, op), r in zip(ins, recs):
    fr = r['Symbol']
    idx = next((i for i, f in enumerate(fr) if 'stepBody' in f['FunctionName']), None)
    key = fr[idx]['Line'] if idx is not None else 0
    tot[key] += 1
    if op.startswith('v_'):
        valu[key] += 1
    if idx:
        callee.setdefault(key, collections.Counter())[short(fr[idx - 1]['FunctionName']) + (':%d' % fr[0]['Line'] if only_scratch else '')] += 1
print('%s: %d instructions' % (kern, len(ins)))
if only_lds:
    print('  by opcode: ' + ', '.join('%s %d' % kv for kv in collections.Counter(op for _, op in ins).most_common(16)))
for k, c in tot.most_common(40):
    cs = ', '.join('%s %d' % kv for kv in callee.get(k, collections.Counter()).most_common(3))
    print('%6d (valu %5d)  stepBody line %4d  %-70s %s' % (c, valu[k], k, src[k - 1].strip()[:70] if k else '(outside stepBody)', cs))
