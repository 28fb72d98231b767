#!/usr/bin/env python3
"""Static attribution of a kernel's VALU instructions to source functions.
Input: assembly from `hipcc -S -gline-tables-only --cuda-device-only`; for one kernel symbol, every
instruction is credited to the innermost source line of its last .loc, lines are mapped to the
enclosing function of render_kernels.h / render_pool_kernel.h by a brace scan.
Usage: tools/isa_attrib.py vimg_g.s <kernel-substring>"""
import re, sys, collections, os
asm, want = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = {}
func_of = {}
def load_funcs(path):
    m = {}
    cur = None
    depth = 0
    sig = re.compile(r'^\s*(?:template\s*<[^>]*>\s*)?(?:VD|static|inline|__device__|__global__|[\w:<>\*&\s])*?\b(\w+)\s*\([^;]*$')
    lines = open(path).read().split('\n')
    name = None
    for i, l in enumerate(lines, 1):
        if depth == 1 or depth == 0:
            mm = re.match(r'^(?:VD|template|__global__|static|inline).*?\b(\w+)\s*\(', l)
            if mm and not l.strip().startswith('//'):
                name = mm.group(1)
        m[i] = name
        depth += l.count('{') - l.count('}')
    return m
lines = open(asm).read().split('\n')
in_k = False
cur = (0, 0)
cnt = collections.Counter()
cnt_line = collections.Counter()
for l in lines:
    mm = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if mm:
        files[int(mm.group(1))] = os.path.join(mm.group(2), mm.group(3)); continue
    if re.match(r'^_Z\w+:', l):
        in_k = want in l
        continue
    if not in_k: continue
    mm = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2))); continue
    if re.match(r'\s+(v_|ds_|global_|scratch_|buffer_)', l):
        w = 2 if re.match(r'\s+v_\w+_f64', l) else 1   # FP64 issues at half rate
        f = files.get(cur[0], '?')
        if f not in func_of and os.path.exists(f) and 'csrc' in f:
            func_of[f] = load_funcs(f)
        fn = func_of.get(f, {}).get(cur[1], os.path.basename(f))
        cnt[fn] += w
        cnt_line[(os.path.basename(f), cur[1])] += w
tot = sum(cnt.values())
print(f"total weighted VALU+mem instructions: {tot}")
for k, v in cnt.most_common(45): print(f"{v:7d} {100*v/tot:5.1f} %  {k}")
print("top lines:")
for k, v in cnt_line.most_common(25): print(f"{v:7d}  {k[0]}:{k[1]}")
