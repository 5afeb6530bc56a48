"""Dev: per-basic-block instruction census of one kernel in a hipcc -S listing.
usage: python scripts/isa_loops.py build/isa/knn.s <kernel-name-substring> <opcode that marks the blocks of interest> [min count]"""
import re, sys
path, kern, mark = sys.argv[1:4]
minc = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(kern) + r'\w*:', l)][0]
end = start
while 's_endpgm' not in lines[end]: end += 1
bb, cur = [], None
for i in range(start, end + 1):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), i, {}]; bb.append(cur)
    elif cur and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        op = l.split()[0]
        cur[2][op] = cur[2].get(op, 0) + 1
for name, i, ops in bb:
    n = ops.get(mark, 0)
    if n >= minc:
        tot = sum(ops.values())
        mf = sum(v for k, v in ops.items() if 'mfma' in k)
        valu = sum(v for k, v in ops.items() if k.startswith('v_') and 'mfma' not in k)
        salu = sum(v for k, v in ops.items() if k.startswith('s_'))
        vm = sum(v for k, v in ops.items() if k.startswith('buffer_') or k.startswith('global_'))
        ds = sum(v for k, v in ops.items() if k.startswith('ds_'))
        print(f"{name} line {i}: mfma {mf} valu {valu} ({valu/max(mf,1):.2f}/mfma) salu {salu} vmem {vm} lds {ds} total {tot}")
        print("   ", sorted(ops.items(), key=lambda x: -x[1])[:30])
