#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc --offload-device-only -S)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "residual"
for f in re.split(r'\n(?=_Z\w+:)', s):
    name = f.split(':', 1)[0]
    if pat not in name:
        continue
    lines = [l.strip() for l in f.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
    c = Counter(l.split()[0] for l in lines)
    f64 = sum(v for k, v in c.items() if '_f64' in k)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    info = {k: re.search(r'; ' + k + r': (\d+)', f) for k in ('NumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize')}
    print(name[:48], 'instr', sum(c.values()), 'valu', valu, 'f64', f64, {k: int(v.group(1)) for k, v in info.items() if v})
    if len(sys.argv) > 3:
        for k, v in c.most_common(int(sys.argv[3])):
            print('   ', k, v)
