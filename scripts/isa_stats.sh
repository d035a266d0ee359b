#!/bin/bash
# usage: scripts/isa_stats.sh <kernel-name-substring>   (device ISA of the library, instruction mix of one kernel)
set -e
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/opt/rocm/include -S --cuda-device-only -o jx.s /root/repo/joxsz_amd/csrc/joxsz_hip.hip -Wno-unused-function 2>/dev/null
python3 - "$1" <<'PY'
import re, sys, collections
s=open('/tmp/isa/jx.s').read()
funcs = re.split(r'\n(?=_Z[^\n]*:\s*;\s*@)', s)
for f in funcs:
    name=f.split(':')[0]
    if sys.argv[1] in name:
        open('/tmp/isa/sel.s','w').write(f)
        print(name)
        for pat in ['NumVgprs','NumAgprs','ScratchSize','Occupancy']:
            m=re.search(r'; %s: (\d+)'%pat,f); print(' ',pat, m.group(1) if m else None)
        lines=[l.strip() for l in f.split('\n') if re.match(r'\s+(v_|s_|ds_|global_|buffer_|flat_)',l)]
        print('  total instr',len(lines))
        c=collections.Counter(l.split()[0] for l in lines)
        print('  '+'  '.join('%s:%d'%(k,v) for k,v in c.most_common(28)))
PY
