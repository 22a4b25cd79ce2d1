#!/bin/bash
# usage: tools/dbg/spills.sh <file.hip> <kernel-substring>   -- resource usage + spill placement by barrier interval
cd /root/repo/2d-vq-ae-2_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -S --cuda-device-only ${EXTRA:-} -o /tmp/w43/k.s "$1" 2>/dev/null
python3 - "$2" <<'PY'
import sys
txt=open('/tmp/w43/k.s').read()
key=sys.argv[1]
import re
for m in re.finditer(r'^(_Z\w*'+re.escape(key)+r'\w*):', txt, re.M):
    start=m.start(); end=txt.index('.end_amdhsa_kernel', start)
    seg=txt[start:end]
    body=seg.split('\n')
    g=lambda k: re.search(k+r'\s+(\d+)', seg)
    print(m.group(1)[:70], 'lines', len(body), 'vgpr', g(r'\.amdhsa_next_free_vgpr').group(1), 'accum_offset', g(r'\.amdhsa_accum_offset').group(1), 'scratch', g(r'\.amdhsa_private_segment_fixed_size').group(1))
    bar=[i for i,l in enumerate(body) if 's_barrier' in l]
    edges=[0]+bar+[len(body)]
    for a,b in zip(edges[:-1],edges[1:]):
        cnt=lambda k: sum(1 for l in body[a:b] if k in l)
        if cnt('scratch_'): print(f'  lines {a:6d}-{b:6d}: mfma {cnt("v_mfma"):4d} gload {cnt("global_load"):3d} spill st {cnt("scratch_store"):4d} ld {cnt("scratch_load"):4d}')
PY
