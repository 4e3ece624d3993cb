#!/usr/bin/env bash
# register and scratch use of the kernels in a built object / library, read from the code objects' metadata
# (llvm-objdump --offloading extracts the gfx950 code objects of a host object or shared library).
# Usage: bash tools/kernel_regs.sh alphazero_amd/csrc/build/az_tree.o [name filter]
f=$(readlink -f $1); pat=${2:-.}
tmp=$(mktemp -d); cp $f $tmp/in.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o >/dev/null 2>&1)
for co in $tmp/in.o.*gfx950*; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co; done | python3 -c "
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'agpr_count' and 'agpr_count' in cur:   # first key of a kernel's map
        rows.append(cur); cur = {}
    cur[k] = v
rows.append(cur)
for r in rows:
    if 'name' not in r: continue
    try: name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    except Exception: name = r['name']
    if re.search(r'''$pat''', name):
        print('%-70s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s B spills %s' % (name[:70], r.get('vgpr_count'), r.get('agpr_count'), r.get('sgpr_count'), r.get('group_segment_fixed_size'), r.get('private_segment_fixed_size'), r.get('vgpr_spill_count')))
"
rm -rf $tmp
