#!/bin/bash
# Register / LDS / scratch use of every kernel of a built library (development aid): tools/dev_kernel_regs.sh [lib] [filter]
LIB=${1:-lrf_amd/liblrf_hip.so}
FILT=${2:-.}
T=$(mktemp -d)
cp "$LIB" $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for CO in $T/lib.so.*gfx950*; do
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | python3 -c "
import sys, re
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'agpr_count' and cur.get('name'): rows.append(cur); cur = {}
    cur[k] = v
rows.append(cur)
for r in rows:
    if 'name' in r and re.search(sys.argv[1], r['name']):
        print(f\"{r.get('name','?')[:60]:60s} vgpr {r.get('vgpr_count','?'):>4s} agpr {r.get('agpr_count','?'):>4s} sgpr {r.get('sgpr_count','?'):>4s} spill {r.get('vgpr_spill_count','?'):>4s} lds {r.get('group_segment_fixed_size','?'):>6s} scratch {r.get('private_segment_fixed_size','?'):>5s}\")
" "$FILT"
done
rm -rf $T
