#!/bin/bash
# Register / LDS / scratch / occupancy figures of every kernel in the library, from hipcc's own
# -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).  Output: profiles/<round>_kernel_resources.md
R=${1:-r02}
cd "$(dirname "$0")/../wavelets.jl_amd/csrc"
OUT=../../profiles/${R}_kernel_resources.md
{
echo "| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block (static) | waves/SIMD | VGPR spill | SGPR spill |"
echo "|---|---|---|---|---|---|---|---|---|"
for f in wl_fwd2d wl_fwd3d wl_inv3d wl_level3 wl_pair2d wl_fwd2d64 wl_pair2d64 wl_fwd2d_long wl_inv2d_long wl_tile wl_tail wl_gtile wl_anyaxis wl_fwd wl_inv wl_wpt wl_lift wl_lift_tile wl_axis wl_vlong wl_ext wl_generic; do
  extra=""; { [ $f = wl_vlong ] || [ $f = wl_inv2d_long ] || [ $f = wl_fwd3d ] || [ $f = wl_inv3d ]; } && extra="-fno-slp-vectorize"          # (as in the Makefile)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -DWL_BUILDING_LIB $extra \
      -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 | python3 -c "
import re, subprocess, sys
rows = {}; cur = None
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = t.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.rsplit(':', 1); rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    r = rows[n]
    depth = 0; cut = len(d)                      # strip the trailing parameter list only (names may contain '(anonymous namespace)')
    for i in range(len(d) - 1, -1, -1):
        if d[i] == ')': depth += 1
        elif d[i] == '(':
            depth -= 1
            if depth == 0: cut = i; break
    d = (d[:cut] if d.endswith(')') else d).replace('void ', '').replace('wl::', '').replace('(anonymous namespace)::', '')
    print('| \`%s\` | %s | %s | %s | %s | %s | %s | %s | %s |' % (d, r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'),
          r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]'), r.get('VGPRs Spill'), r.get('SGPRs Spill')))
"
done
} > $OUT
wc -l $OUT
