#!/bin/bash
# VGPR / spill census of the instantiations listed in tools/probes/regs_probe.hip
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $EXTRA_FLAGS -S --cuda-device-only tools/probes/regs_probe.hip -o /tmp/regs.s 2>/tmp/regs.err || tail -20 /tmp/regs.err
python3 - <<'PY'
import re
txt=open('/tmp/regs.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    name=m.group(1); body=m.group(2)
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k, body)
    vals={k:(g(k).group(1) if g(k) else '?') for k in ('vgpr_count','vgpr_spill_count','sgpr_count','sgpr_spill_count','private_segment_fixed_size')}
    short=re.sub(r'_ZN7sy_conv\d+','',name)
    print('%-60s vgpr %s spill %s | sgpr %s spill %s | scratch %s'%(short[:60],vals['vgpr_count'],vals['vgpr_spill_count'],vals['sgpr_count'],vals['sgpr_spill_count'],vals['private_segment_fixed_size']))
PY
