#!/bin/bash
# Register / spill / LDS metadata of every kernel in a code object of libdpmmhip.so (or of one .hip file compiled on the spot).
#   bash scripts/kernel_regs.sh [file.hip] [name filter]
set -e
cd "$(dirname "$0")/../dpmmsubclusters.jl_amd/csrc"
F=${1:-niw_sweep.hip}; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-pass-failed -Wno-unused-value $EXTRA --cuda-device-only -S $F -o $T/k.s 2>/dev/null
python3 - "$T/k.s" "$PAT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, lds, name, priv, sg, sgs, vg, vgs = m.groups()
    if pat.search(name):
        print(f"{name}: vgpr {vg} agpr {ag} sgpr {sg} | spills: sgpr {sgs} vgpr {vgs} | scratch {priv} B | static LDS {lds} B")
PY
cp $T/k.s /tmp/isa/last_kernel_regs.s
rm -rf $T
