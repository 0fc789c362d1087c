#!/bin/bash
# Report every gfx950 kernel that spills registers or uses scratch (none should).
cd "$(dirname "$0")/.." || exit 1
for f in gaviko_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
   | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' \
   | awk '/Function Name:|^ *Name:/{name=$NF} /ScratchSize/{if ($NF+0>0) print "SCRATCH", $NF, name} /VGPRs Spill/{if ($NF+0>0) print "VGPR-SPILL", $NF, name} /SGPRs Spill/{if ($NF+0>0) print "SGPR-SPILL", $NF, name}'
done
echo "spill check done"
