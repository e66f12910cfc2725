#!/bin/bash
# Compiler resource report of every kernel in the library (no GPU needed): VGPRs, scratch bytes per lane, occupancy, LDS.
# Prints the kernels that use scratch memory (register spills or dynamically indexed local arrays - the latter is how im2col3x3
# sat at 1.95 TB/s until round 2) and, with -a, the whole table.   usage: tools/resource_report.sh [-a] [file.hip ...]
cd "$(dirname "$0")/../image_restoration_amd/csrc"
ALL=0; [ "$1" = "-a" ] && { ALL=1; shift; }
FILES=${@:-*.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //;s/ \[-Rpass.*//' | paste - - - - - \
    | awk -v f=$f -v all=$ALL '{ if (all || $0 !~ /ScratchSize \[bytes\/lane\]: 0\t/) print f": "$0 }'
done
