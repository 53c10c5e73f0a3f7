#!/bin/bash
# Register / scratch usage of every kernel in one .hip file:  tools/kernel_resources.sh <file.hip> [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$(dirname "$f")" "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/ {name=$3} /^VGPRs:/ {v=$2} /^AGPRs:/ {a=$2} /ScratchSize/ {s=$3} /Occupancy/ {o=$4} /LDS Size/ {print name, "vgpr", v, "agpr", a, "scratch", s, "occ", o}' |
  while read n rest; do echo "$(echo $n | c++filt | sed 's/(pnp::ConvArgs)//; s/void //') $rest"; done
