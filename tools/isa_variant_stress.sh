#!/bin/bash
# GPU box: tools/repeat_stress.py once per ISA variant library (tools/isa_variant_build.sh), one JSON line each.
#   tools/isa_variant_stress.sh "<sizes>" <passes> variant [variant ...]
SIZES=$1; PASSES=$2; shift 2
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=dt4image_restoration_amd/csrc/_isa/libpnpadmm_$v.so
  [ "$v" = shipped ] && lib=dt4image_restoration_amd/csrc/libpnpadmm.so
  echo "== $v" 
  PNP_LIB_PATH=$PWD/$lib timeout -k 10 300 python3 tools/repeat_stress.py --passes "$PASSES" --sizes "$SIZES" --modes bf16 --skip-episodes || true
done
