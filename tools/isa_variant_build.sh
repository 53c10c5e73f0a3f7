#!/bin/bash
# Build libpnpadmm variants that differ from each other in hand-edited gfx950 instructions only (tools/isa_patch.py):
#   tools/isa_variant_build.sh <file.hip without extension> "<extra -D flags>" variant [variant ...]
# -> dt4image_restoration_amd/csrc/_isa/libpnpadmm_<variant>.so   (PNP_LIB_PATH selects one; diagnostic builds, never shipped)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/dt4image_restoration_amd/csrc
L=/opt/rocm/lib/llvm/bin
STEM=$1; DEFS=$2; shift 2
OUT=$CS/_isa; TMP=${ISA_TMP:-/tmp/isa_build}; mkdir -p "$OUT" "$TMP"   # assembly (60 MB per variant) stays out of the tree
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $DEFS"
make -C "$CS" -s all
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only "$CS/$STEM.hip" -o "$TMP/$STEM.dev.s" 2>/dev/null
for v in "$@"; do
  python3 "$ROOT/tools/isa_patch.py" "$v" "$TMP/$STEM.dev.s" "$TMP/$STEM.$v.s"
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$TMP/$STEM.$v.s" -o "$TMP/$STEM.$v.dev.o"
  $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$TMP/$STEM.$v.out" "$TMP/$STEM.$v.dev.o"
  $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input="$TMP/$STEM.$v.out" -output="$TMP/$STEM.$v.hipfb"
  /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -c "$CS/$STEM.hip" -Xclang -fcuda-include-gpubinary -Xclang "$TMP/$STEM.$v.hipfb" -o "$TMP/$STEM.$v.o"
  OBJS=""
  for o in conv_kernels conv_bf16_kernels winograd_kernels winograd4_kernels fft_kernels pnp_capi; do
    if [ "$o" = "$STEM" ]; then OBJS="$OBJS $TMP/$STEM.$v.o"; else OBJS="$OBJS $CS/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libpnpadmm_$v.so" $OBJS
  rm -f "$TMP/$STEM.$v.dev.o" "$TMP/$STEM.$v.out" "$TMP/$STEM.$v.hipfb" "$TMP/$STEM.$v.o"
  echo "built $OUT/libpnpadmm_$v.so"
done
