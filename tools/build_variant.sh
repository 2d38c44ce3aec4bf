#!/bin/bash
# Build an experiment variant of libsicn.so beside the product build:
#   tools/build_variant.sh NAME "-DFLAG=1 ..." [file.hip ...]
# -> gpurun_build/NAME/libsicn.so (git-ignored, travels to the GPU box; select it with SICN_LIB=$PWD/gpurun_build/NAME/libsicn.so).
# Only the listed sources are recompiled with the flags (default: all); the other objects are the product build's.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/simple_image_compression_network_amd/csrc
NAME=$1; FLAGS=$2; shift 2
OUT=$ROOT/gpurun_build/$NAME
mkdir -p "$OUT"
make -C "$CSRC" -j4 ARCH=gfx950 >/dev/null
if [ $# -eq 0 ]; then
  make -C "$CSRC" -j4 ARCH=gfx950 OBJDIR="$OUT" OUT="$OUT/libsicn.so" EXTRA="$FLAGS" >/dev/null
else
  rm -f "$OUT"/*.o
  # every object of the product build (the Makefile's SRCS, whatever they are this round: a fixed list once missed k_l0g / k_l7g and the
  # variant failed at dlopen with undefined symbols)
  for o in $(make -s -C "$CSRC" print-objs); do cp "$CSRC/$o" "$OUT"/; done
  for f in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $FLAGS -c "$CSRC/$f" -o "$OUT/${f%.hip}.o" &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsicn.so" "$OUT"/*.o
fi
echo "built gpurun_build/$NAME/libsicn.so ($FLAGS)"
