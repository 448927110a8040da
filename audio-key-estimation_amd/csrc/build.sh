#!/usr/bin/env bash
# Build libake_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libake_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
mkdir -p build
pids=()
for f in common.cpp cqt.hip pcnet.hip pipeline.hip; do
  if [ ! -f build/${f%.*}.o ] || [ "$f" -nt build/${f%.*}.o ] || [ pcnet_kernels.h -nt build/${f%.*}.o ] || [ common.h -nt build/${f%.*}.o ] || [ ../../include/ake_hip.h -nt build/${f%.*}.o ]; then
    ( $HIPCC $FLAGS -x hip -c "$f" -o build/${f%.*}.o ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/common.o build/cqt.o build/pcnet.o build/pipeline.o -o $OUT
echo "built $(realpath $OUT)"
