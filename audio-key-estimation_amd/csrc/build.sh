#!/usr/bin/env bash
# Build libake_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libake_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
# AKE_DIAG=1: the diagnostic build (-DAKE_DIAG: the AKE_* environment switches of the kernel experiments exist, common.h) -> libake_hip_diag.so,
# loaded instead of the shipped library only when AKE_USE_DIAG_LIB=1 is set for the Python process (tools/, tests/tools/).
BUILD=build
if [ "${AKE_DIAG:-0}" = 1 ]; then FLAGS="$FLAGS -DAKE_DIAG=1"; OUT=../libake_hip_diag.so; BUILD=build_diag; fi
mkdir -p $BUILD
HEADERS="common.h cqt_stream.h pcnet_kernels.h pcnet_bwd_kernels.h pcnet_backward.h ../../include/ake_hip.h"
pids=()
for f in common.cpp cqt.hip pcnet.hip pipeline.hip optim.hip audio.hip loss.hip; do
  obj=$BUILD/${f%.*}.o
  stale=0
  [ -f "$obj" ] || stale=1
  for d in "$f" $HEADERS; do [ "$d" -nt "$obj" ] && stale=1; done
  if [ $stale = 1 ]; then
    extra=""
    # cqt.hip: the SLP vectoriser turns the FIR taps into v_pk_* math fed by ds_read2_b32 gathers (4-way LDS bank
    # conflicts, measured: half of all LDS cycles); plain ds_read_b128 + scalar FMAs are faster
    [ "$f" = cqt.hip ] && extra="-fno-slp-vectorize"
    ( $HIPCC $FLAGS $extra -x hip -c "$f" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $BUILD/common.o $BUILD/cqt.o $BUILD/pcnet.o $BUILD/pipeline.o $BUILD/optim.o $BUILD/audio.o $BUILD/loss.o -o $OUT
echo "built $(realpath $OUT)"
