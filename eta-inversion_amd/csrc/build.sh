#!/bin/bash
# Build libetainv_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box with gpurun).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../etainv/lib"
mkdir -p "$OUT" "$HERE/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
pids=()
for f in step_kernels igemm norm attention misc maps aux_nets f32path xsgemm ppgemm ppconv; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/kernels.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/etainv.h" -nt "$HERE/obj/$f.o" ]; then
    EXTRA=""
    # attention: keep MFMA results in VGPRs (the softmax consumes them on the VALU: no v_accvgpr moves) and drop the
    # NaN-canonicalising v_max the compiler inserts in front of every fmaxf on MFMA outputs
    if [ "$f" = "attention" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"; fi
    hipcc $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
hipcc $FLAGS -x hip -c "$HERE/engine.cpp" -o "$HERE/obj/engine.o" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libetainv_hip.so" "$HERE"/obj/{step_kernels,igemm,norm,attention,misc,maps,aux_nets,f32path,xsgemm,ppgemm,ppconv,engine}.o
echo "built $OUT/libetainv_hip.so"
# diagnostic variant (STAMPS=1): igemm with in-kernel s_memtime stamps -> lib/libetainv_hip_stamps.so (load it with ETAINV_LIB=<path>;
# tools/experiments/*: reads SHARES of a K step, never a run time)
if [ "${STAMPS:-0}" = "1" ]; then
  hipcc $FLAGS -DETAINV_IGEMM_STAMPS -c "$HERE/igemm.hip" -o "$HERE/obj/igemm_stamps.o"
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libetainv_hip_stamps.so" "$HERE"/obj/{step_kernels,igemm_stamps,norm,attention,misc,maps,aux_nets,f32path,xsgemm,ppgemm,ppconv,engine}.o
  echo "built $OUT/libetainv_hip_stamps.so"
fi
# A/B variant of one kernel file (same-box comparisons; boxes of the pool differ by several percent):
#   VARIANT=name VARIANT_FILE=igemm VARIANT_FLAGS="-DETAINV_RES_PREFETCH=0" bash build.sh   ->  lib/libetainv_hip_name.so  (load with ETAINV_LIB)
if [ -n "${VARIANT:-}" ]; then
  vf="${VARIANT_FILE:-igemm}"
  EXTRA=""
  if [ "$vf" = "attention" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"; fi
  hipcc $FLAGS $EXTRA ${VARIANT_FLAGS:-} -c "$HERE/$vf.hip" -o "$HERE/obj/${vf}_$VARIANT.o"
  objs=""
  for f in step_kernels igemm norm attention misc maps aux_nets f32path xsgemm ppgemm ppconv engine; do
    if [ "$f" = "$vf" ]; then objs="$objs $HERE/obj/${vf}_$VARIANT.o"; else objs="$objs $HERE/obj/$f.o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libetainv_hip_$VARIANT.so" $objs
  echo "built $OUT/libetainv_hip_$VARIANT.so"
fi
