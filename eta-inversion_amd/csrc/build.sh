#!/bin/bash
# Build libetainv_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box with gpurun).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../etainv/lib"
# EXPERIMENTS=1: a second library, libetainv_hip_experiments.so (objects in obj_exp/), that also carries the opt-in experiments which lost -- xsgemm.hip
# (ETAINV_XSGEMM=1) and pp_gemm_kernel of ppgemm.hip (ETAINV_PP=1).  Load it with ETAINV_LIB=<path>; their tests skip on the default library.
OBJ="$HERE/obj"; LIBNAME=libetainv_hip.so; SRCS="step_kernels igemm norm attention misc maps aux_nets f32path ppgemm ppconv"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
if [ "${EXPERIMENTS:-0}" = "1" ]; then OBJ="$HERE/obj_exp"; LIBNAME=libetainv_hip_experiments.so; SRCS="$SRCS xsgemm"; FLAGS="$FLAGS -DETAINV_EXPERIMENTS"; fi
mkdir -p "$OUT" "$OBJ"
pids=()
for f in $SRCS; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/common.h" -nt "$OBJ/$f.o" ] || [ "$HERE/kernels.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/etainv.h" -nt "$OBJ/$f.o" ]; then
    EXTRA=""
    # attention: keep MFMA results in VGPRs (the softmax consumes them on the VALU: no v_accvgpr moves) and drop the
    # NaN-canonicalising v_max the compiler inserts in front of every fmaxf on MFMA outputs
    if [ "$f" = "attention" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"; fi
    hipcc $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
hipcc $FLAGS -x hip -c "$HERE/engine.cpp" -o "$OBJ/engine.o" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
objs=""; for f in $SRCS engine; do objs="$objs $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIBNAME" $objs
echo "built $OUT/$LIBNAME"
# diagnostic variant (STAMPS=1): igemm with in-kernel s_memtime stamps -> lib/libetainv_hip_stamps.so (load it with ETAINV_LIB=<path>;
# tools/experiments/*: reads SHARES of a K step, never a run time)
if [ "${STAMPS:-0}" = "1" ]; then
  hipcc $FLAGS -DETAINV_IGEMM_STAMPS -c "$HERE/igemm.hip" -o "$OBJ/igemm_stamps.o"
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libetainv_hip_stamps.so" ${objs/$OBJ\/igemm.o/$OBJ\/igemm_stamps.o}
  echo "built $OUT/libetainv_hip_stamps.so"
fi
# A/B variant of one kernel file (same-box comparisons; boxes of the pool differ by several percent):
#   VARIANT=name VARIANT_FILE=igemm VARIANT_FLAGS="-DETAINV_RES_PREFETCH=0" bash build.sh   ->  lib/libetainv_hip_name.so  (load with ETAINV_LIB)
if [ -n "${VARIANT:-}" ]; then
  vf="${VARIANT_FILE:-igemm}"
  EXTRA=""
  if [ "$vf" = "attention" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"; fi
  hipcc $FLAGS $EXTRA ${VARIANT_FLAGS:-} -c "$HERE/$vf.hip" -o "$OBJ/${vf}_$VARIANT.o"
  objs=""
  for f in $SRCS engine; do
    if [ "$f" = "$vf" ]; then objs="$objs $OBJ/${vf}_$VARIANT.o"; else objs="$objs $OBJ/$f.o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libetainv_hip_$VARIANT.so" $objs
  echo "built $OUT/libetainv_hip_$VARIANT.so"
fi
