#!/bin/bash
# Build libconvnet_hip.so (gfx950) in-tree; with "emul" also build the TEST-ONLY CPU emulator
# library used by the `-m "not gpu"` kernel-logic tests.
set -e
cd "$(dirname "$0")"
SRCS="runtime.hip igemm.hip wgrad.hip junction.hip stem.hip conv3x3.hip bn.hip pool.hip loss.hip optim.hip probe.hip comm.hip quant.hip qconv_i8.hip"
OUT=..
if [ "$1" != "emul-only" ]; then
  # CN_EXTRA_FLAGS / CN_LIB_NAME: A/B builds (e.g. CN_EXTRA_FLAGS=-DCN_NT_STORES CN_LIB_NAME=libconvnet_hip_nt.so)
  LIB=${CN_LIB_NAME:-libconvnet_hip.so}
  # one object per source, compiled in parallel (igemm.hip alone is most of the serial build), unchanged objects reused
  ODIR=${CN_OBJ_DIR:-/tmp/cn_hip_obj}${CN_EXTRA_FLAGS:+_ab}
  mkdir -p $ODIR
  pids=""
  for s in $SRCS; do
    o=$ODIR/${s%.hip}.o
    if [ ! -f $o ] || [ $s -nt $o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer $o)" ] || [ -n "$CN_EXTRA_FLAGS" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $CN_EXTRA_FLAGS -c $s -o $o &
      pids="$pids $!"
    fi
  done
  for p in $pids; do wait $p; done
  OBJS=""
  for s in $SRCS; do OBJS="$OBJS $ODIR/${s%.hip}.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS -ldl -o $OUT/$LIB
  echo "built $OUT/$LIB"
fi
if [ "$1" = "emul" ] || [ "$1" = "emul-only" ]; then
  CXX=/opt/rocm/lib/llvm/bin/clang++
  OBJS=""
  for s in $SRCS; do
    $CXX -x c++ -std=c++17 -O2 -fPIC -DCN_EMULATE -Wno-unused-result -Wno-shift-negative-value -c $s -o /tmp/cn_emul_${s%.hip}.o &
  done
  wait
  for s in $SRCS; do OBJS="$OBJS /tmp/cn_emul_${s%.hip}.o"; done
  $CXX -std=c++17 -O2 -fPIC -DCN_EMULATE -c cn_emul.cpp -o /tmp/cn_emul_rt.o
  $CXX -shared -o $OUT/libconvnet_emul.so $OBJS /tmp/cn_emul_rt.o
  echo "built $OUT/libconvnet_emul.so"
fi
