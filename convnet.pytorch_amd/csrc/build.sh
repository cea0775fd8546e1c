#!/bin/bash
# Build libconvnet_hip.so (gfx950) in-tree; with "emul" also build the TEST-ONLY CPU emulator
# library used by the `-m "not gpu"` kernel-logic tests.
set -e
cd "$(dirname "$0")"
SRCS="runtime.hip plan.hip igemm.hip dense.hip wgrad.hip junction.hip stem.hip conv3x3.hip bn.hip pool.hip resize.hip loss.hip optim.hip probe.hip comm.hip quant.hip qconv_i8.hip"
OUT=..
if [ "$1" != "emul-only" ]; then
  # CN_EXTRA_FLAGS / CN_LIB_NAME: A/B builds (e.g. CN_EXTRA_FLAGS=-DCN_NT_STORES CN_LIB_NAME=libconvnet_hip_nt.so)
  LIB=${CN_LIB_NAME:-libconvnet_hip.so}
  # one object per source, compiled in parallel (igemm.hip alone is most of the serial build).  The object cache is
  # keyed by THIS source directory and the extra flags (two checkouts or two A/B builds never share objects), and an
  # object is reused only when the content hash of its source + every header it can see (csrc/*.h and the public
  # include/convnet_hip.h) + the flags equals the hash stored beside it - never on mtimes.
  KEY=$(printf '%s|%s' "$(pwd -P)" "$CN_EXTRA_FLAGS" | sha1sum | cut -c1-12)
  ODIR=${CN_OBJ_DIR:-/tmp/cn_hip_obj}_$KEY
  mkdir -p $ODIR
  HDRS=$(cat *.h ../../include/convnet_hip.h | sha1sum | cut -c1-40)
  # content hash of EVERY source + header of the library, compiled into runtime.hip: cn_build_info() reports which tree
  # the binary was built from (bench.py prints it beside the hash of the tree it runs in; _lib.source_hash() is the same recipe)
  SRC_HASH=$(cat $SRCS *.h ../../include/convnet_hip.h | sha1sum | cut -c1-16)
  pids=""
  for s in $SRCS; do
    o=$ODIR/${s%.hip}.o
    XF=""; if [ $s = runtime.hip ]; then XF="-DCN_SRC_HASH=\"$SRC_HASH\""; fi
    h=$(printf '%s|%s|%s|%s' "$(sha1sum < $s)" "$HDRS" "$CN_EXTRA_FLAGS" "$XF" | sha1sum | cut -c1-40)
    if [ ! -f $o ] || [ "$(cat $o.sha 2>/dev/null)" != "$h" ]; then
      rm -f $o.sha
      ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $CN_EXTRA_FLAGS $XF -c $s -o $o \
        && echo $h > $o.sha ) &
      pids="$pids $!"
    fi
  done
  for p in $pids; do wait $p || { echo "build failed" >&2; exit 1; }; done
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
