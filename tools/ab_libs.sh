#!/bin/bash
# Same-box A/B of COMPILE-TIME kernel experiments: builds libu3d_hip_<tag>.so next to the product library, each with one translation
# unit recompiled under extra -D flags; select one at run time with U3D_LIB_PATH=<path> (pytorch3dunet_amd/_native.py).
#   bash tools/ab_libs.sh u3d_conv.hip burst "-DU3D_EXP_BURST" rb9 "-DU3D_EXP_RB9" both "-DU3D_EXP_BURST -DU3D_EXP_RB9"
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/pytorch-3dunet_amd
unit=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=200000 -I $ROOT/include -I $PKG/csrc"
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
pids=()
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  (
    /opt/rocm/bin/hipcc $FLAGS $defs -c $PKG/csrc/$unit -o $PKG/build/$unit.$tag.o
    objs=""
    for o in $PKG/build/*.hip.o; do
      if [ "$(basename $o)" = "$unit.o" ]; then objs="$objs $PKG/build/$unit.$tag.o"; else objs="$objs $o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/pytorch3dunet_amd/lib/libu3d_hip_$tag.so $objs
    echo "built $PKG/pytorch3dunet_amd/lib/libu3d_hip_$tag.so ($defs)"
  ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
