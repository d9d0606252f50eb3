#!/bin/bash
L=$PWD/pytorch-3dunet_amd/pytorch3dunet_amd/lib
for v in "" cB cA cS cE cP cPE cLoop cAll ""; do
  lib=""; [ -n "$v" ] && lib=$L/libu3d_hip_$v.so
  echo "== variant '$v'"; U3D_LIB_PATH=$lib timeout 300 python tools/b16_bench.py --levels 3 2>&1 | grep "fwd\|dgrad" | grep -v sum | sed 's/fp32 storage.*bf16 storage/b16/'
done
