#!/bin/bash
# Throughput of the reference's SHIPPED shapes (resources/3DUnet_confocal_boundary/train_config.yml:94 patch [80,170,170];
# test_config.yml:37-40 patch [80,170,170] + halo [16,32,32] = 112x234x234) beside the nearest aligned shapes.
# bash tools/run_shipped_shapes.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
M="python tools/model_bench.py --name UNet3D --f-maps 32 --levels 4 --batch 1 --steps 10 --warmup 3"
for p in 80,170,170 80,168,168 80,176,176; do $M --patch $p >> $out/${tag}_shipped_shape_train.jsonl 2>> $out/err.log; done
for p in 112,234,234 112,232,232 112,240,240; do $M --patch $p --forward-only >> $out/${tag}_shipped_shape_predict.jsonl 2>> $out/err.log; done
