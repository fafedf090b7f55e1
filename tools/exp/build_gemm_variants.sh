#!/bin/bash
# Experimental builds of the tuning library that differ only in compile-time constants of gemm.hip:
#   tools/exp/build_gemm_variants.sh name1 "-DX=1 -DY=2" name2 "-DX=3" ...
# -> image-captioning_amd/capdec/libcapdec_exp_<name>.so (linked with the tuning build's other objects).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/image-captioning_amd/csrc
OUT=$CSRC/build/exp
mkdir -p $OUT
pids=()
names=()
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  names+=($name)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DCAPDEC_TUNING_HOOKS $defs \
        -c $CSRC/gemm.hip -o $OUT/gemm_$name.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for name in "${names[@]}"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/image-captioning_amd/capdec/libcapdec_exp_$name.so \
        $OUT/gemm_$name.o $CSRC/build/tune/ops.o $CSRC/build/tune/lstm.o $CSRC/build/tune/engine.o $CSRC/build/tune/train_ops.o
  echo built libcapdec_exp_$name.so
done
