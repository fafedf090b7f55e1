#!/bin/bash
# AddressSanitizer build of the HOST side of libcapdec (device code is compiled as usual and never runs here) and of
# tests/host_asan/validate.cpp, which drives the argument-validation paths of the C ABI with null, misaligned and
# inconsistent arguments.  Needs no GPU.  Usage: tools/host_asan.sh   (objects under image-captioning_amd/csrc/build/asan)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/image-captioning_amd/csrc
B=$S/build/asan
mkdir -p $B
HIPCC=${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}
pids=()
for f in gemm ops lstm engine train_ops decode_fused; do
  if [ ! -f $B/$f.o ] || [ $S/$f.hip -nt $B/$f.o ] || [ $S/common.h -nt $B/$f.o ] || [ $R/include/capdec.h -nt $B/$f.o ]; then
    $HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $S/$f.hip -o $B/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -o $B/libcapdec_asan.so $B/gemm.o $B/ops.o $B/lstm.o $B/engine.o $B/train_ops.o $B/decode_fused.o
$HIPCC -O1 -g -std=c++17 -x c++ -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer $R/tests/host_asan/validate.cpp -o $B/validate_asan -L$B -lcapdec_asan -Wl,-rpath,$B
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 $B/validate_asan
