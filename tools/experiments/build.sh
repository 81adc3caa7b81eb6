#!/bin/bash
# Builds the three retired experiment kernels (see gimmvfi_experiments.h) into tools/experiments/libgimmvfi_experiments.so.
# Not part of __graft_entry__.build(); nothing in the product or the tests loads this library.
set -e
cd "$(dirname "$0")"
ROOT=../..
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -I$ROOT/gimm-vfi_amd/csrc -I$ROOT/include -I."
for f in csrc/*.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o ${f%.hip}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgimmvfi_experiments.so csrc/*.o
echo built tools/experiments/libgimmvfi_experiments.so
