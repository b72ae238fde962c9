#!/bin/bash
# scratch/mklib.sh NAME [hipcc flags...]: build scratch/libs/NAME.so = the product library with encoder_fused.hip
# recompiled with extra flags (e.g. -DEF_STAMPS -DEF_VAR=3); the other objects are reused from tacorl_amd/lib/obj.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scratch/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-load-store-vectorizer=0 "$@" -c tacorl_amd/csrc/encoder_fused.hip -o scratch/libs/$name.o 2>&1 | grep -v "hip-link" || true
objs=$(ls tacorl_amd/lib/obj/*.o | grep -v encoder_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/$name.so scratch/libs/$name.o $objs
rm scratch/libs/$name.o
echo scratch/libs/$name.so
