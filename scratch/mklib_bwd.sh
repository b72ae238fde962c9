#!/bin/bash
# scratch/mklib_bwd.sh NAME [hipcc flags...]: scratch/libs/NAME.so = the product library with encoder_bwd_fused.hip recompiled
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scratch/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 "$@" -c tacorl_amd/csrc/encoder_bwd_fused.hip -o scratch/libs/$name.o 2>&1 | grep -v "hip-link\|loop not unrolled\|__global__\|\^\|warning generated" || true
objs=$(ls tacorl_amd/lib/obj/*.o | grep -v encoder_bwd_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/$name.so scratch/libs/$name.o $objs
rm scratch/libs/$name.o
echo scratch/libs/$name.so
