#!/bin/bash
# scratch/mklib2.sh FILE NAME [hipcc flags...]: scratch/libs/NAME.so = the product library with csrc/FILE.hip recompiled with
# extra flags (e.g. mlp_fused stamps -DMLP_STAMPS); the other objects are reused from tacorl_amd/lib/obj.
set -e
cd "$(dirname "$0")/.."
file=$1; name=$2; shift; shift
mkdir -p scratch/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 "$@" -c tacorl_amd/csrc/$file.hip -o scratch/libs/$name.o
objs=$(ls tacorl_amd/lib/obj/*.o | grep -v "/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/$name.so scratch/libs/$name.o $objs
rm scratch/libs/$name.o
echo scratch/libs/$name.so
