#!/bin/bash
# scratch/mklib_file.sh NAME FILE.hip[@GITREV] [hipcc flags...]: scratch/libs/NAME.so = the product library with ONE source
# file replaced (by the working-tree file compiled with extra flags, or by its version at a git revision) - for same-box
# A/B of a kernel change: TACORL_HIP_LIB=scratch/libs/NAME.so python ...
set -e
cd "$(dirname "$0")/.."
name=$1; spec=$2; shift 2
file=${spec%@*}; rev=""; [[ "$spec" == *@* ]] && rev=${spec#*@}
base=$(basename $file .hip)
mkdir -p scratch/libs
src=tacorl_amd/csrc/$base.hip
if [ -n "$rev" ]; then git show $rev:tacorl_amd/csrc/$base.hip > tacorl_amd/csrc/_ab_$base.hip; src=tacorl_amd/csrc/_ab_$base.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude "$@" -c $src -o scratch/libs/$name.o 2>&1 | grep -v "hip-link" || true
[ -n "$rev" ] && rm -f tacorl_amd/csrc/_ab_$base.hip
objs=$(ls tacorl_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/$name.so scratch/libs/$name.o $objs
rm scratch/libs/$name.o
echo scratch/libs/$name.so
