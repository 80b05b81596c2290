#!/bin/bash
# library variant for an A/B on the GPU box: one source rebuilt with extra flags, linked with the
# other objects of the in-tree build.  Usage: tools/build_variant.sh <name> <source.hip> <flags...>
# -> tools/ab/lib_<name>.so   (TRACEVAE_HIP_LIB=tools/ab/lib_<name>.so python ...)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../tracevae_amd/csrc"
mkdir -p ../../tools/ab
extra=""
case $src in gemm_f32.hip) extra="-fno-slp-vectorize";; esac
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $extra "$@" -c $src -o ../../tools/ab/${name}.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/lib_${name}.so $objs ../../tools/ab/${name}.o
echo built tools/ab/lib_${name}.so
