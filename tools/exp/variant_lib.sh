#!/bin/bash
# usage: tools/exp/variant_lib.sh NAME SOURCE.hip "-DFLAG=1 ..."   ->  svision_amd/NAME_libsvx.so
# One source recompiled with extra flags, linked with the other objects of the main build (svision_amd/csrc/build).
set -e
cd "$(dirname "$0")/../../svision_amd/csrc"
name=$1; src=$2; flags=$3
make -s
mkdir -p build_var
obj=build_var/${name}_$(basename ${src%.*}).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c -o $obj $src
others=$(ls build/*.o | grep -v "build/$(basename ${src%.*}).o")
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o ../${name}_libsvx.so $obj $others -lz -lpthread -ldl
echo ../${name}_libsvx.so
