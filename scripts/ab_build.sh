#!/bin/bash
# A/B builds of one translation unit: scripts/ab_build.sh <unit> <tag> [extra hipcc flags...]
# -> symbolic_music_generation_amd/build/libmusicxl_<tag>.so (load it with MXL_LIB_PATH); the other objects come from the normal build.
set -e
cd "$(dirname "$0")/../symbolic_music_generation_amd"
unit=$1; tag=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form \
  -mllvm -amdgpu-use-amdgpu-trackers "$@" -c csrc/$unit.hip -o build/${unit}_$tag.o
objs=$(for s in csrc/*.hip; do f=$(basename $s .hip); [ $f = $unit ] || echo build/$f.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmusicxl_$tag.so $objs build/${unit}_$tag.o
echo build/libmusicxl_$tag.so
