#!/bin/bash
# Build a variant of the library next to the in-tree one: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2" -> build_variants/NAME.so
# (the objects of the unchanged sources are compiled once into build_variants/obj and reused)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include"
mkdir -p build_variants/obj
for f in p2w_feat p2w_feat_h1; do
  [ build_variants/obj/$f.o -nt pointstowood_amd/csrc/$f.hip ] || /opt/rocm/bin/hipcc $base -c pointstowood_amd/csrc/$f.hip -o build_variants/obj/$f.o &
done
/opt/rocm/bin/hipcc $base $flags -c pointstowood_amd/csrc/p2w_geom.hip -o build_variants/obj/geom_$name.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/$name.so build_variants/obj/geom_$name.o build_variants/obj/p2w_feat.o build_variants/obj/p2w_feat_h1.o
echo built build_variants/$name.so
