#!/bin/bash
# Build a variant of the library next to the in-tree one: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2" -> build_variants/NAME.so
# (the objects of the unchanged sources are compiled once into build_variants/obj and reused)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; what=${3:-geom}     # third argument "feat": the flags go to the feature sources instead of p2w_geom.hip
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include"
mkdir -p build_variants/obj
if [ "$what" = feat ]; then
  for f in p2w_feat p2w_feat_h1; do /opt/rocm/bin/hipcc $base $flags -c pointstowood_amd/csrc/$f.hip -o build_variants/obj/${f}_$name.o & done
  /opt/rocm/bin/hipcc $base -c pointstowood_amd/csrc/p2w_geom.hip -o build_variants/obj/geom_$name.o
  wait
  objs="build_variants/obj/geom_$name.o build_variants/obj/p2w_feat_$name.o build_variants/obj/p2w_feat_h1_$name.o"
else
  for f in p2w_feat p2w_feat_h1; do
    [ build_variants/obj/$f.o -nt pointstowood_amd/csrc/$f.hip -a build_variants/obj/$f.o -nt pointstowood_amd/csrc/p2w_hgemm.h ] || /opt/rocm/bin/hipcc $base -c pointstowood_amd/csrc/$f.hip -o build_variants/obj/$f.o &
  done
  /opt/rocm/bin/hipcc $base $flags -c pointstowood_amd/csrc/p2w_geom.hip -o build_variants/obj/geom_$name.o
  wait
  objs="build_variants/obj/geom_$name.o build_variants/obj/p2w_feat.o build_variants/obj/p2w_feat_h1.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/$name.so $objs
echo built build_variants/$name.so
