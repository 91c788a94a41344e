#!/bin/bash
# A/B builds of the library with compile-time switches, side by side with the production build:
#   profiles/ab_builds.sh tag "-DALS_MINW_LE4=4 -DALS_KB_ONLY=4" [tag2 "flags2" ...]   ->  csrc/libals_hip_<tag>.so
# Select one at run time with ALS_HIP_LIB=<path>.  Only row_solve.hip is rebuilt with the switches
# (-DALS_KB_ONLY=<KB> instantiates one model width: seconds instead of minutes).
set -e
cd "$(dirname "$0")/../collaborative-filtering_amd/csrc"
make -j4 >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -Wno-unused-function $flags -c row_solve.hip -o row_solve_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libals_hip_$tag.so row_solve_$tag.o row_solve_f64.o gs_sweep.o graph_build.o features.o stats.o predict.o w_step.o spd_solve.o host_setup.o
  echo "built libals_hip_$tag.so ($flags)"
done
