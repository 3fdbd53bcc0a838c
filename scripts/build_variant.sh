#!/bin/bash
# A/B builds of the library (developer tool): scripts/build_variant.sh NAME [-DKNOB=1 ...]  ->  abtest/r05/lib_NAME.so
# (abtest/ is git-ignored but travels to the GPU box; time the builds against each other with scripts/ab_time.py, compare results with scripts/cmp_builds.py)
n=$1; shift
mkdir -p abtest/r05
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Xclang -target-feature -Xclang -load-store-opt "$@" \
    -o abtest/r05/lib_$n.so gomel_amd/csrc/gomel_hip.hip 2>&1 | grep -v "is not a recognized feature for this target\|^$" ; echo "built $n"
