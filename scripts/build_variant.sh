#!/bin/bash
# A/B builds of the library (developer tool): scripts/build_variant.sh NAME [-DKNOB=1 ...]  ->  abtest/r06/lib_NAME.so
# (abtest/ is git-ignored but travels to the GPU box; time the builds against each other with scripts/ab_time.py, compare results with scripts/cmp_builds.py)
n=$1; shift
mkdir -p abtest/r06
python3 -m gomel_amd.build --force -o abtest/r06/lib_$n.so "$@" > /dev/null && echo "built $n" || echo "FAILED $n"
