#!/bin/bash
# A/B of compile-time variants of ONE translation unit on the GPU box: builds tmlqcd_amd/lib/variants/libtmlqcd_hip_<tag>.so from the
# objects of the regular build with <file>.hip recompiled under the given flags; select it with TMLQCD_HIP_LIB=<path>.
#   tools/build_variant.sh <tag> <file-without-.hip> <flags...>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; unit=$2; shift 2
mkdir -p $R/tmlqcd_amd/lib/variants
cd $R/tmlqcd_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast "$@" -c $unit.hip -o ../lib/variants/${unit}_$tag.o
objs=$(ls ../lib/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libtmlqcd_hip_$tag.so $objs ../lib/variants/${unit}_$tag.o -L/opt/rocm/lib -lrccl -lrt -lpthread -Wl,-rpath,/opt/rocm/lib
rm -f ../lib/variants/${unit}_$tag.o
echo "built tmlqcd_amd/lib/variants/libtmlqcd_hip_$tag.so"
