#!/bin/bash
# A second libsiammask_hip.so that differs from the product library in ONE translation unit compiled with extra -D flags
# (A/B arms of compile-time choices): build_variants/<name>/libsiammask_hip.so; select it with SMK_LIB=<path> (siammask_amd/_lib.py).
#   usage: build_variant.sh <name> <source.hip> "<extra flags>"      e.g.  build_variant.sh resreg0 conv_seq.hip "-DSMK_C3C1_RESREG=0"
# The product objects must be up to date (make -C siammask_amd/csrc).  Built here on the CPU box; travels with the snapshot.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/siammask_amd/csrc; NAME=$1; SRC=$2; EXTRA=$3
D=$R/build_variants/$NAME; mkdir -p $D
make -C $C -j4 > /dev/null
OBJ=${SRC%.*}.o
XFLAG=""; [ "${SRC##*.}" = "cpp" ] && XFLAG="-x hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $EXTRA $XFLAG -c $C/$SRC -o $D/$OBJ
OTHERS=$(ls $C/*.o | grep -v "/$OBJ$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $D/$OBJ -o $D/libsiammask_hip.so
echo "$D/libsiammask_hip.so"
