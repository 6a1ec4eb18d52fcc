#!/bin/bash
# The MEASURE=1 library (ablation arms of the K loops, smk_tune "ablate") as a SECOND build beside the product one:
# build_variants/measure/siammask_amd/libsiammask_hip.so; select it with SMK_LIB=<path> (siammask_amd/_lib.py).  Sources are copied,
# the product objects are not touched.  Built here on the CPU box; travels with the snapshot (build_variants/ is git-ignored, not
# gpurun-ignored).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); D=$R/build_variants/measure
mkdir -p $D/siammask_amd/csrc
for f in $R/siammask_amd/csrc/*.hip $R/siammask_amd/csrc/*.inc $R/siammask_amd/csrc/*.h $R/siammask_amd/csrc/*.cpp $R/siammask_amd/csrc/Makefile; do
  cmp -s $f $D/siammask_amd/csrc/$(basename $f) || cp $f $D/siammask_amd/csrc/      # (unchanged files keep their time stamps: incremental rebuilds)
done
rm -rf $D/include; cp -r $R/include $D/include
make -C $D/siammask_amd/csrc MEASURE=1 -j8 > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
echo "$D/siammask_amd/libsiammask_hip.so"
