#!/bin/bash
# A/B variants of one kernel file: tools/build_variants.sh <file.hip> name1 "flags1" name2 "flags2" ...  ->  variants/libmcx_<name>.so
# (the other objects of csrc/ are reused; run `make -C montecarlo-risk-engine_amd/csrc` first)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); CS=$ROOT/montecarlo-risk-engine_amd/csrc; F=$1; shift
mkdir -p $ROOT/variants
while [ $# -gt 1 ]; do
  N=$1; FL=$2; shift 2
  ( cd $CS && /opt/rocm/bin/hipcc $FL -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=on -c $F -o /tmp/var_$N.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "kf_leanILi2ELi2ELb0ELi1" | grep "SGPRs\|VGPRs\|Occupancy" | sed "s/.*remark: */  $N: /" ;
    OBJS=$(ls *.o | grep -v "^${F%.hip}.o$"); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/libmcx_$N.so $OBJS /tmp/var_$N.o ) &
done
wait
ls -la $ROOT/variants/
