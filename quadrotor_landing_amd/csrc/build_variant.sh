#!/bin/bash
# build_variant.sh NAME "TUS" "FLAGS": a second build of the library in which the translation units TUS (e.g. "tu_quad tu_step")
# are compiled with FLAGS replacing the default scheduler / vectoriser flags (kernel tuning A/B runs:
# QLE_LIB=quadrotor_landing_amd/libqle_ekf_NAME.so).  Needs the default build in build/.
set -e
cd "$(dirname "$0")"
name=$1; tus=$2; shift 2
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*"
mkdir -p build_$name
excl=""
for t in $tus; do
  /opt/rocm/bin/hipcc $F -DQLE_TU_T=float -c -o build_$name/${t}_f32.o $t.hip &
  /opt/rocm/bin/hipcc $F -DQLE_TU_T=double -c -o build_$name/${t}_f64.o $t.hip &
  excl="$excl -e ${t}_f"
done
wait
objs=$(ls build/*.o | grep -v $excl)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libqle_ekf_$name.so $objs build_$name/*.o
echo built ../libqle_ekf_$name.so
