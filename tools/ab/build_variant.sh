#!/bin/bash
# usage: tools/ab/build_variant.sh NAME SRC.hip "-DFOO=1 ..." : libdanbo_hip with ONE translation unit rebuilt with extra defines -> tools/ab/libdanbo_hip_NAME.so
# (A/B of a kernel variant on the GPU box: DANBO_HIP_LIB=$PWD/tools/ab/libdanbo_hip_NAME.so python tools/micro_....py)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
cd $ROOT/danbo-pytorch_amd/csrc
make -s >/dev/null
mkdir -p build/variants
obj=build/variants/$1_${2%.hip}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $3 -c -o $obj $2
objs=$(ls build/*.o | grep -v "build/${2%.hip}.o")
hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/tools/ab/libdanbo_hip_$1.so $objs $obj
echo built tools/ab/libdanbo_hip_$1.so
