#!/bin/bash
# usage: tools/build_variant.sh NAME FILE.hip "extra hipcc flags"  -> variants/lib_NAME.so (FILE recompiled with the flags, the
# other objects taken from aware_amd/csrc/build).  Timing experiments only.
set -e
NAME=$1; FILE=$2; EXTRA=$3
cd "$(dirname "$0")/.."
mkdir -p variants/obj
OBJ=variants/obj/${NAME}_${FILE%.hip}.o
FL="-O3 --offload-arch=gfx950 -std=c++17 -fPIC"
[ "$FILE" = dsp_stream.hip ] && FL="$FL -fno-slp-vectorize"
hipcc $FL $EXTRA -c aware_amd/csrc/$FILE -o $OBJ
OTHERS=$(ls aware_amd/csrc/build/*.o | grep -v "/${FILE%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$NAME.so $OBJ $OTHERS
echo built variants/lib_$NAME.so
