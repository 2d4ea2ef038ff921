#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS" [sl|main|noslp] -- a library with ONE kernel unit compiled with extra -D flags -- the scalar-load
# unit (nb_nbody_sl.inc, nb_nbody_sym.inc; the default) the main unit (block chain, LDS-tiled kernels, FAST wave form) or the SLP-off unit (boids, j-parallel STRICT) --,
# linked with the current objects of the other units: build/variants/NAME.so (load it with NENBODY_LIB=...).
# For A/B measurements of kernel experiments; `make -C nenbody_amd/csrc` first.
set -e
NAME=$1; EXTRA=$2; UNIT=${3:-sl}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/variants"
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
cd "$ROOT/nenbody_amd/csrc"
if [ "$UNIT" = sl ]; then
    /opt/rocm/bin/hipcc $FLAGS -DNBK_SL_TU -mllvm -enable-misched=false -mllvm -enable-post-misched=false $EXTRA -c -o "$ROOT/build/variants/$NAME.o" nb_kernels.hip
    /opt/rocm/bin/hipcc $FLAGS -shared -o "$ROOT/build/variants/$NAME.so" "$ROOT/build/obj/nb_kernels.o" "$ROOT/build/obj/nb_kernels_noslp.o" "$ROOT/build/variants/$NAME.o" "$ROOT/build/obj/nb_api.o"
elif [ "$UNIT" = noslp ]; then
    /opt/rocm/bin/hipcc $FLAGS -DNBK_NOSLP_TU -fno-slp-vectorize $EXTRA -c -o "$ROOT/build/variants/$NAME.o" nb_kernels.hip
    /opt/rocm/bin/hipcc $FLAGS -shared -o "$ROOT/build/variants/$NAME.so" "$ROOT/build/obj/nb_kernels.o" "$ROOT/build/variants/$NAME.o" "$ROOT/build/obj/nb_kernels_sl.o" "$ROOT/build/obj/nb_api.o"
else
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o "$ROOT/build/variants/$NAME.o" nb_kernels.hip
    /opt/rocm/bin/hipcc $FLAGS -shared -o "$ROOT/build/variants/$NAME.so" "$ROOT/build/variants/$NAME.o" "$ROOT/build/obj/nb_kernels_noslp.o" "$ROOT/build/obj/nb_kernels_sl.o" "$ROOT/build/obj/nb_api.o"
fi
rm -f "$ROOT/build/variants/$NAME.o"
echo "built build/variants/$NAME.so"
