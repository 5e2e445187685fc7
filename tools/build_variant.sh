#!/bin/bash
# A/B build of the library with extra compiler flags for a few translation units (CPU; the result travels to the GPU box in xlb_amd/lib/):
#   tools/build_variant.sh NAME "-DXLB_KBC_GAMMA32=0" step_d3q27_kbc_fast step2_d3q27
# builds the default library first (so that every object is current), copies its objects, rebuilds only the named ones with the
# flags and links xlb_amd/lib/NAME.so.
set -e
NAME=$1; FLAGS=$2; shift 2
make -j8 all > /dev/null
rm -rf build/obj_$NAME
cp -rp build/obj build/obj_$NAME
for o in "$@"; do rm -f build/obj_$NAME/$o.o; done
make -j8 OBJDIR=build/obj_$NAME LIB=xlb_amd/lib/$NAME.so EXTRA="$FLAGS" xlb_amd/lib/$NAME.so 2>&1 | grep -E "error|hipcc.* -c " | sed -E 's/.*-c ([^ ]+).*/  rebuilt \1/' || true
ls -la xlb_amd/lib/$NAME.so
