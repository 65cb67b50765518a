#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." [file.hip ...]: rebuild the named translation units (default gemm2.hip)
# with extra flags and link spn4cir_amd/libspn4cir_hip_NAME.so from them + the regular objects.
# Use with SPN_LIB_PATH=spn4cir_amd/libspn4cir_hip_NAME.so.
set -e
cd "$(dirname "$0")/../spn4cir_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-gemm2.hip}
make -j8 > /dev/null
mkdir -p build_$NAME
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  if echo " $FILES " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c $b.hip -o build_$NAME/$b.o
    OBJS="$OBJS build_$NAME/$b.o"
  else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libspn4cir_hip_$NAME.so $OBJS
echo built spn4cir_amd/libspn4cir_hip_$NAME.so
