#!/bin/bash
# tools/mk_variant.sh NAME "FILE[,FILE...]" [EXTRA FLAGS]: libgrhip_NAME.so = the diagnostic build's objects (build_diag/,
# refreshed for sources newer than their object) with the named csrc files recompiled under the extra flags.  A/B runs.
set -e
cd "$(dirname "$0")/../gnuradio-3.5.0-dmr_amd"
NAME=$1; FILES=$2; shift 2
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result -DGRHIP_DIAG"
mkdir -p build_diag build_$NAME
( flock 9
  for f in csrc/*.hip; do
    o=build_diag/$(basename $f .hip).o
    if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find csrc ../include -name '*.h' -newer $o -print -quit)" ]; then $CC -c $f -o $o & fi
  done; wait ) 9>build_diag/.lock
OBJS=""
for f in csrc/*.hip; do
  b=$(basename $f .hip)
  if echo ",$FILES," | grep -q ",$b,"; then $CC "$@" -c $f -o build_$NAME/$b.o; OBJS="$OBJS build_$NAME/$b.o"; else OBJS="$OBJS build_diag/$b.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgrhip_$NAME.so $OBJS
echo built libgrhip_$NAME.so
