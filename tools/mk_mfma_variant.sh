#!/bin/bash
# tools/mk_mfma_variant.sh NAME "EXTRA FLAGS": libgrhip_NAME.so = the diagnostic build (make variant NAME=diag
# EXTRA=-DGRHIP_DIAG) with csrc/fir_mfma.hip recompiled under the extra flags.  For A/B runs of the headline kernel.
set -e
cd "$(dirname "$0")/../gnuradio-3.5.0-dmr_amd"
NAME=$1; shift
mkdir -p build_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function \
  -Wno-unused-result -DGRHIP_DIAG "$@" -c csrc/fir_mfma.hip -o build_$NAME/fir_mfma.o
OBJS=$(ls build_diag/*.o | grep -v fir_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgrhip_$NAME.so $OBJS build_$NAME/fir_mfma.o
echo built libgrhip_$NAME.so
