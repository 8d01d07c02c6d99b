#!/bin/bash
# A build of libmgx with extra compile-time switches for mgx_stream.hip (tuning experiments):
#   tools/build_variant.sh <name> "<flags>"   ->  variants/libmgx_<name>.so
# Run it with MGX_LIB=$PWD/variants/libmgx_<name>.so (tools/ab.sh takes a list of them).
set -eo pipefail
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../multigrid_amd/csrc"
mkdir -p ../../variants obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-result \
    $FLAGS -c mgx_stream.hip -o obj/mgx_stream_$NAME.o
OBJS=$(ls obj/mgx_*.o | grep -v "mgx_stream" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS obj/mgx_stream_$NAME.o -lpthread -o ../../variants/libmgx_$NAME.so
echo built variants/libmgx_$NAME.so
