#!/bin/bash
# A build of libmgx whose small-level kernels leave time marks (tools/trace_small_levels.py):
#   tools/build_trace.sh   ->  variants/libmgx_trace.so   (the other objects come from the regular build)
set -eo pipefail
cd "$(dirname "$0")/../multigrid_amd/csrc"
mkdir -p ../../variants obj
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-result"
$CC -DMGX_TAIL_TRACE -c mgx_tail.hip -o obj/trace_tail.o
$CC -DMGX_MID_TRACE -c mgx_mid.hip -o obj/trace_mid.o
OBJS=$(ls obj/mgx_*.o | grep -v "mgx_tail.o" | grep -v "mgx_mid.o" | grep -v "mgx_stream_" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS obj/trace_tail.o obj/trace_mid.o -lpthread -o ../../variants/libmgx_trace.so
echo built variants/libmgx_trace.so
