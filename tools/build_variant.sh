#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : memo_amd/libmemo_amd_NAME_ab.so = the AB library with
# memo_sweep_cons.hip, memo_sweep_cons3t.hip and memo_sweep_memb.hip compiled with the given flags (ablations, A/B of kernel variants);
# tools read it through MEMO_AMD_AB_LIB=memo_amd/libmemo_amd_NAME_ab.so
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../memo_amd/csrc"
make -s -j6 >/dev/null
CXX="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -I../../include"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $FLAGS -c memo_sweep_cons.hip -o /tmp/cons_$NAME.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $FLAGS -c memo_sweep_memb.hip -o /tmp/memb_$NAME.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $FLAGS -c memo_sweep.hip -o /tmp/sweep_$NAME.o &      # (fill_args, the stamp buffer)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $FLAGS -c memo_debug.hip -o /tmp/debug_$NAME.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $FLAGS -c memo_sweep_cons3t.hip -o /tmp/cons3t_$NAME.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libmemo_amd_${NAME}_ab.so \
  /tmp/sweep_$NAME.o /tmp/cons_$NAME.o /tmp/cons3t_$NAME.o /tmp/memb_$NAME.o memo_interleave.o memo_view.o memo_index.o memo_hostpack.o memo_hostcore.o memo_multi.o memo_transport.o memo_sort.o memo_dap.o memo_emit.o /tmp/debug_$NAME.o
echo built libmemo_amd_${NAME}_ab.so
