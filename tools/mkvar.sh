#!/bin/bash
# usage: mkvar.sh name -DFLAG=1 ...
name=$1; shift
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC "$@" -c helicon_amd/csrc/helicon_hip.hip -o build_var/$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/$name.so build_var/$name.o helicon_amd/csrc/_obj/gen_rows_*.o && rm build_var/$name.o && echo built $name
