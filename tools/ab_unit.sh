#!/bin/bash
# Dev tool: build variants of ONE unit of the library with extra compiler flags into qpmpc_amd/lib/ab/<tag>.so (all other
# objects are reused), for A/B runs on the GPU box with MPCQP_LIB=<that file>.
# usage: tools/ab_unit.sh mpcqp_pair tag "-DKNOB=1 -mllvm -some-option" [tag2 "flags2" ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
unit=$1; shift
mkdir -p $R/qpmpc_amd/lib/ab
OBJS=$(ls -t $R/qpmpc_amd/lib/obj/*.o | grep -v "/${unit}\." | awk -F/ '{split($NF,a,"."); if (!(a[1] in seen)) {seen[a[1]]=1; print}}')
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -I$R/include -I$R/qpmpc_amd/csrc -c $R/qpmpc_amd/csrc/$unit.hip -o $R/qpmpc_amd/lib/ab/$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/qpmpc_amd/lib/ab/$tag.o -o $R/qpmpc_amd/lib/ab/$tag.so && rm $R/qpmpc_amd/lib/ab/$tag.o && echo built $tag ) &
done
wait
