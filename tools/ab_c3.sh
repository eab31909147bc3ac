#!/bin/bash
# Dev tool: config-3 A/B over library variants built by tools/ab_unit.sh. usage: tools/ab_c3.sh tag [tag ...]  ("main" = the library)
for t in "$@"; do
  if [ "$t" = main ]; then unset MPCQP_LIB; else export MPCQP_LIB=$PWD/qpmpc_amd/lib/ab/$t.so; fi
  echo "== $t"
  python tools/bench_pipe.py 1024 50 2>&1 | grep -v amdgpu.ids | cut -c1-70
  python tools/probe_config3_loop.py 2>&1 | grep -A4 "^reuse" | grep "u0"
done
