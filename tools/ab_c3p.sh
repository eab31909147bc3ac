#!/bin/bash
# Dev tool: config-3 sweep phases (reuse mode) over library variants. usage: tools/ab_c3p.sh tag [tag ...]
for t in "$@"; do
  if [ "$t" = main ]; then unset MPCQP_LIB; else export MPCQP_LIB=$PWD/qpmpc_amd/lib/ab/$t.so; fi
  echo "== $t"
  python tools/probe_config3_loop.py ${B:-1024} 2>&1 | grep -A4 "^reuse" | grep "u0"
done
