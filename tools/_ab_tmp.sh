cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in ${VARIANTS:-base}; do
  for b in ${BATCHES:-8192}; do MPCQP_LIB=$PWD/qpmpc_amd/lib/ab/$v.so timeout 300 python bench.py --config 5 --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v batch $b', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms', 'err', (d.get('accuracy') or {}).get('max_rel_err_vs_oracle'))"; done; done
