cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in ${VARIANTS:-gbase}; do echo "== $v"; MPCQP_LIB=$PWD/qpmpc_amd/lib/ab/$v.so timeout 120 python tools/probe_general_phases.py 20 6 40 4 2>&1 | grep -v amdgpu.ids | tail -13; done
