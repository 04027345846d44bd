cd $GRAFT_REPO_ROOT
for cfg in "vmm_leak 0" "vmm 0"; do set -- $cfg; echo "== staging=$1 poison=$2"; DFFT_RELAY_STAGING=$1 DFFT_RELAY_POISON=$2 python tools/exp/r5_relay_stress.py 10 2>&1 | grep -v "direct vs\|amdgpu.ids"; done
