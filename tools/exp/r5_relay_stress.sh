cd $GRAFT_REPO_ROOT
for cfg in "vmm 0" "vmm 1" "vmm 2"; do set -- $cfg; echo "== staging=$1 DFFT_VMM_KEEP_VA=$2 (0: unmap + free the address range, 1: unmap, keep the range reserved, 2: unmap, synchronize, free)"; DFFT_RELAY_STAGING=$1 DFFT_VMM_KEEP_VA=$2 python tools/exp/r5_relay_stress.py 8 2>&1 | grep -v "direct vs\|amdgpu.ids" | grep "66, 50\|chunks=3"; done
