#!/bin/bash
# scripts/regs.sh <file.hip> [waves_per_eu] [kernel-name-filter]: VGPR / scratch / SGPR-spill summary of a kernel source, optionally
# with __attribute__((amdgpu_waves_per_eu(w))) forced on every `__launch_bounds__(BLK)` kernel of the file (compile-only probe)
f=$1; w=${2:-0}; pat=${3:-k_}
t=$(dirname $f)/_regs_tmp.hip
if [ "$w" = 0 ]; then cp $f $t; else sed "s/__launch_bounds__(BLK) k_/__launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu($w))) k_/" $f > $t; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -c $t -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|  VGPRs:|AGPRs:|ScratchSize|SGPRs Spill|Occupancy" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste -sd' ' | sed 's/Function Name:/\n/g' | grep "$pat"
rm -f $t
