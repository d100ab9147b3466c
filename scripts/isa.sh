#!/bin/bash
# scripts/isa.sh <file.hip> <mangled-name substring> [extra hipcc flags] : the gfx950 ISA of ONE kernel of a source file on stdout,
# compiled with the product's flags (deblur4dgs_amd/build.py: FLAGS + FILE_FLAGS).  Round 5's instruction diet of the composites was done
# with this listing in hand: count the VALU instructions of the inner loop, look for exec-masked loads, duplicated compares, 64-bit mads,
# v_readlane / v_writelane SGPR spills.  Example: scripts/isa.sh raster_bwd.hip k_raster_bwd_q8ILi3ELb1E | less
f=$1; pat=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
extra=$(python3 - "$f" <<'PY'
import sys
sys.path.insert(0, ".")
from deblur4dgs_amd.build import FILE_FLAGS
print(" ".join(FILE_FLAGS.get(sys.argv[1], [])))
PY
)
t=$(mktemp /tmp/isa_XXXX.s)
(cd $here && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $extra "$@" -Iinclude -Ideblur4dgs_amd/csrc -S --cuda-device-only deblur4dgs_amd/csrc/$f -o $t 2>/dev/null)
start=$(grep -n "^_Z.*${pat}.*:" $t | head -1 | cut -d: -f1)
[ -z "$start" ] && { echo "no kernel matching $pat in $f" >&2; grep -o "^_Z[A-Za-z0-9_]*" $t | sort -u | head -40 >&2; rm -f $t; exit 1; }
end=$(awk -v s=$start 'NR>s && /s_endpgm/ {print NR; exit}' $t)
sed -n "${start},${end}p" $t
awk -v s=$end 'NR>s && NR<s+60' $t | grep -E "NumVgprs|NumAgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize" | head -6
rm -f $t
