#!/bin/bash
# the reference's validation job line (jobs/argon/pencil/validation.json) on the MI355X library, next to what the reference's own logs say
cd "$(dirname "$0")/../.." || exit 1
make -s -C tools drivers
L=/tmp/mpilib; mkdir -p $L; for l in libmpi.so.12 libgfortran.so.4 libquadmath.so.0; do [ -e /opt/conda/lib/$l ] && ln -sf /opt/conda/lib/$l $L/$l; done
export LD_LIBRARY_PATH=/opt/rocm/lib:$L
for g in "128 128 128" "256 256 256" "512 512 512" "512 1024 1024" "1024 1024 1024"; do
  set -- $g
  for o in 0 1; do
    r=$(/opt/conda/bin/mpiexec -n 4 tools/pencil -nx $1 -ny $2 -nz $3 -p1 2 -p2 2 -o $o -t 4 -w 1 -i 0 -d -b /tmp/t4bench 2>&1 | grep Result | tr '\n' ' ')
    echo "pencil 2x2 opt=$o ${1}x${2}x${3}: $r"
  done
  r=$(/opt/conda/bin/mpiexec -n 4 tools/slab -nx $1 -ny $2 -nz $3 -o 1 -t 4 -w 1 -i 0 -d -b /tmp/t4bench 2>&1 | grep Result | tr '\n' ' ')
  echo "slab 4 opt=1 ${1}x${2}x${3}: $r"
done
python - <<'PY'
import json
r = json.load(open("tests/golden/ref_testcase4_results.json"))
print("\nthe reference's own logs (tests/golden/ref_testcase4_results.json):")
for k, v in r.items():
    if "opt=1 seq=ZY_Then_X" in k:
        print(" ", k, [(e["avg"], e["max"]) for e in v])
PY
