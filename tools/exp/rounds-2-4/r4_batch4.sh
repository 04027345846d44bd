#!/bin/bash
# round 4, GPU batch 4
# 1. placement: is the slow case "virtually consecutive chunks are physically consecutive" (rows 16 MiB apart land in the same banks)?
#    fresh processes, interleaved: hipMalloc | vmm 1024 | vmm 2 | vmm 2 shuffled | vmm 64 shuffled, three rounds, right after a
#    64 GiB allocation has come and gone (the state in which batch 3 saw the default backing run 37 ms)
# 2. chunkprobe: run size and store shape of a 256 KiB-per-workgroup copy
# 3. the MPI form of the CPU oracle on this host: what the environment allows (cores, cgroup quota, /dev/shm) and timings at 32 .. 128 ranks
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b4
mkdir -p $OUT
cd $R
KS=$R/tools/kbench
{
echo "== churn: one 2048^3 fp32 run (64 GiB buffers) first"
timeout 200 $KS --size 2048 --prec f32 --iters 1 2>&1 | grep -E "^PLAN" | cut -c1-120
for rep in 1 2 3; do
echo "== round $rep: 1024^3 fp64 C2C, fresh processes: hipMalloc | vmm 1024 | vmm 2 | vmm 2 shuffled | vmm 64 shuffled | vmm 16 | vmm 16 shuffled"
timeout 100 $KS --size 1024 --prec f64 --iters 8 2>&1 | grep -E "^PLAN|total" | cut -c1-140
for v in "1024" "2" "2 --shuffle" "64 --shuffle" "16" "16 --shuffle"; do
  timeout 100 $KS --size 1024 --prec f64 --iters 8 --vmm $v 2>&1 | grep -E "^PLAN|total" | cut -c1-140
done
done
} > $OUT/r4_placement_shuffle.txt 2>&1
grep -E "^==|PLAN|total" $OUT/r4_placement_shuffle.txt | paste - - 2>/dev/null | cut -c1-230
timeout 300 $R/tools/chunkprobe 4 > $OUT/r4_chunkprobe.txt 2>&1; cat $OUT/r4_chunkprobe.txt
{
echo "nproc $(nproc); affinity $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))'); OMP_NUM_THREADS=${OMP_NUM_THREADS:-unset}"
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cfs quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
df -h /dev/shm | tail -1
free -g | head -2
cd $R/oracle
for cfg in "32 4 8 256" "64 8 8 256" "128 8 16 256" "64 8 8 512" "128 8 16 512" "64 8 8 1024" "128 8 16 1024"; do
  set -- $cfg
  t0=$(date +%s.%N)
  OMP_NUM_THREADS=1 timeout 120 /opt/conda/bin/mpiexec -n $1 ./mpi_pencil $4 $2 $3 1 2>&1 | tail -1 | cut -c1-200
  echo "   ranks $1 grid $4^3: wall $(echo "$(date +%s.%N) - $t0" | bc) s"
done
} > $OUT/r4_mpi_probe.txt 2>&1
cat $OUT/r4_mpi_probe.txt
