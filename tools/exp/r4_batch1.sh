#!/bin/bash
# round 4, GPU batch 1 (prepared at the end of round 3): PERSIST = 3 -- stores of a tile fused with the loads of the next -- on the
# strided read of the API layout (multi-rank inverse x pass).  Needs an EXPERIMENTS build made on the CPU beforehand:
#   make -C distributedfft_amd/csrc clean && make -j8 -C distributedfft_amd/csrc EXTRA=-DDFFT_EXPERIMENTS && make -C tools kbench
# (and the shipped build again afterwards: make -C distributedfft_amd/csrc clean && make -j8 -C distributedfft_amd/csrc)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b1
mkdir -p $OUT
cd $R
K=$R/tools/kbench
S="variant_ix=1;variant_ix=8;variant_ix=9;variant_ix=1"          # table store (P1 > 1)
S1="variant_ix=1;variant_ix=10;variant_ix=11;variant_ix=1"       # one-block store (P1 = 1: one rank)
{
echo "== 1024^3 fp64 multi-rank path: x^-1 plain (1) | PERSIST 3 + hints (10) | PERSIST 3 (11) | plain"
timeout 200 $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "$S1" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== rank 0 of 2x4"
timeout 100 $K --size 1024 --prec f64 --iters 10 --ranks 2x4 --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== rank 0 of 8x1"
timeout 100 $K --size 1024 --prec f64 --iters 10 --ranks 8x1 --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_persist3.txt 2>&1
cat $OUT/r4_persist3.txt | cut -c1-160
