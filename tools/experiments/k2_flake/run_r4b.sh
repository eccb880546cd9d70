#!/bin/bash
# Round 4, second step: the r4 variants with their highest registers moved to the top of the allocation (make_r4_top.py)
cd tools/experiments/k2_flake
R=./co_runner.bin
echo "== as built (28 of 32 registers used; 30 of 32 for dual)"
$R r4_co/base256_asbuilt.co 300 256 9 0
echo "== highest registers on the last allocated ones"
$R r4_co/base256_top.co 300 256 9 0
$R r4_co/base256_top.co 300 256 9 70000   # two workgroups per CU
$R r4_co/base256_top.co 300 256 9 100000  # one workgroup per CU
$R r4_co/dual_top.co 300 512 5 0
$R r4_co/dual_top.co 300 512 5 100000     # one 512-thread workgroup per CU: two waves per SIMD, ONE workgroup
$R r4_co/nt128_top.co 300 128 9 0
$R r4_co/nt128_top.co 300 128 9 40000     # four 2-wave workgroups per CU
$R r4_co/nt128_top.co 300 128 9 100000    # one 2-wave workgroup per CU
$R r4_co/nt64_top.co 300 64 9 0
$R r4_co/nt64_top.co 300 64 9 40000       # four 1-wave workgroups per CU (one per SIMD if they spread)
$R r4_co/nt64_top.co 300 64 9 100000      # one wave per CU
echo "== one-wave workgroups WITH an s_barrier instruction"
$R r4_co/nt64_barrier_top.co 300 64 9 0
$R r4_co/nt64_barrier_top.co 300 64 9 40000
