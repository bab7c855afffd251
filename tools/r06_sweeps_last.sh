#!/bin/bash
# the last sweeps of round 6, on the tree that is handed over (fresh seeds): estimateBatch, estimate() from a team of 64, the Gibbs models
out=gpurun_out/r06/sweeps_last; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m tests.fuzz_parity 1000 90000 > $out/general_1000_from_90000.txt 2>&1; tail -1 $out/general_1000_from_90000.txt
RPVG_FUZZ_TEAM=64 timeout 900 python -m tests.fuzz_parity 500 92000 > $out/general_500_from_92000_through_estimate_team_of_64.txt 2>&1; tail -1 $out/general_500_from_92000_through_estimate_team_of_64.txt
timeout 600 python -m tests.fuzz_parity 300 94000 gibbs > $out/gibbs_300_from_94000.txt 2>&1; tail -1 $out/gibbs_300_from_94000.txt
grep -h "MISMATCH" -A1 $out/*.txt | cut -c1-260
