#!/bin/bash
# parity sweeps against the oracle on the FINAL tree of round 6 (fresh seeds): estimateBatch, the Gibbs models, estimate() from teams
# of 256 (batches of up to 256 joined calls) and of 7
out=gpurun_out/r06/sweeps_final; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m tests.fuzz_parity 900 81000 > $out/general_900_from_81000.txt 2>&1; tail -1 $out/general_900_from_81000.txt
timeout 600 python -m tests.fuzz_parity 300 83000 gibbs > $out/gibbs_300_from_83000.txt 2>&1; tail -1 $out/gibbs_300_from_83000.txt
RPVG_FUZZ_TEAM=256 timeout 1200 python -m tests.fuzz_parity 500 85000 > $out/general_500_from_85000_through_estimate_team_of_256.txt 2>&1; tail -1 $out/general_500_from_85000_through_estimate_team_of_256.txt
RPVG_FUZZ_TEAM=7 timeout 600 python -m tests.fuzz_parity 200 87000 gibbs > $out/gibbs_200_from_87000_through_estimate_team_of_7.txt 2>&1; tail -1 $out/gibbs_200_from_87000_through_estimate_team_of_7.txt
timeout 600 python -m tests.fuzz_rows 150 88000 > $out/rows_150_from_88000.txt 2>&1; tail -1 $out/rows_150_from_88000.txt
grep -h "MISMATCH" $out/*.txt | cut -c1-260
