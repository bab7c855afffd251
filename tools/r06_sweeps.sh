#!/bin/bash
# large parity sweeps against the oracle on the final kernels of round 6: estimateBatch, the Gibbs models, estimate() from teams
out=gpurun_out/r06/sweeps; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m tests.fuzz_parity 800 70000 > $out/general_800_from_70000.txt 2>&1; tail -1 $out/general_800_from_70000.txt
timeout 1200 python -m tests.fuzz_parity 250 72000 gibbs > $out/gibbs_250_from_72000.txt 2>&1; tail -1 $out/gibbs_250_from_72000.txt
RPVG_FUZZ_TEAM=64 timeout 2400 python -m tests.fuzz_parity 400 74000 > $out/general_400_from_74000_through_estimate_team_of_64.txt 2>&1; tail -1 $out/general_400_from_74000_through_estimate_team_of_64.txt
RPVG_FUZZ_TEAM=5 timeout 1200 python -m tests.fuzz_parity 150 76000 gibbs > $out/gibbs_150_from_76000_through_estimate_team_of_5.txt 2>&1; tail -1 $out/gibbs_150_from_76000_through_estimate_team_of_5.txt
timeout 900 python -m tests.fuzz_rows 80 78000 > $out/rows_80_from_78000.txt 2>&1; tail -1 $out/rows_80_from_78000.txt
