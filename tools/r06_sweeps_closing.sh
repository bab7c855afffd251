#!/bin/bash
# closing sweeps of round 6 (the tree with the direct single-path matrices and six Gibbs batches in flight): fresh seeds
out=gpurun_out/r06/sweeps_closing; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m tests.fuzz_parity 700 96000 > $out/general_700_from_96000.txt 2>&1; tail -1 $out/general_700_from_96000.txt
timeout 500 python -m tests.fuzz_parity 400 97000 gibbs > $out/gibbs_400_from_97000.txt 2>&1; tail -1 $out/gibbs_400_from_97000.txt
RPVG_FUZZ_TEAM=128 timeout 400 python -m tests.fuzz_parity 300 98000 > $out/general_300_from_98000_through_estimate_team_of_128.txt 2>&1; tail -1 $out/general_300_from_98000_through_estimate_team_of_128.txt
grep -h "MISMATCH" -A1 $out/*.txt | cut -c1-260
