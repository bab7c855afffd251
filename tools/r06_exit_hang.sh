#!/bin/bash
# runs the reference-shaped factory binary (tests/cpp/reference_factory.cpp) many times: a run that does not end in 20 s shows where it stands
R=$GRAFT_REPO_ROOT
cd $R
python -c "from tests import small_cases; print(small_cases.build_reference_factory())"
b=tests/cpp/_build/reference_factory
hung=0
for i in $(seq 1 ${RUNS:-60}); do
  for m in haplotypes haplotype-transcripts; do
    RPVG_AMD_TRACE_EXIT=1 timeout 20 $b $m > /tmp/out.txt 2> /tmp/err.txt
    rc=$?
    if [ $rc -ne 0 ]; then hung=$((hung+1)); echo "run $i $m rc $rc"; tail -8 /tmp/err.txt; fi
  done
done
echo "runs that did not end: $hung"
