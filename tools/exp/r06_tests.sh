#!/bin/bash
# round 6: the whole GPU suite with the measured margins on record (OEH_TEST_REPORT) and the new reference-fixture tests' prints
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
O=$ROOT/gpurun_out/r06_tests
mkdir -p $O
rm -f $O/report.tsv
OEH_TEST_REPORT=$O/report.tsv python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids > $O/out.txt
tail -5 $O/out.txt
grep -E "reference fixture|h12|cfg4 " $O/out.txt | head -60
