# round 3: whole GPU suite, the driver-shaped bench, the other BASELINE configurations, and the PMC counter names of this box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > $O/full_tests.txt 2>&1
echo "pytest rc=$?" >> $O/full_tests.txt
tail -6 $O/full_tests.txt
