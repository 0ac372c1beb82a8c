cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
MAUS_POPGET_TIMING=1 SPLIT_SYNC=1 timeout -k 10 300 python tools/host_profile.py c4 1 > $O/host_c4.txt 2>&1; head -24 $O/host_c4.txt
