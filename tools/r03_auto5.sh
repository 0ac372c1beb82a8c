cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 400 python tools/host_profile.py c5 3 > $O/host_c5.txt 2>&1; head -40 $O/host_c5.txt | cut -c1-170
