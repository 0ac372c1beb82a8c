cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python tools/body_probe.py 11 > $O/body_probe_c2.txt 2>&1; tail -5 $O/body_probe_c2.txt
