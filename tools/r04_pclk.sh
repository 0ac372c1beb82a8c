cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for g in 256 32; do LU_N=1024 MAUS_LIB=tools/bin/libmaus_hip_clk.so timeout -k 10 120 python tools/panel_clocks.py $g; done > $O/pclk_1024.txt 2>&1
for g in 181 32; do MAUS_LIB=tools/bin/libmaus_hip_clk.so timeout -k 10 200 python tools/panel_clocks.py $g; done > $O/pclk_4096.txt 2>&1
cat $O/pclk_1024.txt $O/pclk_4096.txt
