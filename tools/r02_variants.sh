set -e
MAUS_LU_STREAMS=1 MAUS_PANEL_MW=0 timeout -k 10 200 python tools/panel_clocks.py 32 > gpurun_out/panel_clocks_fast.txt 2>&1
MAUS_LU_STREAMS=1 MAUS_PANEL_MW=0 timeout -k 10 200 python tools/panel_clocks.py 181 >> gpurun_out/panel_clocks_fast.txt 2>&1
cat gpurun_out/panel_clocks_fast.txt
