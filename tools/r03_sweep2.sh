set -e
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gputests2.txt 2>&1 || true
tail -15 $O/gputests2.txt
