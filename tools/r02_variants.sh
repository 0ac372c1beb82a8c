python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r02_gputests2.log 2>&1
tail -40 gpurun_out/r02_gputests2.log
