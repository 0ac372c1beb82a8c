python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r02_gputests_full.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r02_gputests_full.log | cut -c1-250 | head -30
python -c "import __graft_entry__ as g; g.smoke()"
