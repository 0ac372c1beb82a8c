# round 3: does the sub-batch stream split still pay with the tile-major workspace?  Driver-shaped bench at fixed stream counts.
set -e
O=gpurun_out/r03
mkdir -p $O
for s in 1 2; do
  MAUS_LU_STREAMS=$s timeout -k 10 300 python bench.py --no-cpu-baseline --no-small-batch --no-isolated > $O/streams_$s.json 2> $O/streams_$s.err
  python - $O/streams_$s.json $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("streams", sys.argv[2], "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "k256", round(d["roofline"]["achieved"],1), d["per_step_summary"])
PY
done
