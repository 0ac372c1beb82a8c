# round 4: two ranks on ONE GPU through the gloo rehearsal transport (the multi-process path end to end on the device library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
MAUS_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-side --no-cpu-baseline > $O/gloo2_gpu.json 2> $O/gloo2_gpu.err
echo "rc=$?"; tail -3 $O/gloo2_gpu.err | cut -c1-200
python - $O/gloo2_gpu.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], d["ms_per_step"], d.get("per_rank"))
PY
