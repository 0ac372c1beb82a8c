cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for nbo in 512 256 384 768 1024; do echo "== MAUS_LU_NBO=$nbo"; MAUS_LU_NBO=$nbo LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 181 2>&1 | grep "G="; done > gpurun_out/r04/nbo_rates.txt 2>&1; cat gpurun_out/r04/nbo_rates.txt
