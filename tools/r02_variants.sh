set -e
for ms in 32 16 8; do for st in 2 3 4; do
echo "## MIN_SUB=$ms STREAMS=$st"
MAUS_LU_MIN_SUB=$ms MAUS_LU_STREAMS=$st timeout -k 10 200 python tools/lu_batch_rates.py 32 64 2>&1 | grep "G="
done; done > gpurun_out/small_streams.txt 2>&1
cat gpurun_out/small_streams.txt
WARM=8 timeout -k 10 300 python tools/host_profile.py > gpurun_out/host_profile_it9.txt 2>&1
head -50 gpurun_out/host_profile_it9.txt | cut -c1-160
