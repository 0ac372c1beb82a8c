# round 3, call 1: per-kernel-class times of one maus_shifted_lu_solve call (single stream) at several batch sizes, NBO sweep
set -e
O=gpurun_out/r03
mkdir -p $O
for nbo in 512 384 256; do
  echo "## MAUS_LU_NBO=$nbo" >> $O/sweep1.txt
  MAUS_LU_NBO=$nbo MAUS_LU_STREAMS=1 LU_BATCH_KERNELS=1 timeout -k 10 240 python tools/lu_batch_rates.py 32 64 181 256 >> $O/sweep1.txt 2>> $O/sweep1.err
done
echo "## default streams (tuner), NBO 512" >> $O/sweep1.txt
timeout -k 10 240 python tools/lu_batch_rates.py 32 64 128 181 256 331 >> $O/sweep1.txt 2>> $O/sweep1.err
cat $O/sweep1.txt
