# round 4: host profiles of c5 / c4 loop bodies; outer block width at n = 1024
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 300 python tools/host_profile.py c5 3 > $O/host_c5.txt 2>&1; head -45 $O/host_c5.txt
timeout -k 10 300 python tools/host_profile.py c4 3 > $O/host_c4.txt 2>&1; head -40 $O/host_c4.txt
for nbo in 256 512 1024; do echo "== MAUS_LU_NBO=$nbo"; MAUS_LU_NBO=$nbo LU_N=1024 LU_BATCH_KERNELS=1 timeout -k 10 200 python tools/lu_batch_rates.py 256 271 2>&1 | grep "G="; done > $O/c2_nbo.txt 2>&1; cat $O/c2_nbo.txt
