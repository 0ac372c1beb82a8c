# round 4: does the LU's operand form (tile-major, rows through the permutation) cost the K >= 256 update anything?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
T=${1:-zlu}
mkdir -p $O
for shape in "512 544 512 1088 256" "3584 3616 512 4160 96" "2048 2080 512 4160 181" "1536 1568 512 4160 181"; do
  set -- $shape
  timeout -k 10 60 tools/bin/zgemm_bench $shape 5
  timeout -k 10 60 tools/bin/zgemm_bench $shape 5 lu
  timeout -k 10 60 tools/bin/zgemm_bench $shape 5 lu perm
done > $O/${T}.txt 2>&1
cat $O/${T}.txt
