# round 4: the population products of configs[2] / configs[4] in isolation; 64 x 64 against 64 x 32 tiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for shape in "6144 2048 2048" "6174 2048 2048" "6200 2048 2048" "6250 2048 2048" "6300 2048 2048" "6400 2048 2048" "2048 4096 4096" "1600 4096 4096"; do
  for b in zgemm_bench zgemm_bench_small; do echo -n "$b "; timeout -k 10 60 tools/bin/$b $shape 0 1 10 pop; done
done > $O/pop_tiles.txt 2>&1
cat $O/pop_tiles.txt
