# round 4: the population products of configs[2] / configs[4] in isolation (profiles/r04_population_products.txt; the
# 64 x 64 against 64 x 32 comparison in that file came from a second build of the tool with a temporary threshold switch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
mkdir -p tools/bin && hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/zgemm_bench tools/zgemm_bench.hip
for shape in "6174 2048 2048" "6144 2048 2048" "512 4096 4096" "527 4096 4096" "640 4096 4096" "677 4096 4096" "256 4096 4096" "128 8192 8192"; do
  for mode in "pop" "pop conjb" "pop plain" "pop plain conja"; do timeout -k 10 60 tools/bin/zgemm_bench $shape 0 1 10 $mode; done
done > $O/pop_products.txt 2>&1
cat $O/pop_products.txt
