for d in 44 1; do
  MAUS_GEMM_DMA=$d python -m pytest tests/test_gpu_kernels.py -x -q -k "zgemm or gemm" 2>&1 | tail -1
  MAUS_GEMM_DMA=$d TAG=dma$d python tools/gemm_k512_check.py
done
python bench.py --no-cpu-baseline --no-small-batch > gpurun_out/r02_v_dma_mix.json 2>gpurun_out/r02_v_dma_mix.err; python tools/bench_summary.py gpurun_out/r02_v_dma_mix.json | cut -c1-500
