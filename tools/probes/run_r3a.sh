python tools/probes/mfma_peak.py > gpurun_out/r3a_mfma_peak.log 2>&1
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ar --no-extra 2>&1 | cut -c1-1500 > gpurun_out/r3a_bench.log
