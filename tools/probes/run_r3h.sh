python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "attention or attn or L771 or L1603" > gpurun_out/r3h_tests.log 2>&1; tail -3 gpurun_out/r3h_tests.log
for i in 1 2; do
echo "== old (round-2 mapping)"; UNIGEN_HIP_LIB=$PWD/tools/probes/_build/libunigen_hip_full.so python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
echo "== new (XCD-aware mapping)"; python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ar --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['by_family']['attention'])"
