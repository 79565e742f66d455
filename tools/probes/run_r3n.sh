python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "rmsnorm" 2>&1 | tail -2
python tools/elementwise_bench.py 2>&1 | grep -v amdgpu
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ar --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); [print(k, v['ms_per_step']) for k,v in d['roofline']['by_family'].items()]"
