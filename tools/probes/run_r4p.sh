cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
UNIGEN_ADAMW_SPREAD=1 python -m pytest tests/test_model_gpu.py -x -q -k "adamw or loss_curve" 2>&1 | tail -2
for v in 0 1 0 1; do echo "SPREAD=$v: $(UNIGEN_ADAMW_SPREAD=$v python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "ms/step; fwd_bwd", r["fwd_bwd_1p5b"]["ms"], "ms; gemm", r["gemm_ms_per_step"], "fam", {k: v["ms_per_step"] for k, v in r["by_family"].items()}, "loss", d["loss_first_last"])')"; done | tee gpurun_out/r4p_adamw_spread.txt
