"""MAGVITv2.get_code on 16 images of 256^2 alone (nothing else on the GPU), and beside a flat AdamW update of the 1.5B backbone
on a side stream (what the training step does): how much the overlap costs each side.
Measured (round 2): get_code alone 19.4-19.7 ms, the update alone 8.3-9.6 ms (whole chip) / 7.7-8.8 ms (256 workgroups);
beside each other 34.7 / 28.7 / 25.2 / 25.1 / 25.6 / 27.3 ms at 64 / 96 / 128 / 192 / 256 / 512 workgroups -- the overlap hides about
2.5 of the update's 8 ms whatever its width, because the convolutions' loads queue behind the update's HBM stream.  Launching
the update (whole chip) from inside the encoder after its 256^2 / 128^2 / 64^2 / 32^2 level: 27.6-27.7 ms each, i.e. the serial sum."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from models import MAGVITv2
from bench import init_magvit_device

dev = torch.device("cuda:0")
vq = MAGVITv2().to(dev).eval().requires_grad_(False)
init_magvit_device(vq, 10084)
g = torch.Generator(device=dev).manual_seed(1)
images = torch.rand(16, 3, 256, 256, device=dev, generator=g) * 2 - 1


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"get_code(16 x 256^2) alone: {timed(lambda: vq.get_code(images)):.2f} ms", flush=True)
if os.environ.get("ONLY_GET_CODE") == "1":       # kernel-trace runs: rocprofv3 --kernel-trace --stats -- python3 tools/tokenizer_bench.py
    sys.exit(0)
from unigen_hip import ops
n = 1_543_000_000
p, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
b16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()


from unigen_hip import lib as _l


def adam(blocks):
    _l.check(_l.load().ug_adamw_flat(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), b16.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 0.01,
                                     3, 1.0, blocks, torch.cuda.current_stream().cuda_stream), "ug_adamw_flat")


print(f"flat AdamW alone (whole chip): {timed(lambda: adam(0)):.2f} ms;  256 workgroups: {timed(lambda: adam(256)):.2f} ms", flush=True)


def both(blocks):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam(blocks)
    vq.get_code(images)
    torch.cuda.current_stream().wait_stream(side)


for blocks in (32, 64, 96, 128, 192, 256, 512):
    print(f"get_code beside the AdamW update ({blocks:3d} workgroups on a side stream): {timed(lambda: both(blocks)):.2f} ms "
          f"(the update alone at that width: {timed(lambda: adam(blocks)):.2f} ms)", flush=True)

