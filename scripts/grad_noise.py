"""Run-to-run spread of the HIP gradients against the oracle on the ill-conditioned parity case (large Gaussians).

    gpurun -- 'python scripts/grad_noise.py 12'
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import parity_utils as pu
from oracle import cpu_oracle as orc
import sugar_amd.synthetic as syn
from tests.test_gpu_parity import GRAD_NAMES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = syn.make_scene(3000, 11, 0.01, 0.3)
cam = syn.orbit_cameras(250, 190)[2]
bg = torch.tensor([0.1, 0.2, 0.3])
st = pu.run_oracle(scene, cam, bg)
g = np.random.default_rng(0).standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
gr = orc.backward(st, g)
for it in range(n):
    hp = pu.run_hip(scene, cam, bg, grad_out=g)
    out = []
    for k, v in hp["grads"].items():
        ref = gr[GRAD_NAMES[k]]
        e = pu.rel_stats(v.reshape(ref.shape), ref)
        out.append(f"{k}={e['norm_rel']:.2e}")
    print(it, " ".join(out))
