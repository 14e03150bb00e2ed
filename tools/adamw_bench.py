"""Time of the fused clip + AdamW + re-pack over the Hiera-L arena (hipGraph of 5 steps); SPG_LIBRARY selects the build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd.models import SPEGNet
from spegnet_amd.engine.arena import Arena
from nt_check import timeit
m = SPEGNet({"encoder": {"variant": "large"}, "compute_dtype": "bf16", "init_seed": 0}).cuda().train()
arena = Arena(m)
m.mark_params_changed()
arena.set_hyper(1e-4, 1e-5, 0.05)
eng = m.engine
arena.g.normal_(0, 1e-3)
def step():
    arena._clean = False
    arena.step(1.0, packer=eng)
t = timeit(step, iters=5)
print(f"{os.path.basename(os.environ.get('SPG_LIBRARY', 'product'))}: sumsq + adamw_pack {t*1e6:8.1f} us", flush=True)
