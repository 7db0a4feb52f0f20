import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
import test_gpu_training as T
from helpers import InjectedDraws
dev = torch.device("cuda:0")
smpl = syn.make_smpl_table(0)
m, table, batch = T._config3_scene(dev, smpl)
vr = ana.VolumeRenderer(n_coarse=64, n_fine=32)
hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
loss, details, grads, drawn = T._step_gradients(m, table, vr, hp, batch, InjectedDraws(seed=5))
print(loss.item(), {k: round(float(v), 5) for k, v in details.items()})
print({k: float(g.abs().max()) for k, g in grads.items() if float(g.abs().max()) == 0 or "smpl" in k})
