"""debug: which autograd Function delivers a gradient for a sink-owned tensor?"""
import os, sys, traceback
import torch, torch.distributed as dist
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn, autograd as ag, training as trn
from helpers import golden, seeded_model
import ddp_worker as W

LOG = []
for name in ("MLPFunction", "QuadSigmaFunction", "NormalFunction", "FeatureFunction", "CompositeFunction", "WarpFunction", "TrainLossFunction"):
    cls = getattr(ag, name)
    orig = cls.backward
    def make(orig, name):
        def wrapped(ctx, *g):
            LOG.append(("enter", name, getattr(ctx, "sink", "n/a") is not None, getattr(ctx, "sigma_only", None)))
            out = orig(ctx, *g)
            LOG.append(("exit", name, [i for i, o in enumerate(out) if o is not None]))
            return out
        return staticmethod(wrapped)
    cls.backward = make(orig, name)

orig_arrived = trn.GradientReducer._arrived
def arrived(self, p):
    if self.active and self._pending and id(p) in self._sink_params:
        LOG.append(("hook", tuple(p.shape), [i for i, q in enumerate(self.buckets[self.slot[p][0]]) if q is p], self.slot[p][0], self._next))
    try:
        return orig_arrived(self, p)
    except RuntimeError:
        if dist.get_rank() == 0:
            for l in LOG[-40:]:
                print(l, flush=True)
        raise
trn.GradientReducer._arrived = arrived

dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
hp = ana.TrainHParams(n_samples=16, n_importance=8, chunk=64, lambda_normals=0.05, lr=1e-3)
vr = ana.VolumeRenderer(n_coarse=16, n_fine=8)
m = seeded_model(tbl, g["seed"], True, g["gain"], g["shift"], device=dev)
tr = ana.Trainer(m, vr, hp)
# re-register: Trainer registered the original bound method
pose, rays, tgt, alp, fg, bg = W.batch(rank, dev)
torch.manual_seed(500 + rank)
tr.step(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
print("ok", rank)
dist.destroy_process_group()
