import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import seeded_model
from anim_nerf_amd import synthetic
from anim_nerf_amd.autograd import MLPFunction
tbl = synthetic.make_smpl_table(0)
dev = torch.device("cuda:0")
m = seeded_model(tbl, 9, True, gain=50.0, device=dev)
net = m.nerf
gen = torch.Generator().manual_seed(3)
for n in (777, 832, 64):
    pts = torch.cat([torch.rand(n, 3, generator=gen) * 2 - 1, torch.ones(n, 1)], -1)
    pts[::13, 3] = 0.0
    g = torch.randn(n, 4, generator=gen).to(dev)
    res = []
    for lib in (False, True):
        MLPFunction.LIBRARY_GEMMS = lib
        net.zero_grad()
        p = pts.to(dev).requires_grad_(True)
        out = net.eval_points(p, "f32", sigma_only=False)
        (out.reshape(n, -1) * g).sum().backward()
        res.append({k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    MLPFunction.LIBRARY_GEMMS = False
    print("n", n, {k: round(((res[0][k] - res[1][k]).norm() / (res[1][k].norm() + 1e-20)).item(), 6) for k in res[1]})
