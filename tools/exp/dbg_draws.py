import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anim_nerf_amd import ops
dev = torch.device("cuda:0")
vt = torch.randn(1, 6890, 3, generator=torch.Generator().manual_seed(0)).to(dev)
st = torch.tensor([1234] + [0] * 34, dtype=torch.int64, device=dev)
d = ops.train_draws(st, verts_template=vt, point_scale=0.2 * 0.5, neighbour_scale=0.01)
n0, n1, pair = d["n0"], d["n1"], d["pair"]
sep = vt + n0 * 0.2 * 0.5
fma = torch.addcmul(vt, n0, torch.tensor(0.1, device=dev))
sep64 = (vt.double() + (n0.double() * float(torch.tensor(0.1, dtype=torch.float32))).float().double()).float()
got = pair[:6890][None]
for name, ref in (("separate", sep), ("addcmul", fma), ("separate via fp64", sep64)):
    print(name, int((got != ref).sum()), (got - ref).abs().max().item())
nb = sep + n1 * 0.01
print("neighbours", int((pair[6890:][None] != nb).sum()), (pair[6890:][None] - nb).abs().max().item())
nb2 = got + n1 * 0.01
print("neighbours from kernel pts", int((pair[6890:][None] != nb2).sum()))
