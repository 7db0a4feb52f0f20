"""Forward-mode normals as plain tensor ops (any device, any dtype): the mathematical reference of
`anim_nerf_amd.autograd.NormalFunction`, which runs the same computation in the fused HIP kernels.  TEST INFRASTRUCTURE:
tests/test_host_logic.py holds it to autograd-of-autograd in fp64 on the CPU; tests/test_gpu_training.py holds the kernels
to it and to the oracle."""
import torch


def _encode(xyz: torch.Tensor, n_freqs: int = 10) -> torch.Tensor:
    cols = [xyz]
    for k in range(n_freqs):
        cols += [torch.sin(xyz * float(2 ** k)), torch.cos(xyz * float(2 ** k))]
    return torch.cat(cols, -1)


def _splitk_tn(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dy[rows, o]^T @ x[rows, c] with the long reduction split into S batches: a [256 x rows] x [rows x 256] product gives
    the library 16 output tiles for 256 CUs."""
    rows = dy.shape[0]
    S = 1
    while S < 64 and rows % (2 * S) == 0 and rows // (2 * S) >= 512:
        S *= 2
    if S == 1:
        return dy.t() @ x
    return torch.bmm(dy.view(S, rows // S, -1).transpose(1, 2), x.view(S, rows // S, -1)).sum(0)


class NormalFunctionTorch(torch.autograd.Function):
    """normal[n,3] = d alpha / d xyz with alpha = 1 - exp(-delta relu(sigma(xyz))) (models/nerf.py:177-190), and its
    gradient w.r.t. the trunk and sigma weights — the second-order term of the normals regulariser (train.py:288-309).

    Forward mode instead of autograd-of-autograd: the three tangents d/dx, d/dy, d/dz ride through the trunk as three
    extra rows per point (no bias, gated by the ReLU mask of the point's own activations), so the whole term is ONE
    4n-row pass through 9 library GEMMs; ReLU has zero curvature almost everywhere, so the backward is the plain linear
    backward of that 4n-row pass (bias gradients from the primal rows only).  ~80 launches instead of ~1400."""

    KEYS = [k for i in range(1, 9) for k in (f"xyz_encoding_{i}.0.weight", f"xyz_encoding_{i}.0.bias")] + ["sigma.weight", "sigma.bias"]
    _tables = {}

    @staticmethod
    def _tangent_tables(device, dtype):
        key = (str(device), dtype)
        if key not in NormalFunctionTorch._tables:
            perm, scale = list(range(63)), [0.0] * 63
            for c in range(3):
                scale[c] = 0.0                                        # d x / d x = 1: patched below via the ones column
            for k in range(10):
                f = float(2 ** k)
                for d in range(3):
                    s_ch, c_ch = 3 + 6 * k + d, 6 + 6 * k + d
                    perm[s_ch], scale[s_ch] = c_ch, f
                    perm[c_ch], scale[c_ch] = s_ch, -f
            axis = torch.zeros(3, 63)
            for c in range(63):
                axis[c % 3, c] = 1.0
            add = torch.zeros(63)
            add[:3] = 1.0
            NormalFunctionTorch._tables[key] = (torch.tensor(perm, device=device), torch.tensor(scale, device=device, dtype=dtype),
                                           axis.to(device=device, dtype=dtype), add.to(device=device, dtype=dtype))
        t = NormalFunctionTorch._tables[key]
        return t[0], t[1], t[2]

    @staticmethod
    def forward(ctx, xyz, delta, *params):
        P = dict(zip(NormalFunctionTorch.KEYS, [p.detach() for p in params]))
        n = xyz.shape[0]
        x = xyz.detach()
        e = _encode(x)      # [n,63]
        # tangents of the encoding d e / d x_d -> T0[3, n, 63]: channel c belongs to axis c % 3; d sin(f x) = f cos(f x)
        # and d cos(f x) = -f sin(f x) are the partner channel times +-f
        perm, scale, axis = NormalFunctionTorch._tangent_tables(x.device, x.dtype)
        dE = torch.addcmul(NormalFunctionTorch._tables[(str(x.device), x.dtype)][3], e.index_select(1, perm), scale)
        T0 = dE[None] * axis[:, None, :]
        X0 = torch.cat([e[None], T0], 0)                              # [4, n, 63]: primal row group + 3 tangent groups
        saved_in, masks = [], []
        h = X0
        for l in range(1, 9):
            inp = X0 if l == 1 else torch.cat([X0, h], -1) if l == 5 else h
            pre = inp @ P[f"xyz_encoding_{l}.0.weight"].t()           # [4, n, 256]
            pre[0] += P[f"xyz_encoding_{l}.0.bias"]
            mask = pre[0] > 0
            h = pre * mask
            saved_in.append(inp)
            masks.append(mask)
        sig = h @ P["sigma.weight"].t()                               # [4, n, 1]
        sig[0] += P["sigma.bias"]
        s0 = sig[0, :, 0]
        pos = s0 > 0
        scale = torch.where(pos, delta * torch.exp(-delta * s0), torch.zeros_like(s0))     # d alpha / d sigma
        normal = (scale[None] * sig[1:, :, 0]).t().contiguous()       # [n, 3]
        ctx.save_for_backward(h, sig, scale, *saved_in, *masks, *params)
        ctx.delta = delta
        return normal

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        saved = ctx.saved_tensors
        h8, sig, scale = saved[:3]
        saved_in, masks, params = saved[3:11], saved[11:19], saved[19:]
        P = dict(zip(NormalFunctionTorch.KEYS, params))
        delta = ctx.delta
        grads = {}
        gt = g.t()                                                    # [3, n]
        d_sig = torch.empty_like(sig)                                 # [4, n, 1]
        d_sig[1:, :, 0] = gt * scale[None]
        # d scale / d sigma = -delta * scale where sigma > 0
        d_sig[0, :, 0] = (gt * sig[1:, :, 0]).sum(0) * (-delta) * scale
        grads["sigma.weight"] = torch.einsum("gno,gnc->oc", d_sig, h8)
        grads["sigma.bias"] = d_sig[0].sum(0)
        dh = d_sig * P["sigma.weight"]                                # [4, n, 256]
        for l in range(8, 0, -1):
            dpre = dh * masks[l - 1]
            inp = saved_in[l - 1]
            W = P[f"xyz_encoding_{l}.0.weight"]
            grads[f"xyz_encoding_{l}.0.weight"] = _splitk_tn(dpre.reshape(-1, dpre.shape[-1]), inp.reshape(-1, inp.shape[-1]))
            grads[f"xyz_encoding_{l}.0.bias"] = dpre[0].sum(0)
            if l > 1:
                dh = dpre @ (W[:, 63:] if l == 5 else W)
        out = [grads[k].reshape(p.shape) if ctx.needs_input_grad[2 + i] else None
               for i, (k, p) in enumerate(zip(NormalFunctionTorch.KEYS, params))]
        return (None, None, *out)


