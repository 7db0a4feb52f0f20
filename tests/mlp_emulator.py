"""Numpy emulation of the fused MLP kernel's DATAFLOW (anim-nerf_amd/csrc/mlp.hip): the weight pack
layout, the MFMA fragment/accumulator lane maps and the slot algebra that lets accumulators feed
the next layer without any shuffle.  Arithmetic is float64 — this checks indexing, not rounding.

MFMA 32x32 lane maps (cdna guide section 3):  A: lane l holds row i = l&31, k-group h = l>>5;
B: lane l holds column n = l&31, k-group h; D: lane l, reg r -> row (r&3) + 8(r>>2) + 4h, col l&31.
"""
import numpy as np

N_TILES = 78


def enc_channel(j, h):
    if j < 30:
        return 3 + 6 * (j // 3) + 3 * h + (j % 3)
    if j == 30:
        return 2 if h else 0
    return -1 if h else 1


def hidden_feature(epf, f, h, e):
    return 16 * f + 8 * (e >> 2) + 4 * h + (e & 3) if epf == 8 else 8 * f + 4 * h + e


def stages(P, epf):
    EF, HF, DF = 32 // epf, 128 // epf, 64 // epf
    s = [(P["xyz_encoding_1.0.weight"], P["xyz_encoding_1.0.bias"], 256, 63, 63, 8, EF, 0)]
    for l in range(1, 8):
        W, B = P[f"xyz_encoding_{l+1}.0.weight"], P[f"xyz_encoding_{l+1}.0.bias"]
        s.append((W, B, 256, 319, 63, 8, EF, HF) if l == 4 else (W, B, 256, 256, 0, 8, 0, HF))
    s.append((P["sigma.weight"], P["sigma.bias"], 1, 256, 0, 1, 0, HF))
    s.append((P["xyz_encoding_final.weight"], P["xyz_encoding_final.bias"], 256, 256, 0, 8, 0, HF))
    s.append((P["dir_encoding.0.weight"], P["dir_encoding.0.bias"], 128, 256, 0, 4, 0, HF))
    s.append((P["rgb.0.weight"], P["rgb.0.bias"], 3, 128, 0, 1, 0, DF))
    return s


def pack(P, epf):
    """-> list of tiles; tile = (frags[nf][64 lanes][epf], bias[2][16])  (mlp_pack_kernel)."""
    tiles = []
    for (W, B, out_dim, in_dim, enc_cols, n_tiles, nfe, nfh) in stages(P, epf):
        for t in range(n_tiles):
            frags = np.zeros((nfe + nfh, 64, epf))
            for kf in range(nfe + nfh):
                for lane in range(64):
                    i, h = lane & 31, lane >> 5
                    row = 32 * t + i
                    for e in range(epf):
                        if kf < nfe:
                            ch = enc_channel(epf * kf + e, h)
                            col = ch if 0 <= ch < enc_cols else -1
                        else:
                            col = enc_cols + hidden_feature(epf, kf - nfe, h, e)
                        if row < out_dim and 0 <= col < in_dim:
                            frags[kf, lane, e] = W[row, col]
            bias = np.zeros((2, 16))
            for h in range(2):
                for reg in range(16):
                    row = 32 * t + 8 * (reg >> 2) + 4 * h + (reg & 3)
                    if row < out_dim:
                        bias[h, reg] = B[row]
            tiles.append((frags, bias))
    assert len(tiles) == N_TILES
    return tiles


def mfma_tile(wfrags, xfrags, bias):
    """acc[lane, reg] for one out-tile: sum over frags/elems/k-groups of A[i][k] B[k][n]."""
    nf, _, epf = wfrags.shape
    # A[i, f, h, e], B[n, f, h, e]
    A = wfrags.reshape(nf, 2, 32, epf).transpose(2, 0, 1, 3)
    B = xfrags.reshape(nf, 2, 32, epf).transpose(2, 0, 1, 3)
    D = np.einsum("ifhe,nfhe->in", A, B)                     # D[row i][col n]
    acc = np.zeros((64, 16))
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        for reg in range(16):
            acc[lane, reg] = D[(reg & 3) + 8 * (reg >> 2) + 4 * h, n] + bias[h, reg]
    return acc


def run(P, xyz, epf):
    """xyz[32,3] (one column tile) -> rgb[32,3], sigma[32]."""
    EF, HF, DF = 32 // epf, 128 // epf, 64 // epf
    fpt = 16 // epf
    tiles = pack({k: np.asarray(v, dtype=np.float64) for k, v in P.items()}, epf)
    E = np.zeros((EF, 64, epf))
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        for j in range(32):
            if j < 30:
                a = xyz[n, j % 3] * 2.0 ** (j // 3)
                v = np.cos(a) if h else np.sin(a)
            elif j == 30:
                v = xyz[n, 2] if h else xyz[n, 0]
            else:
                v = 0.0 if h else xyz[n, 1]
            E[j // epf, lane, j % epf] = v
    c = [0]

    def layer(n_tiles, inputs, relu, n_out_frags):
        Y = np.zeros((n_out_frags, 64, epf))
        for t in range(n_tiles):
            frags, bias = tiles[c[0]]
            c[0] += 1
            acc = mfma_tile(frags, inputs, bias)
            if relu:
                acc = np.maximum(acc, 0)
            for f in range(fpt):
                Y[t * fpt + f] = acc[:, f * epf:(f + 1) * epf]
        return Y

    Y = layer(8, E, True, HF)
    for l in range(2, 9):
        Y = layer(8, np.concatenate([E, Y]) if l == 5 else Y, True, HF)
    frags, bias = tiles[c[0]]
    c[0] += 1
    sigma = mfma_tile(frags, Y, bias)[:32, 0]
    X = layer(8, Y, False, HF)
    G = layer(4, X, True, DF)
    frags, bias = tiles[c[0]]
    c[0] += 1
    acc = mfma_tile(frags, G, bias)
    assert c[0] == N_TILES
    return 1.0 / (1.0 + np.exp(-acc[:32, :3])), sigma
