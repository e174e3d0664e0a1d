"""Third, independent restatement of modulated deformable convolution (DCNv2), written from the PAPER's equations
(Zhu et al., "Deformable ConvNets v2", eq. 1 with the bilinear kernel of Dai et al., "Deformable Convolutional
Networks", eqs. 3-4) in float64 NumPy -- NOT from oracle/crfp_oracle.py::dcnv2 nor from oracle/dcnv2_ref.c:

    y(p) = b + sum_k  w_k . x(p + p_k + dp_k) * dm_k                          (v2, eq. 1)
    x(p) = sum_q G(q, p) x(q),  G(q, p) = g(q_y, p_y) g(q_x, p_x),  g(a, b) = max(0, 1 - |a - b|)   (v1, eqs. 3-4)

with x(q) = 0 for q outside the image (zero padding), p_k the regular 3x3 grid {-1,0,1}^2 (pad 1, dilation 1), and the
layer's tensor conventions: offset channels (dy, dx) interleaved per tap per deformable group, mask channel per tap per
group.  The sum over q is taken over ALL integer rows / columns the hat function can reach (no floor / corner
bookkeeping), which makes the boundary behaviour at -1, 0, H-1, H fall out of the equations themselves.
Test infrastructure only.
"""
import numpy as np


def _hat_matrix(p: np.ndarray, size: int) -> np.ndarray:
    """G factor for one axis: [..., size] with g(q, p) = max(0, 1 - |q - p|) for q = 0..size-1."""
    q = np.arange(size, dtype=np.float64)
    return np.maximum(0.0, 1.0 - np.abs(q - p[..., None]))


def dcnv2_paper(x, offset, mask, weight, bias, dg):
    x = np.asarray(x, np.float64)
    offset = np.asarray(offset, np.float64)
    mask = np.asarray(mask, np.float64)
    weight = np.asarray(weight, np.float64)
    bias = np.asarray(bias, np.float64)
    B, C, H, W = x.shape
    O = weight.shape[0]
    cpg = C // dg
    out = np.zeros((B, O, H, W)) + bias.reshape(1, O, 1, 1)
    ys = np.arange(H, dtype=np.float64).reshape(H, 1)
    xs = np.arange(W, dtype=np.float64).reshape(1, W)
    for b in range(B):
        for g in range(dg):
            xg = x[b, g * cpg:(g + 1) * cpg]                                  # [cpg, H, W]
            for k in range(9):
                pky, pkx = k // 3 - 1, k % 3 - 1
                py = ys + pky + offset[b, 2 * (g * 9 + k)]                    # [H, W]
                px = xs + pkx + offset[b, 2 * (g * 9 + k) + 1]
                Gy = _hat_matrix(py, H)                                       # [H, W, H]
                Gx = _hat_matrix(px, W)                                       # [H, W, W]
                # x(p) = sum_qy sum_qx Gy Gx x(q)
                tmp = np.einsum("yxq,cqr->cyxr", Gy, xg)                      # rows interpolated, [cpg, H, W, W]
                val = np.einsum("cyxr,yxr->cyx", tmp, Gx) * mask[b, g * 9 + k]
                wk = weight[:, g * cpg:(g + 1) * cpg, k // 3, k % 3]          # [O, cpg]
                out[b] += np.einsum("oc,cyx->oyx", wk, val)
    return out


def boundary_offsets(rs, B, dg, H, W):
    """Offsets whose sampling positions straddle -1, 0, H-1 and H (and the same on x): for pixel (y, x), tap k the
    position is y + k//3 - 1 + dy, so dy = target - (y + k//3 - 1)."""
    off = rs.uniform(-2.5, 2.5, (B, 2 * dg * 9, H, W))
    ys = np.arange(H).reshape(H, 1)
    xs = np.arange(W).reshape(1, W)
    targets_y = [-1.0, -1.0 + 1e-3, -0.5, 0.0, 0.25, H - 1.0, H - 1.0 + 0.5, H - 1e-3, float(H), H + 0.75, -1.25]
    targets_x = [-1.0, -1.0 + 1e-3, -0.5, 0.0, 0.25, W - 1.0, W - 1.0 + 0.5, W - 1e-3, float(W), W + 0.75, -1.25]
    i = 0
    for b in range(B):
        for g in range(dg):
            for k in range(9):
                ty = targets_y[i % len(targets_y)]
                tx = targets_x[(i // 2) % len(targets_x)]
                i += 1
                # apply on a sub-lattice of pixels so that interior random offsets stay covered as well
                sel = ((ys + xs + k) % 3 == 0)
                ch = 2 * (g * 9 + k)
                off[b, ch] = np.where(sel, ty - (ys + k // 3 - 1), off[b, ch])
                off[b, ch + 1] = np.where(sel, tx - (xs + k % 3 - 1), off[b, ch + 1])
    return off.astype(np.float32)
