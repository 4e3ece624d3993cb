#!/usr/bin/env python3
"""Round-2 planning datum for K3: how far are the network's outputs from fp32 when every
convolution product a*w is computed as hi*hi + hi*lo + lo*hi with hi/lo the fp16 halves of the fp32 operands and the
accumulation stays in fp32 -- the arithmetic an fp16-MFMA version of the kernel would do?  CPU only (torch), on the 256
positions of tests/golden/cnn.npz whose fp32 outputs were recorded from the reference.

    python tools/split_fp16_accuracy.py
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def take(blob, off, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(blob[off:off + n].reshape(shape).copy()), off + n


def load(blob):
    """state_dict order of Connect4/Network.py (conv weight, then BN weight / bias / running_mean / running_var)"""
    off, L = 0, {}

    def conv_bn(name, co, ci, k):
        nonlocal off
        w, off = take(blob, off, (co, ci, k, k))
        g, off = take(blob, off, (co,))
        b, off = take(blob, off, (co,))
        m, off = take(blob, off, (co,))
        v, off = take(blob, off, (co,))
        L[name] = (w, g, b, m, v)

    conv_bn("stem", 128, 3, 3)
    for i in range(3):
        conv_bn("b%d_3" % i, 128, 128, 3)
        conv_bn("b%d_1" % i, 128, 128, 1)
    conv_bn("ph", 3, 128, 1)
    L["pw"], off = take(blob, off, (7, 126))
    L["pb"], off = take(blob, off, (7,))
    conv_bn("vh", 1, 128, 1)
    L["vw"], off = take(blob, off, (3, 42))
    L["vb"], off = take(blob, off, (3,))
    assert off == blob.size, (off, blob.size)
    return L


def halves(x, scale=1.0):
    xs = x * scale
    hi = xs.half().float()
    lo = (xs - hi).half().float()
    return hi, lo


def conv(x, w, pad, mode, dtype):
    if mode == "exact":
        return F.conv2d(x.to(dtype), w.to(dtype), padding=pad)
    # weights are pre-scaled by a power of two per layer (largest that keeps them below 2^14) so that their lo halves
    # stay clear of the fp16 subnormals; undone after the sum
    s = float(2.0 ** np.floor(np.log2(16384.0 / float(w.abs().max()))))
    xh, xl = halves(x)
    wh, wl = halves(w, s)
    if mode == "fp16":      # plain fp16 operands, fp32 accumulation
        return F.conv2d(xh, wh, padding=pad) / s
    y = F.conv2d(xh, wh, padding=pad) + F.conv2d(xh, wl, padding=pad) + F.conv2d(xl, wh, padding=pad)
    return y / s


def forward(L, x, mode, dtype=torch.float32):
    def cbn(name, x, pad):
        w, g, b, m, v = L[name]
        y = conv(x, w, pad, mode, dtype).to(dtype)
        return (y - m.to(dtype)[None, :, None, None]) / torch.sqrt(v.to(dtype) + 1e-5)[None, :, None, None] * \
            g.to(dtype)[None, :, None, None] + b.to(dtype)[None, :, None, None]

    x = x.to(dtype)
    h = F.silu(cbn("stem", x, 1))
    for i in range(3):
        t = F.silu(cbn("b%d_3" % i, h, 1))
        h = F.silu(h + cbn("b%d_1" % i, t, 0))
    p = F.silu(cbn("ph", h, 0)).flatten(1) @ L["pw"].to(dtype).T + L["pb"].to(dtype)
    v = F.silu(cbn("vh", h, 0)).flatten(1) @ L["vw"].to(dtype).T + L["vb"].to(dtype)
    return torch.log_softmax(p, 1), torch.log_softmax(v, 1)


def main():
    torch.set_num_threads(8)
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    L = load(blob)
    d = np.load(os.path.join(G, "cnn.npz"))
    cells = torch.from_numpy(d["cells"].reshape(-1, 6, 7).astype(np.float32))
    turn = torch.from_numpy(d["turn"].astype(np.float32))
    x = torch.stack([(cells == 1).float(), (cells == -1).float(), turn[:, None, None].expand(-1, 6, 7)], 1)
    ref64 = forward(L, x, "exact", torch.float64)
    gold = (torch.from_numpy(d["logp"]).double(), torch.from_numpy(d["logv"]).double())

    def err(a, b):
        e = (a.double() - b).abs()
        return float(e.max()), float((e / b.abs().clamp_min(1e-3)).max())

    print("structure check, float64 forward vs recorded reference fp32: logp max abs %.2e, logv max abs %.2e" %
          (err(ref64[0], gold[0])[0], err(ref64[1], gold[1])[0]))
    for mode in ("exact", "split", "fp16"):
        lp, lv = forward(L, x, mode)
        a, r = err(lp, ref64[0])
        a2, r2 = err(lv, ref64[1])
        agree = float((lp.argmax(1) == ref64[0].argmax(1)).float().mean())
        print("%-6s vs float64: logp max abs %.2e (rel %.2e)  logv max abs %.2e (rel %.2e)  top-move agreement %.4f" %
              (mode, a, r, a2, r2, agree))


if __name__ == "__main__":
    main()
