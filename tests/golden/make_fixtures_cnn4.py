#!/usr/bin/env python3
"""Golden vectors for K3 from ALL FOUR checkpoints the reference ships (params/AZ{,2}_Connect4_CNN_{current,best}.pt) and
from range-stress variants of one of them, by RUNNING THE REFERENCE'S OWN CNN (environments/Connect4/Network.py) on CPU.

Container-only (needs /root/reference and oracle/_ref/env_cython.so).  Writes data only:
  tests/golden/cnn4.npz       128 positions; per checkpoint and per stress variant the reference's float32
                              forward/predict outputs AND the outputs of the same module in float64 (net.double()),
                              against which the kernels' and torch-float32's own rounding error is measured
  tests/golden/weights/*.f32  the three other checkpoints re-exported as flat float32 blobs (state_dict order), beside the
                              one round 1 exported; MIT notice in weights/LICENSE.reference
Checkpoints are loaded with torch.load(weights_only=True).

Stress variants (tests/helpers.py: stress_blob applies the SAME transformation to the blob at test time, so no extra blob
is shipped).  Each multiplies BatchNorm affine parameters by 2^k somewhere and divides the weights that consume that
activation by 2^k, which leaves the pre-activation of the consumer unchanged where SiLU is linear and pushes the
activation itself far outside (or deep inside) fp16's range -- the case a fixed-scale fp16 split cannot survive:
  big_t      t = SiLU(BN(conv3x3(x))) of every block x 2^14 (values ~ 2^18), conv1x1 weights x 2^-14
  small_t    the same with 2^-18 (values ~ 2^-16), conv1x1 weights x 2^18
  big_x      the trunk x x 2^15 in every layer (stem and block-output BN x 2^15, conv3x3 and head conv weights x 2^-15)
  small_x    the same with 2^-20
  bn_jitter  running_var x [0.25, 4], running_mean + N(0, 0.5 sigma), gamma x [0.5, 2] per channel (seeded), all layers
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refload  # noqa: E402
from helpers import STRESS_VARIANTS, blob_to_state_dict, stress_blob  # noqa: E402

ref = _refload.load_reference()
CKPTS = ["AZ_Connect4_CNN_current", "AZ_Connect4_CNN_best", "AZ2_Connect4_CNN_current", "AZ2_Connect4_CNN_best"]


def export_blob(sd):
    parts = [v.detach().cpu().numpy().astype(np.float32).ravel() for k, v in sd.items() if v.dtype == torch.float32]
    return np.concatenate(parts)


def cells_of(env):
    return np.asarray(env.board, dtype=np.int8)


def positions(rng, n):
    cells, turns, states = [], [], []
    while len(cells) < n:
        env = ref.C4.Env()
        depth = int(rng.integers(0, 38))
        for _ in range(depth):
            if env.done():
                break
            env.step(int(rng.choice(env.valid_move())))
        if env.done():
            continue
        if len(cells) % 2 == 1:
            env = env.flip()
        cells.append(cells_of(env))
        turns.append(env.turn)
        states.append(env.current_state()[0])
    return np.array(cells), np.array(turns), np.array(states)


def run(sd, states):
    """the reference module in float32 (forward + predict) and in float64 (forward)"""
    net = ref.C4.CNN(0, device="cpu")
    net.load_state_dict(sd)
    net.eval()
    st = torch.from_numpy(states).float()
    with torch.no_grad():
        logp, logv = net.forward(st)
    probs, value = net.predict(st)
    net64 = ref.C4.CNN(0, device="cpu")
    net64.load_state_dict(sd)
    net64.double().eval()
    with torch.no_grad():
        logp64, logv64 = net64.forward(st.double())
    return dict(logp=logp.numpy(), logv=logv.numpy(), probs=np.asarray(probs), value=np.asarray(value),
                logp64=logp64.numpy(), logv64=logv64.numpy())


def main():
    torch.set_num_threads(1)
    rng = np.random.default_rng(20261111)
    cells, turns, states = positions(rng, 128)
    out = dict(cells=cells, turn=turns)
    meta = dict(numpy=np.__version__, torch=torch.__version__, checkpoints={}, variants=list(STRESS_VARIANTS))
    wdir = os.path.join(HERE, "weights")
    base_blob = None
    for name in CKPTS:
        sd = torch.load(os.path.join(_refload.REF, "params", name + ".pt"), map_location="cpu", weights_only=True)
        blob = export_blob(sd).astype("<f4")
        path = os.path.join(wdir, name + ".f32")
        if os.path.exists(path):
            assert np.array_equal(np.fromfile(path, dtype="<f4"), blob), name
        else:
            blob.tofile(path)
        meta["checkpoints"][name] = dict(file="weights/%s.f32" % name, floats=int(blob.size),
                                         sha256=hashlib.sha256(blob.tobytes()).hexdigest())
        # the blob -> state_dict helper the tests use must reproduce the checkpoint exactly
        sd2 = blob_to_state_dict(blob)
        for k, v in sd.items():
            if v.dtype == torch.float32:
                assert torch.equal(v, sd2[k]), (name, k)
        for k, v in run(sd, states).items():
            out["%s/%s" % (name, k)] = v
        if name == CKPTS[0]:
            base_blob = blob
    for var in STRESS_VARIANTS:
        sd = blob_to_state_dict(stress_blob(base_blob, var))
        for k, v in run(sd, states).items():
            assert np.all(np.isfinite(v)), (var, k)
            out["stress_%s/%s" % (var, k)] = v
    np.savez_compressed(os.path.join(HERE, "cnn4.npz"), **out)
    with open(os.path.join(HERE, "cnn4.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for name in CKPTS:
        e = np.abs(out[name + "/logp"] - out[name + "/logp64"])
        print("%-28s torch f32 vs f64: max %.2e mean %.2e" % (name, e.max(), e.mean()))
    for var in STRESS_VARIANTS:
        e = np.abs(out["stress_%s/logp" % var] - out["stress_%s/logp64" % var])
        print("stress %-21s torch f32 vs f64: max %.2e mean %.2e" % (var, e.max(), e.mean()))


if __name__ == "__main__":
    main()
