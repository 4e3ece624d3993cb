"""az_net_forward_static (include/azamd.h): the forward that allocates nothing, through the C ABI, against
az_net_forward on the same positions -- bit for bit, on two calls in a row, the second with a smaller input (the static
staging array keeps the first call's rows), for both product-grade K3 kernels; the size limit is an error, not a fault."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _positions(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        e = orc.Env("Connect4")
        for _ in range(int(rng.integers(0, 34))):
            if e.done():
                break
            e.step(int(rng.choice(e.valid_move())))
        if not e.done():
            out.append(e.packed())
    return np.array(out)


@pytest.mark.parametrize("kernel", ["wino", "split"])
def test_net_forward_static_equals_net_forward(kernel):
    from alphazero_amd import _abi
    from alphazero_amd.engine import SelfPlayEngine
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=4, n_playout=4, evaluator="cnn")
    eng.load_net(blob, kernel=kernel)
    big, small = _positions(301, 1), _positions(7, 2)
    for states in (big, small, big[:1], _positions(4096, 3)[:4096]):   # 301, then 7 (smaller), then 1, then the maximum
        want = eng.net_forward(states)
        got = eng.net_forward_static(states)
        for w, g in zip(want, got):
            assert np.array_equal(w.view(np.uint32), g.view(np.uint32))
        np.testing.assert_allclose(got[2].sum(1), 1.0, rtol=1e-5)
    # against the oracle's fp32 forward as well (1e-4), so that "equal" is not "equally wrong"
    onet = orc.CNN(blob)
    op, ov = onet.predict(np.array([orc.Env.from_packed("Connect4", w).current_state()[0] for w in small]))
    _, _, probs, value = eng.net_forward_static(small)
    np.testing.assert_allclose(probs, op, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(value, ov, rtol=1e-4, atol=1e-5)
    # the limit: a clear error, nothing launched
    too_many = np.zeros((4097, 2), dtype=np.uint64)
    with pytest.raises(_abi.AzError, match="exceed AZ_NET_FORWARD_STATIC_MAX"):
        eng.net_forward_static(too_many)
    assert eng.net_forward_static(np.zeros((0, 2), dtype=np.uint64))[0].shape == (0, 7)
    got = eng.net_forward_static(small)          # still usable after the error
    assert np.array_equal(got[2], probs)
    eng.close()
