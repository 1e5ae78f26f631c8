"""Weight loaders (d2slam_amd/weights.py): the .pth route of the reference's notebook (superpoint.ipynb cell 3: `torch.load('superpoint_v1.pth')` into
SuperPointNet.load_state_dict) and its round trip with the .npz / ONNX-initializer routes.  VERDICT r04 #2: `load_superpoint_pth` had no test."""
import json
import os

import numpy as np
import pytest
import torch

from d2slam_amd import weights as W

NB = "/root/reference/d2frontend/superpoint.ipynb"


def _state_dict(w):
    sd = {}
    for n in W.SP_LAYERS:
        sd[n + ".weight"] = torch.from_numpy(w[n][0].copy()); sd[n + ".bias"] = torch.from_numpy(w[n][1].copy())
    return sd


def _same(a, b):
    return all(np.array_equal(a[n][0], b[n][0]) and np.array_equal(a[n][1], b[n][1]) and a[n][0].dtype == np.float32 for n in W.SP_LAYERS)


def test_load_superpoint_pth_round_trip(tmp_path):
    """a state_dict with the MagicLeap key names (conv1a.weight ... convDb.bias) saved with torch.save comes back layer for layer, bit for bit, in the
    shapes of superpoint.ipynb:306-321 -- also from a float64 checkpoint (converted) and with extra keys present (ignored)."""
    w = W.synthetic_superpoint_weights(seed=77)
    p = str(tmp_path / "sp.pth")
    torch.save(_state_dict(w), p)
    got = W.load_superpoint_pth(p)
    assert list(got) == W.SP_LAYERS and _same(got, w)
    for n in W.SP_LAYERS:
        co, ci, k = W.SP_SHAPES[n]
        assert got[n][0].shape == (co, ci, k, k) and got[n][1].shape == (co,)
    sd = {k: v.double() for k, v in _state_dict(w).items()}
    sd["num_batches_tracked"] = torch.zeros(1)
    torch.save(sd, p)
    assert _same(W.load_superpoint_pth(p), w)
    # the three on-disk routes agree
    W.save_superpoint_npz(str(tmp_path / "sp.npz"), w)
    assert _same(W.load_superpoint_npz(str(tmp_path / "sp.npz")), w)


def test_load_superpoint_pth_missing_layer_fails_loudly(tmp_path):
    sd = _state_dict(W.synthetic_superpoint_weights())
    del sd["convDb.bias"]
    p = str(tmp_path / "bad.pth")
    torch.save(sd, p)
    with pytest.raises(KeyError):
        W.load_superpoint_pth(p)


@pytest.mark.skipif(not os.path.exists(NB), reason="needs the reference tree (the notebook's module is executed, not copied)")
def test_load_superpoint_pth_from_the_notebook_module(tmp_path):
    """torch.save of the state_dict of the reference's OWN module (SuperPointNet, superpoint.ipynb cell 1, executed from the notebook as
    tests/golden/make_golden_ref.py does): its key set is exactly the loader's, and what the loader returns loads back into the module
    (load_state_dict is strict) and reproduces its parameters."""
    cells = ["".join(c["source"]) for c in json.load(open(NB))["cells"]]
    ns = {"torch": torch}
    exec(compile(cells[1], NB + ":cell1", "exec"), ns)
    torch.manual_seed(5)
    net = ns["SuperPointNet"]()
    p = str(tmp_path / "superpoint_v1.pth")
    torch.save(net.state_dict(), p)
    assert sorted(net.state_dict()) == sorted(n + s for n in W.SP_LAYERS for s in (".weight", ".bias"))
    got = W.load_superpoint_pth(p)
    for n in W.SP_LAYERS:
        assert np.array_equal(got[n][0], getattr(net, n).weight.detach().numpy()) and np.array_equal(got[n][1], getattr(net, n).bias.detach().numpy())
        assert tuple(got[n][0].shape) == (W.SP_SHAPES[n][0], W.SP_SHAPES[n][1], W.SP_SHAPES[n][2], W.SP_SHAPES[n][2])
    net2 = ns["SuperPointNet"]()
    net2.load_state_dict(_state_dict(got))
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)
