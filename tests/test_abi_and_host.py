"""Host-side checks that need no GPU: the C ABI library loads and exports every symbol the header
declares, the HBM-resident loader reproduces the reference's batch order, the model's state_dict
has the reference layout and round-trips, and compute entry points fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_case, sub


def test_header_symbols_are_exported():
    from onto_vae_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ontovae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ontovae_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in include/ontovae_hip.h but not exported"
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    assert L.ontovae_abi_version() == _lib.ABI_VERSION


def test_loader_matches_reference_order():
    from onto_vae_amd.fast_data_loader import FastTensorDataLoader
    fx = dict(np.load(os.path.join(GOLDEN, "loader.npz")))
    N, B, seed, epochs = (int(fx[k]) for k in ("N", "B", "seed", "epochs"))
    X = torch.arange(N, dtype=torch.float32).reshape(N, 1) * torch.ones(1, 2)
    torch.manual_seed(seed)
    dl = FastTensorDataLoader(X, batch_size=B, shuffle=True, device="cpu")
    assert len(dl) == 4
    for e in range(epochs):
        batches = [b for b in dl]
        assert all(isinstance(b, tuple) and len(b) == 1 for b in batches)
        got = np.concatenate([b[0][:, 0].numpy().astype(np.int64) for b in batches])
        assert np.array_equal(got, fx["order"][e])
        assert [len(b[0]) for b in batches] == [16, 16, 16, 5]
    dl2 = FastTensorDataLoader(X, batch_size=B, shuffle=False, device="cpu")
    got = np.concatenate([b[0][:, 0].numpy().astype(np.int64) for b in dl2])
    assert np.array_equal(got, fx["order_noshuffle"])
    # index view used by the training loop: same rows without copying
    it = iter(dl2)
    idx = it.next_indices()
    assert idx.dtype == torch.int32 and idx.tolist() == list(range(16))


def test_state_dict_layout_and_round_trip(case):
    from onto_vae_amd.synthetic import StubOntobj
    from onto_vae_amd.vae_model import OntoVAE
    _, fx, masks = case
    nu = int(fx["nu"])
    onto = StubOntobj(masks, data={"d": np.zeros((3, masks[-1].shape[0]))})
    model = OntoVAE(onto, "d", neuronnum=nu, drop=0.0, z_drop=0.0)
    want = sub(fx, "sd0/")
    sd = model.state_dict()
    assert list(sd.keys()) == list(want.keys())          # same keys, same order as the reference
    for k, v in want.items():
        assert tuple(sd[k].shape) == tuple(np.shape(v)), k
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in want.items()})
    back = model.state_dict()
    for k, v in want.items():
        assert np.array_equal(back[k].cpu().numpy(), np.asarray(v)), k
    # the reference's attributes
    assert model.latent_dim == masks[0].shape[1] * nu and model.in_features == masks[-1].shape[0]
    assert [tuple(int(v) for v in s) for s in model.decoder.layer_shapes] == \
           [(cin, cout) for cout, cin in model._engine.plan.layer_shapes]
    from ontovae_oracle import expand_masks
    for i, m in enumerate(expand_masks(masks, nu)):
        assert np.array_equal(model.decoder.masks[i].cpu().numpy(), m)
        w = model.decoder.decoder[i][0].weight
        assert tuple(w.shape) == m.shape and float((w.cpu() * (1 - torch.as_tensor(m))).abs().max()) == 0.0
    # off-mask values in a dense import are dropped with a warning
    bad = {k: torch.as_tensor(np.asarray(v)).clone() for k, v in want.items()}
    m0 = expand_masks(masks, nu)[0]
    r, c = np.argwhere(m0 == 0)[0]
    bad["decoder.decoder.0.0.weight"][r, c] = 0.5
    with pytest.warns(UserWarning):
        model.load_state_dict(bad)
    assert float(model.decoder.decoder[0][0].weight[r, c]) == 0.0


def test_errors_follow_the_reference():
    from onto_vae_amd.synthetic import StubOntobj, synthetic_masks
    from onto_vae_amd.vae_model import OntoVAE
    masks = synthetic_masks([3, 4], 12, 1)
    onto = StubOntobj(masks, data={"d": np.zeros((3, 12))})
    with pytest.raises(ValueError, match="Available trimming thresholds are: 1000_30"):
        OntoVAE(onto, "d", top_thresh=5)
    with pytest.raises(ValueError, match="Available datasets are: d"):
        OntoVAE(onto, "nope")
    model = OntoVAE(onto, "d")
    other = StubOntobj(masks, data={"d": np.zeros((3, 12))}, description="other")
    with pytest.raises(ValueError, match="Wrong ontology provided, should be synthetic"):
        model.get_pathway_activities(other, "d")
    if not torch.cuda.is_available():      # no CPU fallback: compute fails loudly
        from onto_vae_amd._lib import OntoVAEHipError
        with pytest.raises(OntoVAEHipError):
            model.forward(torch.zeros(4, 12))
        with pytest.raises(OntoVAEHipError):
            model.get_pathway_activities(onto, "d")


def test_optimizer_state_carries_its_packed_order():
    """FusedAdamW's flat moments follow the plan's packed order, which changed between library ABIs (row_k, ABI 2 -> 7): the
    state names the ABI it was written under and another order is refused instead of being loaded scrambled."""
    from onto_vae_amd._lib import ABI_VERSION
    from onto_vae_amd.synthetic import StubOntobj, synthetic_masks
    from onto_vae_amd.vae_model import FusedAdamW, OntoVAE
    masks = synthetic_masks([3, 4], 12, 1)
    model = OntoVAE(StubOntobj(masks, data={"d": np.zeros((3, 12))}), "d")
    opt = FusedAdamW(model, lr=1e-3)
    opt.step_count = 5
    model._engine.M.fill_(0.25)
    sd = opt.state_dict()
    assert sd["format"] == "onto_vae_amd.flat.v2" and sd["packed_order_abi"] == ABI_VERSION
    opt2 = FusedAdamW(model, lr=1e-4)
    model._engine.M.zero_()
    opt2.load_state_dict(sd)
    assert opt2.step_count == 5 and float(model._engine.M.min()) == 0.25 and opt2.param_groups[0]["lr"] == 1e-3
    for bad in (dict(sd, format="onto_vae_amd.flat.v1"), dict(sd, packed_order_abi=ABI_VERSION - 1),
                {k: v for k, v in sd.items() if k != "packed_order_abi"}):
        with pytest.raises(ValueError, match="packed order"):
            opt2.load_state_dict(bad)
