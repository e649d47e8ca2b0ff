"""onto_vae_amd.mask_builder against masks made by the reference's own
Ontobj.create_masks / utils.create_binary_matrix code (tools/gen_golden_masks.py): bit-exact."""
import json
import os

import numpy as np
import pytest

from onto_vae_amd.mask_builder import PackedMasks, create_masks, links_from_graph
from onto_vae_amd.packing import MaskPlan

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, f"masks_{name}.npz"), allow_pickle=False)
    graph = json.loads(bytes(z["graph_json"]).decode())
    annot = {"ID": [str(v) for v in z["annot_ids"]], "depth": [int(v) for v in z["annot_depth"]]}
    genes = [str(v) for v in z["genes"]]
    masks = [z[f"mask{i}"].astype(np.int64) for i in range(int(z["nmask"]))]
    return graph, annot, genes, masks


@pytest.mark.parametrize("name", ["graph_small", "graph_deep"])
def test_links_reproduce_reference_masks(name):
    graph, annot, genes, want = load(name)
    links, shapes = links_from_graph(graph, annot, genes)
    pm = PackedMasks(links, shapes)
    assert len(pm) == len(want)
    for i, m in enumerate(want):
        assert pm.shapes[i] == m.shape
        assert np.array_equal(pm[i], m), i
    # and through the packer: unpack(pack(links)) is the reference's expanded mask (modules.py:268-275)
    nu = 2
    plan = MaskPlan(neuronnum=nu, links=pm.links, shapes=pm.shapes)
    for i, m in enumerate(want):
        exp = np.repeat(m, nu, axis=1) if i == len(want) - 1 else np.repeat(np.repeat(m, nu, axis=0), nu, axis=1)
        assert np.array_equal(plan.unpack_mask(i), exp.astype(plan.unpack_mask(i).dtype)), i
    same = MaskPlan(want, nu)
    assert np.array_equal(same.rowptr, plan.rowptr) and np.array_equal(same.colidx, plan.colidx)


@pytest.mark.parametrize("name", ["graph_small", "graph_deep"])
def test_encoder_links_reproduce_reference_masks(name):
    """The encoder masks of OntoEncVAE (ontobj.py:394-404) on random DAGs with links that skip levels, a trimmed-away parent and a
    duplicated parent entry: bit-exact against what the reference's code made of the same graph."""
    from onto_vae_amd.mask_builder import encoder_links_from_graph
    graph, annot, genes, dec = load(name)
    z = np.load(os.path.join(GOLDEN, f"masks_{name}.npz"), allow_pickle=False)
    want = [z[f"emask{i}"].astype(np.int64) for i in range(len(dec))]
    links, shapes = encoder_links_from_graph(graph, annot, genes)
    pm = PackedMasks(links, shapes)
    assert len(pm) == len(want) and len(links[0]) == sum(int(m.sum()) for m in want) == sum(int(m.sum()) for m in dec)
    for i, m in enumerate(want):
        assert pm.shapes[i] == m.shape and np.array_equal(pm[i], m), i


def test_create_masks_on_ontobj_like_object_and_model():
    graph, annot, genes, want = load("graph_small")

    class Obj:
        description = "syn"
    o = Obj()
    o.graph, o.annot, o.genes, o.masks = {"1000_30": graph}, {"1000_30": annot}, {"1000_30": genes}, {}
    o.data = {"1000_30": {"d": np.zeros((3, len(genes)))}}
    with pytest.raises(ValueError, match="trim_dag function needs to be run first"):
        create_masks(o, 10, 3)
    create_masks(o)
    assert all(np.array_equal(a, b) for a, b in zip(o.masks["1000_30"]["decoder"], want))
    from onto_vae_amd.vae_model import OntoVAE
    model = OntoVAE(o, "d", neuronnum=3)
    ref = OntoVAE(type("O", (), {"description": "syn", "genes": o.genes, "data": o.data,
                                 "masks": {"1000_30": {"decoder": want}}})(), "d", neuronnum=3)
    assert model.latent_dim == ref.latent_dim and list(model.layer_dims_dec) == list(ref.layer_dims_dec)
    assert np.array_equal(model._engine.plan.colidx, ref._engine.plan.colidx)


def test_bad_graphs():
    with pytest.raises(ValueError):
        links_from_graph({}, {"ID": [], "depth": []}, ["g"])
    with pytest.raises(ValueError, match="two depths"):
        links_from_graph({}, {"ID": ["a", "b"], "depth": [0, 1]}, ["a"])
