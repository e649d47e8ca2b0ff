"""Mask packing is bit-exact: unpack(pack(M)) equals the reference's expanded masks
(onto_vae/modules.py:271-275), and the CSR/CSC forms describe the same operator."""
import numpy as np
import pytest

from conftest import load_case
from onto_vae_amd.packing import MaskPlan
from onto_vae_amd.synthetic import synthetic_masks
from ontovae_oracle import expand_masks


def check_plan(masks, nu):
    plan = MaskPlan(masks, nu)
    exp = expand_masks(masks, nu)
    assert plan.K == len(masks) and plan.G == masks[-1].shape[0] and plan.T == masks[-1].shape[1]
    assert plan.L == masks[0].shape[1] * nu and plan.C == plan.T * nu
    assert plan.nnz == int(sum(m.sum() for m in exp))
    for i, m in enumerate(exp):
        assert plan.layer_shapes[i] == m.shape
        assert np.array_equal(plan.unpack_mask(i), m.astype(np.uint8)), f"layer {i}"
    # entry coordinates address exactly the ones of the expanded masks: row-major within a term layer, column-major
    # (by panel node, then gene) within the gene layer
    for i, m in enumerate(exp):
        k0, k1 = plan.layer_entries[i]
        r, c = plan.entry_row[k0:k1], plan.entry_col[k0:k1]
        if i < plan.K - 1:
            rr, cc = np.nonzero(m)
        else:
            cc, rr = np.nonzero(m.T)
        assert np.array_equal(r, rr) and np.array_equal(c, cc)
    # CSR node mapping (positions j): row node = level block + row, col node = C - in_i + col; row_k[j] = packed position
    C = plan.C
    assert sorted(plan.row_k[:plan.nnz].tolist()) == list(range(plan.nnz))
    nnz_term = int(plan.rowptr[C])
    assert np.array_equal(plan.row_k[:nnz_term], np.arange(nnz_term))
    rows = np.repeat(np.arange(C + plan.G), np.diff(plan.rowptr))
    for i in range(plan.K):
        k0, k1 = plan.layer_entries[i]
        n0, nout = plan.layer_nodes[i]
        cin = plan.layer_shapes[i][1]
        js = np.nonzero((rows >= n0) & (rows < n0 + nout))[0]
        assert sorted(plan.row_k[js].tolist()) == list(range(k0, k1))
        assert np.array_equal(rows[js], n0 + plan.entry_row[plan.row_k[js]])
        assert np.array_equal(plan.colidx[js], C - cin + plan.entry_col[plan.row_k[js]])
    # CSC is the transpose: same (row, col, k) triples; the gene part of the lists walks k in steps of 1
    csc_cols = np.repeat(np.arange(C), np.diff(plan.cscptr))
    a = set(zip(rows.tolist(), plan.colidx[:plan.nnz].tolist(), plan.row_k[:plan.nnz].tolist()))
    b = set(zip(plan.csc_row[:plan.nnz].tolist(), csc_cols.tolist(), plan.csc_k[:plan.nnz].tolist()))
    assert a == b
    gene_k = plan.csc_k[:plan.nnz][plan.csc_row[:plan.nnz] >= C]
    assert np.array_equal(gene_k, np.arange(nnz_term, plan.nnz))
    for p in range(C):  # rows ascend inside a column
        seg = plan.csc_row[plan.cscptr[p]:plan.cscptr[p + 1]]
        assert np.all(np.diff(seg) > 0)
    # SELL form of the gene layer: every gene owns one lane, its quads hold exactly the (panel row, packed position) pairs of
    # its CSR list, padding has k = -1; genes of a slice come from one window of 512 and lists shorten along a window
    gs = plan.gsell
    if C <= 65535:
        lanes = gs["gene"][:gs["n_slices"] * 64]
        assert sorted(lanes[lanes >= 0].tolist()) == list(range(plan.G))
        col = gs["col"][:gs["n_quads"] * 256].reshape(-1, 64, 4)
        kk = gs["k"][:gs["n_quads"] * 256].reshape(-1, 64, 4)
        assert int((kk >= 0).sum()) == plan.nnz - nnz_term
        prev_len = {}
        for sl in range(gs["n_slices"]):
            q0, q1 = gs["quad0"][sl], gs["quad0"][sl + 1]
            gl = lanes[sl * 64:(sl + 1) * 64]
            assert len({g // 512 for g in gl[gl >= 0].tolist()}) == 1
            lens = []
            for lane in range(64):
                g = gl[lane]
                c_, k_ = col[q0:q1, lane].reshape(-1), kk[q0:q1, lane].reshape(-1)
                if g < 0:
                    assert (k_ < 0).all()
                    continue
                j0, j1 = plan.rowptr[C + g], plan.rowptr[C + g + 1]
                n = j1 - j0
                assert np.array_equal(c_[:n], plan.colidx[j0:j1]) and np.array_equal(k_[:n], plan.row_k[j0:j1])
                assert (k_[n:] < 0).all() and (c_[n:] == 0).all()
                lens.append(n)
            assert lens == sorted(lens, reverse=True) and q1 - q0 == (lens[0] + 3) // 4
            w = gl[0] // 512
            assert prev_len.get(w, 1 << 30) >= lens[0]
            prev_len[w] = lens[-1]
    else:
        assert gs["n_slices"] == 0
    # 32-row tiles of the column-block gene kernels: every row of a tile once, longest list first, with its list
    gt = plan.gene_tiles.reshape(-1, 32, 4)
    pt = plan.panel_tiles.reshape(-1, 32, 4)
    assert gt.shape[0] == (plan.G + 31) // 32 and pt.shape[0] == (C + 31) // 32
    nterm = C - plan.L
    tt = plan.term_tiles[:(nterm + 31) // 32 * 32 * 4].reshape(-1, 32, 4)
    for rec in tt.reshape(-1, 4):
        if rec[2] < nterm:
            assert rec[0] == plan.rowptr[rec[2]] and rec[1] == plan.rowptr[rec[2] + 1] - rec[0]
    for tiles, n, unused in ((gt, plan.G, plan.G), (pt, C, C), (tt, nterm, nterm)):
        for t, rec in enumerate(tiles):
            rows_ = rec[:, 2]
            want_rows = [r for r in range(t * 32, t * 32 + 32) if r < n]
            assert sorted(rows_[rows_ != unused].tolist()) == want_rows
            assert np.all(np.diff(rec[:, 1]) <= 0) and np.all(rec[rows_ == unused, 1] == 0)
    for rec in gt.reshape(-1, 4):
        if rec[2] < plan.G:
            assert rec[0] == plan.rowptr[C + rec[2]] and rec[1] == plan.rowptr[C + rec[2] + 1] - rec[0]
    for rec in pt.reshape(-1, 4):
        if rec[2] < C:
            p_ = rec[2]
            assert rec[0] == plan.csc_split[p_] and rec[1] == plan.cscptr[p_ + 1] - plan.csc_split[p_]
            if rec[1] > 0:
                assert rec[3] == plan.csc_k[rec[0]]
    # 32-term tiles of the backward gene kernel: every term once, longest list first; the term's nu nodes share the gene list
    # of its first node and their links sit `len` apart in the packed order
    nu = plan.nu
    if nu <= 4:
        assert plan.tpanel_tiles is not None
        T_ = C // nu
        tp = plan.tpanel_tiles.reshape(-1, 32, 4)
        assert tp.shape[0] == (T_ + 31) // 32
        for t, rec in enumerate(tp):
            nodes = rec[:, 2]
            assert sorted(nodes[nodes != C].tolist()) == [nu * r for r in range(t * 32, t * 32 + 32) if r < T_]
            assert np.all(np.diff(rec[:, 1]) <= 0) and np.all(rec[nodes == C, 1] == 0)
        for k0, ln, node, kb in tp.reshape(-1, 4):
            if node == C:
                continue
            assert k0 == plan.csc_split[node] and ln == plan.cscptr[node + 1] - k0
            genes = plan.csc_row[k0:k0 + ln]
            for e in range(nu):
                ke = plan.csc_split[node + e]
                assert plan.cscptr[node + e + 1] - ke == ln and np.array_equal(plan.csc_row[ke:ke + ln], genes)
                assert np.array_equal(plan.csc_k[ke:ke + ln], kb + e * ln + np.arange(ln))
    # levels tile the panel in reverse order and then the genes
    assert plan.level_begin[0] == C - plan.L and plan.level_end[0] == C
    for d in range(1, plan.K):
        assert plan.level_end[d] == plan.level_begin[d - 1]
    assert plan.level_begin[plan.K - 1] == 0 and plan.level_begin[plan.K] == C and plan.level_end[plan.K] == C + plan.G
    return plan


def test_pack_golden_cases(case):
    _, fx, masks = case
    check_plan(masks, int(fx["nu"]))


def test_pack_gobp_syn_counts():
    masks = synthetic_masks()
    plan = MaskPlan(masks, 3)
    assert (plan.G, plan.T, plan.K, plan.L, plan.C) == (20000, 3000, 12, 360, 9000)
    assert plan.nnz == 520551 and int(plan.dims.nnz_links) == 4595 + 159732
    assert [s for s in plan.layer_shapes][:3] == [(180, 360), (450, 540), (990, 990)]
    assert plan.layer_shapes[-1] == (20000, 9000)
    # spot-check one inner layer and the gene layer bit for bit
    exp = expand_masks([masks[3], masks[3]], 3)[0]
    assert np.array_equal(plan.unpack_mask(3), exp.astype(np.uint8))
    gene = np.repeat(masks[-1], 3, axis=1).astype(np.uint8)
    assert np.array_equal(plan.unpack_mask(11), gene)


def test_pack_from_links_equals_dense():
    masks = synthetic_masks(counts=[4, 6, 5, 3], n_genes=40, seed=7)
    layer, row, col = [], [], []
    for i, m in enumerate(masks):
        r, c = np.nonzero(m)
        layer += [i] * len(r); row += r.tolist(); col += c.tolist()
    perm = np.random.RandomState(0).permutation(len(layer))  # order must not matter, duplicates collapse
    layer, row, col = (np.array(a)[perm] for a in (layer, row, col))
    layer, row, col = (np.concatenate([a, a[:5]]) for a in (layer, row, col))
    a = MaskPlan(masks, 2)
    b = MaskPlan(neuronnum=2, links=(layer, row, col), shapes=[m.shape for m in masks])
    for f in ("rowptr", "colidx", "cscptr", "csc_row", "csc_k", "row_k", "entry_layer", "entry_row", "entry_col"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_pack_rejects_bad_input():
    good = synthetic_masks(counts=[3, 4], n_genes=10, seed=1)
    with pytest.raises(ValueError):
        MaskPlan([good[0], good[1][:, :-1]], 3)          # column count breaks the layout rule
    bad = [m.copy() for m in good]
    bad[0][0, 0] = 2
    with pytest.raises(ValueError):
        MaskPlan(bad, 3)
    with pytest.raises(ValueError):
        MaskPlan(neuronnum=1, links=([0], [99], [0]), shapes=[m.shape for m in good])


def check_sell(plan):
    """The level-scheduled SELL-64 form holds exactly the term-layer entries of the CSR (forward)
    and of the term part of the CSC (backward), every node once per direction, levels in order."""
    C = plan.C
    for direction, sd in enumerate(plan.sell):
        ns, nl = sd["n_slices"], sd["n_levels"]
        levels = list(range(1, plan.K)) if direction == 0 else list(range(plan.K - 1, -1, -1))
        assert nl == len(levels)
        seen = {}
        for li, d in enumerate(levels):
            f, c = int(sd["level_first"][li]), int(sd["level_count"][li])
            want_nodes = set(range(plan.level_begin[d], plan.level_end[d]))
            got_nodes = set()
            for sl in range(f, f + c):
                c0, c1 = int(sd["chunk0"][sl]), int(sd["chunk0"][sl + 1])
                info = sd["info"][sl * 64:(sl + 1) * 64]
                node = sd["node"][sl * 64:(sl + 1) * 64]
                lg = (info >> 16) & 7
                assert len(set(lg.tolist())) == 1
                g = 1 << int(lg[0])
                width = (info >> 20) & 15
                assert len(set(width.tolist())) == 1
                W = int(width[0])
                assert W <= 8 and (c1 - c0 == 1 or W == 8)
                # rows a slice may read: the levels before it in processing order
                ok_row = (lambda r: r >= plan.level_end[d]) if direction == 0 else (lambda r: r < plan.level_begin[d])
                for lane in range(64):
                    n = int(node[lane])
                    cnt = int(info[lane] & 0xFFFF)
                    writer = (int(info[lane]) >> 24) & 1
                    if n < 0:
                        assert cnt == 0 and not writer
                        continue
                    assert writer == (lane % g == g - 1)
                    assert cnt <= (c1 - c0) * 8
                    assert c1 - c0 > 1 or cnt <= W
                    got_nodes.add(n)
                    for e in range(cnt, W):                       # padding inside the width: weight 0, a readable row
                        pos = c0 * 512 + e * 64 + lane
                        assert int(sd["k"][pos]) == -1 and ok_row(int(sd["idx"][pos]))
                    for e in range(cnt):
                        pos = (c0 + e // 8) * 512 + (e % 8) * 64 + lane
                        seen.setdefault(n, []).append((int(sd["idx"][pos]), int(sd["k"][pos])))
                # lanes of one node are adjacent and aligned
                for lane in range(0, 64, g):
                    assert len(set(node[lane:lane + g].tolist())) == 1
            if direction == 1:      # backward: nodes without term children are left out (their D row does not change)
                want_nodes = {n for n in want_nodes if plan.csc_split[n] > plan.cscptr[n]}
            assert got_nodes == want_nodes, (direction, d)
        if sd["all_single_chunk"]:
            assert np.array_equal(sd["chunk0"][:ns + 1], np.arange(ns + 1))
        for d in levels:
            for n in range(plan.level_begin[d], plan.level_end[d]):
                if direction == 0:
                    ks = range(plan.rowptr[n], plan.rowptr[n + 1])
                    want = [(int(plan.colidx[k]), k) for k in ks]
                else:
                    qs = range(plan.cscptr[n], plan.csc_split[n])
                    want = [(int(plan.csc_row[q]), int(plan.csc_k[q])) for q in qs]
                    assert all(r < C for r, _ in want)
                assert seen.get(n, []) == want, (direction, n)


def test_sell_golden_cases(case):
    _, fx, masks = case
    check_sell(MaskPlan(masks, int(fx["nu"])))


def test_sell_long_rows():
    """Terms with hundreds of parents/children: many lanes per node and multi-chunk lanes."""
    rng = np.random.RandomState(0)
    n0, n1, n2, G = 700, 3, 2, 12
    m0 = (rng.rand(n1, n0) < 0.9).astype(np.int64)          # level-1 terms with ~630 parents each
    m1 = (rng.rand(n2, n1 + n0) < 0.5).astype(np.int64)
    m2 = (rng.rand(G, n2 + n1 + n0) < 0.05).astype(np.int64)
    plan = MaskPlan([m0, m1, m2], 3)
    assert not plan.sell[0]["all_single_chunk"]
    check_sell(plan)
