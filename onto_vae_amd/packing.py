"""Mask packing: the reference's list of dense 0/1 masks -> one sparse operator (CSR + CSC).

Host-side wrapper over the ontovae_plan_* entry points (onto_vae_amd/csrc/ontovae_plan.cpp).
Replaces the dense repeat_interleave expansion of reference onto_vae/modules.py:268-275.
"""
import ctypes as C

import numpy as np

from . import _lib


def _i32p(a):
    return a.ctypes.data_as(_lib.c_i32p)


class MaskPlan:
    """Packed decoder structure.  All arrays are host numpy int32.

    Node space: [0, C) the activation panel in the reference's final concat order
    [level K-1 | ... | level 1 | level 0 (z)], then [C, C+G) the genes.
    """

    def __init__(self, mask_list=None, neuronnum=3, links=None, shapes=None):
        L = _lib.lib()
        nu = int(neuronnum)
        handle = C.c_void_p()
        if mask_list is not None:
            ms = []
            for i, m in enumerate(mask_list):
                m = np.asarray(m.detach().cpu().numpy() if hasattr(m, "detach") else m)
                if m.ndim != 2:
                    raise ValueError(f"mask {i} is not a matrix")
                if not np.all((m == 0) | (m == 1)):
                    raise ValueError(f"mask {i} holds values other than 0/1")
                ms.append(np.ascontiguousarray(m != 0, dtype=np.uint8))
            K = len(ms)
            rows = np.array([m.shape[0] for m in ms], dtype=np.int32)
            cols = np.array([m.shape[1] for m in ms], dtype=np.int32)
            ptrs = (_lib.c_u8p * K)(*[m.ctypes.data_as(_lib.c_u8p) for m in ms])
            _lib.check(L.ontovae_plan_create(K, ptrs, _i32p(rows), _i32p(cols), nu, C.byref(handle)), "pack masks")
        else:
            layer, row, col = (np.ascontiguousarray(a, dtype=np.int32) for a in links)
            rows = np.ascontiguousarray([s[0] for s in shapes], dtype=np.int32)
            cols = np.ascontiguousarray([s[1] for s in shapes], dtype=np.int32)
            K = len(rows)
            _lib.check(L.ontovae_plan_create_from_links(K, _i32p(rows), _i32p(cols), nu, len(layer), _i32p(layer),
                                                        _i32p(row), _i32p(col), C.byref(handle)), "pack links")
        self._h = handle
        self._L = L
        self.dims = _lib.Dims()
        _lib.check(L.ontovae_plan_dims(handle, C.byref(self.dims)), "plan dims")
        d = self.dims
        self.G, self.T, self.K, self.nu, self.L, self.C, self.nnz = d.G, d.T, d.K, d.nu, d.L, d.C, int(d.nnz)
        self.level_begin = np.zeros(self.K + 1, dtype=np.int32)
        self.level_end = np.zeros(self.K + 1, dtype=np.int32)
        _lib.check(L.ontovae_plan_levels(handle, _i32p(self.level_begin), _i32p(self.level_end)), "plan levels")
        n = max(self.nnz, 1)
        self.rowptr = np.zeros(self.C + self.G + 1, dtype=np.int32)
        self.colidx = np.zeros(n, dtype=np.int32)
        self.cscptr = np.zeros(self.C + 1, dtype=np.int32)
        self.csc_row = np.zeros(n, dtype=np.int32)
        self.csc_k = np.zeros(n, dtype=np.int32)
        _lib.check(L.ontovae_plan_export(handle, _i32p(self.rowptr), _i32p(self.colidx), _i32p(self.cscptr),
                                         _i32p(self.csc_row), _i32p(self.csc_k)), "plan export")
        self.csc_split = np.zeros(max(self.C, 1), dtype=np.int32)
        _lib.check(L.ontovae_plan_csc_split(handle, _i32p(self.csc_split)), "plan csc split")
        # packed position k of CSR position j: identity for the term layers, (panel node, gene) order for the gene layer
        self.row_k = np.zeros(n, dtype=np.int32)
        _lib.check(L.ontovae_plan_row_k(handle, _i32p(self.row_k)), "plan row_k")
        self.sell = [self._export_sell(d) for d in (0, 1)]
        # SELL-64 form of the gene layer (ontovae_gsell_t): one gene per lane, lists in quads of 4 entries
        ns, nq = C.c_int32(), C.c_int32()
        _lib.check(L.ontovae_plan_gsell_sizes(handle, C.byref(ns), C.byref(nq)), "gsell sizes")
        self.gsell = dict(n_slices=ns.value, n_quads=nq.value, gene=np.full(max(ns.value, 1) * 64, -1, dtype=np.int32),
                          quad0=np.zeros(ns.value + 1, dtype=np.int32), col=np.zeros(max(nq.value, 1) * 256, dtype=np.uint16),
                          k=np.full(max(nq.value, 1) * 256, -1, dtype=np.int32))
        _lib.check(L.ontovae_plan_gsell_export(handle, _i32p(self.gsell["gene"]), _i32p(self.gsell["quad0"]),
                                               self.gsell["col"].ctypes.data_as(C.POINTER(C.c_uint16)), _i32p(self.gsell["k"])),
                   "gsell export")
        # 32-row tiles of the column-block gene kernels, rows ranked by list length (ontovae_plan_tile_info)
        self.gene_tiles = np.zeros((self.G + 31) // 32 * 32 * 4, dtype=np.int32)
        self.panel_tiles = np.zeros(max((self.C + 31) // 32 * 32 * 4, 4), dtype=np.int32)
        _lib.check(L.ontovae_plan_tile_info(handle, 0, _i32p(self.gene_tiles)), "gene tiles")
        _lib.check(L.ontovae_plan_tile_info(handle, 1, _i32p(self.panel_tiles)), "panel tiles")
        self.term_tiles = np.zeros(max((self.C - self.L + 31) // 32 * 32 * 4, 4), dtype=np.int32)
        _lib.check(L.ontovae_plan_tile_info(handle, 2, _i32p(self.term_tiles)), "term tiles")
        # 32-term tiles of the backward gene kernel: one lane group per term, the nu nodes share the gathers (None: not available)
        tp = np.zeros(4 + (self.C // max(self.nu, 1) + 31) // 32 * 32 * 4, dtype=np.int32)
        _lib.check(L.ontovae_plan_tile_info(handle, 3, _i32p(tp)), "term-panel tiles")
        self.tpanel_tiles = tp[4:].copy() if tp[0] == 1 else None
        self.tpanel_max_entries = int(tp[1]) if tp[0] == 1 else 0
        self.entry_layer = np.zeros(n, dtype=np.int32)
        self.entry_row = np.zeros(n, dtype=np.int32)
        self.entry_col = np.zeros(n, dtype=np.int32)
        _lib.check(L.ontovae_plan_entry_coords(handle, _i32p(self.entry_layer), _i32p(self.entry_row),
                                               _i32p(self.entry_col)), "plan coords")
        # per-layer slices of the packed value array and of the bias array
        self.layer_entries = []
        self.layer_nodes = []
        for i in range(self.K):
            ks = np.nonzero(self.entry_layer[:self.nnz] == i)[0]
            k0, k1 = (int(ks[0]), int(ks[-1]) + 1) if len(ks) else (0, 0)
            assert k1 - k0 == len(ks), "layer entries must be contiguous"
            self.layer_entries.append((k0, k1))
            n0, no = C.c_int32(), C.c_int32()
            _lib.check(L.ontovae_plan_layer_node0(handle, i, C.byref(n0), C.byref(no)), "layer node0")
            self.layer_nodes.append((n0.value, no.value))
        # dense shapes of the reference's Linear layers: (out_i, in_i)  (modules.py:269)
        self.layer_shapes = []
        cin = self.L
        for i in range(self.K):
            cout = self.layer_nodes[i][1]
            self.layer_shapes.append((cout, cin))
            cin += cout if i < self.K - 1 else 0
        self.layout = _lib.Layout()
        _lib.check(L.ontovae_param_layout(C.byref(self.dims), C.byref(self.layout)), "param layout")

    def _export_sell(self, direction):
        """Level-scheduled SELL-64 form of the term layers (see ontovae_sell_t in the header)."""
        L, h = self._L, self._h
        nl, ns, nc, single = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(L.ontovae_plan_sell_sizes(h, direction, C.byref(nl), C.byref(ns), C.byref(nc), C.byref(single)), "sell sizes")
        z = lambda n: np.zeros(max(int(n), 1), dtype=np.int32)
        d = dict(n_levels=nl.value, n_slices=ns.value, n_chunks=nc.value, all_single_chunk=single.value,
                 level_first=z(nl.value), level_count=z(nl.value), node=z(ns.value * 64), info=z(ns.value * 64),
                 chunk0=z(ns.value + 1), idx=z(nc.value * 512), k=z(nc.value * 512))
        _lib.check(L.ontovae_plan_sell_export(h, direction, _i32p(d["level_first"]), _i32p(d["level_count"]), _i32p(d["node"]),
                                              _i32p(d["info"]), _i32p(d["chunk0"]), _i32p(d["idx"]), _i32p(d["k"])), "sell export")
        return d

    def unpack_mask(self, layer):
        """Expanded dense 0/1 mask of layer i, uint8 [out_i, in_i] (== OntoDecoder.masks[i])."""
        cout, cin = self.layer_shapes[layer]
        out = np.zeros((cout, cin), dtype=np.uint8)
        _lib.check(self._L.ontovae_plan_unpack_mask(self._h, layer, out.ctypes.data_as(_lib.c_u8p), out.size), "unpack mask")
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._L.ontovae_plan_destroy(h)
            except Exception:
                pass
