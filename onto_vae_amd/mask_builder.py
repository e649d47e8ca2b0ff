"""Decoder (and encoder) masks straight from the trimmed ontology graph, without the dense intermediates.

The reference builds its masks in Ontobj.create_masks (onto_vae/ontobj.py:327-392) with
utils.create_binary_matrix (onto_vae/utils.py:67-90): for every pair of depth levels it enumerates ALL
(child, parent) combinations in Python, pivots them into a dense 0/1 DataFrame and concatenates those into dense
int64 matrices (4.2 s for the Pathway Ontology, minutes to hours and a 480 MB gene mask at GO scale).  The same
wiring follows directly from the parent lists: one pass over the edges gives (layer, row, column) links, which
ontovae_plan_create_from_links packs (CSR / CSC / SELL) without any dense matrix.

Orderings reproduced from the reference, because they define what a mask row / column means:
  * levels are the sorted distinct depths of annot['depth'], genes form one more level (max depth + 1)
    (ontobj.py:366-369);
  * inside a level the IDs are in sorted order (DataFrame.pivot sorts its index and columns, utils.py:86);
  * decoder mask i has the IDs of level i+1 as rows and [level i | level i-1 | ... | level 0] as columns
    (ontobj.py:382-392);
  * encoder mask i (OntoEncVAE) runs the other way: with K the gene level, its rows are the IDs of level K-1-i (the PARENTS)
    and its columns the children [level K-i | level K-i+1 | ... | level K = genes] (ontobj.py:394-404: the transposed
    pair matrices in the order itertools.combinations lists them).
"""
import numpy as np


class PackedMasks:
    """What create_masks stores in ontobj.masks[key]['decoder']: the links and the shapes.  Indexing materialises the
    dense int64 mask the reference would hold (for code that wants to look at one); OntoVAE packs the links directly."""

    def __init__(self, links, shapes):
        self.links = tuple(np.ascontiguousarray(a, dtype=np.int32) for a in links)
        self.shapes = [tuple(int(v) for v in s) for s in shapes]

    def __len__(self):
        return len(self.shapes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        layer, row, col = self.links
        m = np.zeros(self.shapes[i], dtype=np.int64)
        sel = layer == i
        m[row[sel], col[sel]] = 1
        return m

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def links_from_graph(onto_dict, annot, genes):
    """onto_dict: {ID: list of parent IDs} over terms and genes (Ontobj.graph[key]); annot: DataFrame (or mapping of
    columns) with 'ID' and 'depth' of the terms (Ontobj.annot[key]); genes: list of gene IDs (Ontobj.genes[key]).
    Returns ((layer, row, col) int32 arrays, shapes)."""
    ids = list(annot['ID'])
    depths = [int(d) for d in annot['depth']]
    if len(ids) == 0:
        raise ValueError("annot lists no terms")
    levels = sorted(set(depths))
    gene_depth = max(levels) + 1
    members = {d: [] for d in levels}
    for t, d in zip(ids, depths):
        members[d].append(t)
    members[gene_depth] = list(genes)
    levels.append(gene_depth)
    pos = {}                                     # ID -> (level index, position inside the sorted level)
    sizes = []
    for li, d in enumerate(levels):
        srt = sorted(members[d])
        if len(set(srt)) != len(srt):
            raise ValueError(f"duplicate IDs at depth {d}")
        sizes.append(len(srt))
        for p, t in enumerate(srt):
            if t in pos:
                raise ValueError(f"ID {t!r} appears at two depths")
            pos[t] = (li, p)
    K = len(levels) - 1
    if K < 1:
        raise ValueError("the graph has a single level")
    # column offset of parent level j inside mask i: [level i | level i-1 | ... | level 0]
    layer, row, col = [], [], []
    shapes = []
    for i in range(K):
        shapes.append((sizes[i + 1], int(sum(sizes[:i + 1]))))
    for child, (li, r) in pos.items():
        if li == 0:
            continue
        i = li - 1
        seen = set()
        for parent in onto_dict.get(child, ()):
            pp = pos.get(parent)
            if pp is None or pp[0] >= li or parent in seen:       # trimmed away, not above the child, or listed twice
                continue
            seen.add(parent)
            off = int(sum(sizes[pp[0] + 1:li]))
            layer.append(i)
            row.append(r)
            col.append(off + pp[1])
    return (np.asarray(layer, dtype=np.int32), np.asarray(row, dtype=np.int32), np.asarray(col, dtype=np.int32)), shapes


def encoder_links_from_graph(onto_dict, annot, genes):
    """The encoder masks of Ontobj.create_masks(module='encoder') (ontobj.py:394-404) as links: arguments and return value
    as links_from_graph."""
    (dl, dr, dc), dshapes = links_from_graph(onto_dict, annot, genes)
    K = len(dshapes)
    sizes = [dshapes[0][1]] + [s[0] for s in dshapes]            # level sizes, level 0 (roots) .. level K (genes)
    # a decoder link (layer i, row r, col c): child = (level i+1, r); c walks [level i | level i-1 | ... | level 0]
    bounds = [np.cumsum([0] + [sizes[j] for j in range(i, -1, -1)]) for i in range(K)]
    layer = np.empty(len(dl), np.int32)
    row = np.empty(len(dl), np.int32)
    col = np.empty(len(dl), np.int32)
    for i in range(K):
        sel = np.nonzero(dl == i)[0]
        blk = np.searchsorted(bounds[i], dc[sel], side="right") - 1      # block j of mask i holds level i - j
        plevel = i - blk
        ppos = dc[sel] - bounds[i][blk]
        layer[sel] = K - 1 - plevel
        row[sel] = ppos
        # columns of the parent's mask: [level p+1 | ... | level K]; the child sits at level i+1
        starts = np.array([int(sum(sizes[p + 1:i + 1])) for p in range(i + 1)], np.int64)
        col[sel] = starts[plevel] + dr[sel]
    shapes = [(sizes[K - 1 - j], int(sum(sizes[K - j:]))) for j in range(K)]
    order = np.lexsort((col, row, layer))
    return (layer[order], row[order], col[order]), shapes


def create_masks(ontobj, top_thresh=1000, bottom_thresh=30, module='decoder'):
    """Drop-in for Ontobj.create_masks (ontobj.py:327-379): fills ontobj.masks[key][module] with a PackedMasks built from
    ontobj.graph / .annot / .genes (module 'decoder' for OntoVAE, anything else the encoder masks of OntoEncVAE, as the
    reference's else branch, ontobj.py:373-376)."""
    key = str(top_thresh) + '_' + str(bottom_thresh)
    if key not in ontobj.graph.keys():
        raise ValueError('Trimmed graph with specified thresholds missing, trim_dag function needs to be run first!')
    if key not in ontobj.annot.keys():
        raise ValueError('Trimmed annotation with specified thresholds missing, trim_dag function needs to be run first!')
    if key not in ontobj.genes.keys():
        raise ValueError('Trimmed gene list with specified thresholds missing, trim_dag function needs to be run first!')
    build = links_from_graph if module == 'decoder' else encoder_links_from_graph
    links, shapes = build(ontobj.graph[key], ontobj.annot[key], ontobj.genes[key])
    if key not in ontobj.masks.keys():
        ontobj.masks[key] = {}
    ontobj.masks[key][module] = PackedMasks(links, shapes)
