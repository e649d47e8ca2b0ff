"""Decoder masks straight from the trimmed ontology graph, without the dense intermediates.

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
  * mask i has the IDs of level i+1 as rows and [level i | level i-1 | ... | level 0] as columns
    (ontobj.py:380-385).
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


def create_masks(ontobj, top_thresh=1000, bottom_thresh=30, module='decoder'):
    """Drop-in for Ontobj.create_masks (ontobj.py:327-374), decoder module only: fills
    ontobj.masks[key]['decoder'] with a PackedMasks built from ontobj.graph / .annot / .genes."""
    key = str(top_thresh) + '_' + str(bottom_thresh)
    if key not in ontobj.graph.keys():
        raise ValueError('Trimmed graph with specified thresholds missing, trim_dag function needs to be run first!')
    if key not in ontobj.annot.keys():
        raise ValueError('Trimmed annotation with specified thresholds missing, trim_dag function needs to be run first!')
    if key not in ontobj.genes.keys():
        raise ValueError('Trimmed gene list with specified thresholds missing, trim_dag function needs to be run first!')
    if module != 'decoder':
        raise NotImplementedError("only the decoder masks are on the OntoVAE path (OntoEncVAE is out of scope)")
    links, shapes = links_from_graph(ontobj.graph[key], ontobj.annot[key], ontobj.genes[key])
    if key not in ontobj.masks.keys():
        ontobj.masks[key] = {}
    ontobj.masks[key][module] = PackedMasks(links, shapes)
