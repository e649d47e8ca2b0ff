// Host-side mask packer: the reference's list of dense 0/1 ontology masks -> one sparse operator
// over the node space [panel | genes] in CSR (forward, weight gradient) and CSC (backward) form.
//
// Replaces reference onto_vae/modules.py:268-275 (dense repeat_interleave expansion) and makes the
// gradient mask of onto_vae/vae_model.py:168-169 implicit.  Input contract: reference
// onto_vae/ontobj.py:386-392 (mask i: rows = level i+1, cols = [level i | ... | level 0]).
#include "ontovae_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <tuple>
#include <vector>

static thread_local char g_err[512] = "";

void ontovae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ontovae_last_error(void) { return g_err; }
extern "C" int ontovae_abi_version(void) { return ONTOVAE_ABI_VERSION; }

struct ontovae_plan {
    ontovae_dims_t dims;
    std::vector<int32_t> n_level;      // terms per level, K entries
    std::vector<int32_t> cum;          // C_d = nu * sum_{j<=d} n_j, K entries
    std::vector<int32_t> level_begin;  // K+1 entries (level K = genes)
    std::vector<int32_t> level_end;
    std::vector<int32_t> rowptr, colidx, cscptr, csc_row, csc_k;
    std::vector<int32_t> row_k;        // packed position k of CSR position j (identity for the term layers)
    std::vector<int32_t> e_layer, e_row, e_col;   // dense coordinates per packed entry
    std::vector<int64_t> layer_k0;     // first packed entry of each layer (K+1)
    // level-scheduled SELL-64 form of the term layers, [0] forward (CSR rows), [1] backward (term part
    // of the CSC columns)
    struct Sell {
        std::vector<int32_t> level_first, level_count;   // slices of each level, in processing order
        std::vector<int32_t> node, info;                 // per lane of each slice
        std::vector<int32_t> chunk0;                     // first 64x8 chunk of each slice (+ sentinel)
        std::vector<int32_t> idx, k;                     // per chunk: [8][64] gather node / packed-weight index
        int32_t all_single = 1;
    } sell[2];
    // SELL-64 form of the gene layer (ontovae_gsell_t): one gene per lane, genes sorted by list length inside windows of 512
    struct GSell {
        std::vector<int32_t> gene, quad0, k;
        std::vector<uint16_t> col;
    } gsell;
    // tiles of 32 rows as the column-block gene kernels take them (ontovae_plan_tile_info), 4 ints per slot
    std::vector<int32_t> gene_tiles, panel_tiles, term_tiles, tpanel_tiles;
};

// The column-block gene kernels give every lane group of a 32-row tile one row, longest list first, so that the 8 groups
// of a wave walk lists of similar length.  The order is a property of the masks: it is laid down here once, one 16-B record
// per slot {first list position, length, row, packed position of the first link}, instead of being found by every workgroup
// (row pointers -> LDS -> rank by counting -> LDS, two barriers) before its first useful load.
//   gene tiles : rows = genes g (record.z = g, G for an unused slot), lists = CSR rows C + g, record.w = 0
//   panel tiles: rows = panel nodes p (record.z = p, C if unused), lists = gene part of CSC column p
//   term tiles : rows = the term layers' output nodes n in [0, C - L) (record.z = n, C - L if unused), lists = CSR rows n
//   term-panel tiles: rows = TERMS t of the panel (record.z = first node nu * t, C if unused), list = gene part of CSC column
//                nu * t.  The nu neurons of a term have the same genes below them (the gene mask is expanded along its columns,
//                modules.py:275), in the same order, and their links sit nu runs of `len` apart in the packed order
//                (record.w, + len, + 2 len, ...): the backward gene kernel gathers a gene's D row ONCE for the nu nodes.
//                Left empty when a structure does not have that property (then the per-node tiles are used).
static void build_tiles(ontovae_plan* p, int32_t nu) {
    const int32_t C = p->dims.C, G = (int32_t)(p->rowptr.size() - 1) - C;
    auto make = [&](int32_t n, int32_t unused, auto k0_of, auto len_of, auto kb_of, std::vector<int32_t>& out) {
        const int32_t ntiles = (n + 31) / 32;
        out.assign((size_t)ntiles * 32 * 4, 0);
        for (int32_t t = 0; t < ntiles; ++t) {
            std::vector<std::pair<int32_t, int32_t>> v;     // (len, row); rows beyond n: length 0
            for (int32_t j = 0; j < 32; ++j) {
                const int32_t r = t * 32 + j;
                v.push_back({r < n ? len_of(r) : 0, r});
            }
            std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            for (int32_t s = 0; s < 32; ++s) {
                const int32_t r = v[s].second;
                int32_t* rec = out.data() + ((size_t)t * 32 + s) * 4;
                rec[0] = r < n ? k0_of(r) : 0;
                rec[1] = v[s].first;
                rec[2] = r < n ? r : unused;
                rec[3] = (r < n && v[s].first > 0) ? kb_of(r) : 0;
            }
        }
    };
    make(G, G, [&](int32_t g) { return p->rowptr[C + g]; }, [&](int32_t g) { return p->rowptr[C + g + 1] - p->rowptr[C + g]; },
         [&](int32_t) { return 0; }, p->gene_tiles);
    auto split = [&](int32_t c) {
        const int32_t* b = p->csc_row.data() + p->cscptr[c];
        const int32_t* e = p->csc_row.data() + p->cscptr[c + 1];
        return (int32_t)(std::lower_bound(b, e, C) - p->csc_row.data());
    };
    make(C, C, split, [&](int32_t c) { return p->cscptr[c + 1] - split(c); }, [&](int32_t c) { return p->csc_k[split(c)]; },
         p->panel_tiles);
    const int32_t nterm = p->level_begin.empty() ? 0 : p->level_begin[0];      // nodes [0, C - L): levels 1 .. K-1
    make(nterm, nterm, [&](int32_t n) { return p->rowptr[n]; }, [&](int32_t n) { return p->rowptr[n + 1] - p->rowptr[n]; },
         [&](int32_t) { return 0; }, p->term_tiles);
    p->tpanel_tiles.clear();
    if (nu >= 1 && nu <= 4 && C % nu == 0) {
        const int32_t T = C / nu;
        bool ok = true;
        for (int32_t t = 0; t < T && ok; ++t) {
            const int32_t c0 = t * nu, k0 = split(c0), len = p->cscptr[c0 + 1] - k0;
            for (int32_t e = 1; e < nu && ok; ++e) {
                const int32_t ke = split(c0 + e);
                ok = p->cscptr[c0 + e + 1] - ke == len && std::equal(p->csc_row.begin() + k0, p->csc_row.begin() + k0 + len, p->csc_row.begin() + ke) &&
                     (len == 0 || p->csc_k[ke] == p->csc_k[k0] + e * len);
            }
            for (int32_t j = 1; j < len && ok; ++j) ok = p->csc_k[k0 + j] == p->csc_k[k0] + j;
        }
        if (ok) {
            std::vector<int32_t> tmp;
            make(T, T, [&](int32_t t) { return split(t * nu); }, [&](int32_t t) { return p->cscptr[t * nu + 1] - split(t * nu); },
                 [&](int32_t t) { return p->csc_k[split(t * nu)]; }, tmp);
            for (size_t i = 0; i + 3 < tmp.size(); i += 4) tmp[i + 2] = tmp[i + 2] >= T ? C : tmp[i + 2] * nu;      // first node of the term
            p->tpanel_tiles.swap(tmp);
        }
    }
}

// Gene layer, one gene per lane.  List lengths run from 3 to 60 links at GO-BP scale: slices of 64 consecutive genes would be
// 1.85x padding, slices of genes sorted inside windows of 512 are 1.15x (entries in quads of 4), and a wave's genes stay
// inside 2 KB of an x row.  Empty (n_slices = 0) when a panel index does not fit 16 bits.
static void build_gsell(ontovae_plan* p) {
    const int32_t C = p->dims.C, G = (int32_t)(p->rowptr.size() - 1) - C;
    ontovae_plan::GSell& S = p->gsell;
    S.quad0.assign(1, 0);
    if (C > 65535 || G < 1) return;
    for (int32_t w0 = 0; w0 < G; w0 += 512) {
        const int32_t cnt = std::min<int32_t>(512, G - w0);
        std::vector<std::pair<int32_t, int32_t>> v;        // (len, gene)
        for (int32_t g = w0; g < w0 + cnt; ++g) v.push_back({p->rowptr[C + g + 1] - p->rowptr[C + g], g});
        std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        for (int32_t i = 0; i < cnt; i += 64) {
            const int32_t n = std::min<int32_t>(64, cnt - i);
            const int32_t nq = (v[i].first + 3) / 4;
            const size_t lane0 = S.gene.size(), e0 = S.col.size();
            S.gene.resize(lane0 + 64, -1);
            S.col.resize(e0 + (size_t)nq * 256, 0);        // padding: weight 0 (k = -1) times panel row 0
            S.k.resize(e0 + (size_t)nq * 256, -1);
            for (int32_t lane = 0; lane < n; ++lane) {
                const int32_t len = v[i + lane].first, g = v[i + lane].second, base = p->rowptr[C + g];
                S.gene[lane0 + lane] = g;
                for (int32_t e = 0; e < len; ++e) {
                    const size_t pos = e0 + (size_t)(e / 4) * 256 + (size_t)lane * 4 + (e % 4);
                    S.col[pos] = (uint16_t)p->colidx[base + e];
                    S.k[pos] = p->row_k[base + e];
                }
            }
            S.quad0.push_back(S.quad0.back() + nq);
        }
    }
}

static void build_sell(ontovae_plan* p, int dir) {
    const int K = p->dims.K;
    const int32_t C = p->dims.C;
    ontovae_plan::Sell& S = p->sell[dir];
    std::vector<int> levels;
    if (dir == 0) for (int d = 1; d < K; ++d) levels.push_back(d);
    else for (int d = K - 1; d >= 0; --d) levels.push_back(d);
    auto lo = [&](int32_t n) { return dir == 0 ? p->rowptr[n] : p->cscptr[n]; };
    auto hi = [&](int32_t n) {
        if (dir == 0) return p->rowptr[n + 1];
        const int32_t* b = p->csc_row.data() + p->cscptr[n];
        const int32_t* e = p->csc_row.data() + p->cscptr[n + 1];
        return (int32_t)(std::lower_bound(b, e, C) - p->csc_row.data());
    };
    S.chunk0.push_back(0);
    int32_t nchunks = 0;
    for (int d : levels) {
        std::vector<std::pair<int32_t, int32_t>> nodes;   // (len, node)
        for (int32_t n = p->level_begin[d]; n < p->level_end[d]; ++n) {
            // backward: a node without term children keeps the gene layer's contribution as it is (the chain kernel loads
            // every panel row first and writes every row back), so it needs no lane
            if (dir == 1 && hi(n) == lo(n)) continue;
            nodes.push_back({hi(n) - lo(n), n});
        }
        std::stable_sort(nodes.begin(), nodes.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        S.level_first.push_back((int32_t)(S.node.size() / 64));
        size_t i = 0;
        while (i < nodes.size()) {
            // lanes per node and chunks per lane of this slice, from its longest (first) node
            const int32_t len0 = nodes[i].first;
            int g = 1;
            while (g < 64 && (int64_t)g * 8 < len0) g *= 2;
            const int32_t R = std::max<int32_t>(1, (len0 + g * 8 - 1) / (g * 8));
            if (R > 1) S.all_single = 0;
            int lg = 0;
            while ((1 << lg) < g) ++lg;
            // entries per lane: an even split over the g lanes when one chunk is enough (W <= 8 is the slice's
            // width: the kernel gathers W entries per lane, not 8), else R chunks of 8 per lane
            const int32_t W = R == 1 ? (len0 + g - 1) / g : 8;
            const int32_t per_lane = R == 1 ? W : R * 8;
            const size_t lane0 = S.node.size();
            S.node.resize(lane0 + 64, -1);
            S.info.resize(lane0 + 64, (lg << 16) | (W << 20));
            const size_t e0 = S.idx.size();
            S.idx.resize(e0 + (size_t)R * 512, -1);
            S.k.resize(e0 + (size_t)R * 512, -1);
            int32_t slice_pad = -1;                                          // a row this slice really reads
            for (int slot = 0; slot < 64 / g && i < nodes.size(); ++slot, ++i) {
                const int32_t len = nodes[i].first, n = nodes[i].second;
                const int32_t base = lo(n);
                for (int gl = 0; gl < g; ++gl) {
                    const int lane = slot * g + gl;
                    const int32_t first = gl * per_lane;                    // this lane's entries: [first, first + per_lane)
                    const int32_t cnt = std::max<int32_t>(0, std::min<int32_t>(len - first, per_lane));
                    S.node[lane0 + lane] = n;
                    S.info[lane0 + lane] = cnt | (lg << 16) | (W << 20) | ((gl == g - 1) ? (1 << 24) : 0);
                    for (int32_t e = 0; e < cnt; ++e) {
                        const int32_t q = base + first + e;
                        const size_t pos = e0 + (size_t)(e / 8) * 512 + (size_t)(e % 8) * 64 + lane;
                        S.idx[pos] = dir == 0 ? p->colidx[q] : p->csc_row[q];
                        S.k[pos] = dir == 0 ? q : p->csc_k[q];
                        if (slice_pad < 0) slice_pad = S.idx[pos];
                    }
                }
            }
            // padding slots carry weight 0 (k = -1) and the index of a row that is valid input of this slice:
            // the lane's own first entry, else the slice's first, so that 0 * value never meets unwritten memory
            for (int lane = 0; lane < 64; ++lane) {
                const int32_t own = S.idx[e0 + lane];
                const int32_t pad = own >= 0 ? own : std::max<int32_t>(slice_pad, 0);
                for (int32_t e = 0; e < R * 8; ++e) {
                    const size_t pos = e0 + (size_t)(e / 8) * 512 + (size_t)(e % 8) * 64 + lane;
                    if (S.k[pos] < 0) S.idx[pos] = pad;
                }
            }
            nchunks += R;
            S.chunk0.push_back(nchunks);
        }
        S.level_count.push_back((int32_t)(S.node.size() / 64) - S.level_first.back());
    }
}

struct Link { int32_t layer, row, col; };

static int build(int32_t K, const int32_t* rows, const int32_t* cols, int32_t nu, std::vector<Link>& links,
                 ontovae_plan_t** out) {
    if (K < 1 || nu < 1 || !rows || !cols || !out) {
        ontovae_set_error("plan: bad arguments (K=%d nu=%d)", K, nu);
        return ONTOVAE_EINVAL;
    }
    ontovae_plan* p = new (std::nothrow) ontovae_plan();
    if (!p) return ONTOVAE_ENOMEM;
    // level sizes from the shapes; cols[i] must equal the running sum (ontobj.py:386-392)
    p->n_level.resize(K);
    p->n_level[0] = cols[0];
    int64_t run = cols[0];
    for (int i = 0; i < K; ++i) {
        if (rows[i] <= 0 || cols[i] <= 0 || cols[i] != run) {
            ontovae_set_error("plan: mask %d has shape [%d, %d], expected %lld columns ([level %d | ... | level 0])",
                              i, rows[i], cols[i], (long long)run, i);
            delete p;
            return ONTOVAE_EINVAL;
        }
        if (i < K - 1) {
            p->n_level[i + 1] = rows[i];
            run += rows[i];
        }
    }
    const int64_t T = run;
    const int64_t G = rows[K - 1];
    const int64_t C = T * nu;
    if (C + G > 0x7fffff00LL) {
        ontovae_set_error("plan: node space too large");
        delete p;
        return ONTOVAE_EINVAL;
    }
    p->cum.resize(K);
    int64_t acc = 0;
    for (int d = 0; d < K; ++d) {
        acc += (int64_t)p->n_level[d] * nu;
        p->cum[d] = (int32_t)acc;
    }
    p->level_begin.resize(K + 1);
    p->level_end.resize(K + 1);
    for (int d = 0; d < K; ++d) {
        p->level_begin[d] = (int32_t)(C - p->cum[d]);
        p->level_end[d] = (int32_t)(d == 0 ? C : C - p->cum[d - 1]);
    }
    p->level_begin[K] = (int32_t)C;
    p->level_end[K] = (int32_t)(C + G);

    // validate + sort + dedupe links
    for (const Link& l : links) {
        if (l.layer < 0 || l.layer >= K || l.row < 0 || l.row >= rows[l.layer] || l.col < 0 || l.col >= cols[l.layer]) {
            ontovae_set_error("plan: link (%d, %d, %d) out of range", l.layer, l.row, l.col);
            delete p;
            return ONTOVAE_EINVAL;
        }
    }
    std::sort(links.begin(), links.end(), [](const Link& a, const Link& b) {
        return std::tie(a.layer, a.row, a.col) < std::tie(b.layer, b.row, b.col);
    });
    links.erase(std::unique(links.begin(), links.end(),
                            [](const Link& a, const Link& b) { return a.layer == b.layer && a.row == b.row && a.col == b.col; }),
                links.end());

    // expanded nnz per node
    const int64_t NN = C + G;
    p->rowptr.assign(NN + 1, 0);
    int64_t nnz = 0;
    for (const Link& l : links) {
        const bool gene = l.layer == K - 1;
        nnz += gene ? nu : (int64_t)nu * nu;
    }
    if (nnz > 0x7fffff00LL) {
        ontovae_set_error("plan: too many non-zeros");
        delete p;
        return ONTOVAE_EINVAL;
    }
    p->colidx.resize(nnz);
    p->e_layer.resize(nnz);
    p->e_row.resize(nnz);
    p->e_col.resize(nnz);
    p->layer_k0.assign(K + 1, 0);

    // Nodes are ordered [level K-1 | ... | level 1 | level 0 | genes]; layer i<K-1 writes level i+1.
    // Emit rows in node order: layers K-2, K-3, ..., 0, then the gene layer K-1.
    std::vector<int64_t> lstart(K + 1, 0);   // first link of each layer in the sorted list
    {
        size_t j = 0;
        for (int i = 0; i <= K; ++i) {
            while (j < links.size() && links[j].layer < i) ++j;
            lstart[i] = (int64_t)j;
        }
    }
    int64_t k = 0;
    std::vector<int> order;
    for (int i = K - 2; i >= 0; --i) order.push_back(i);
    order.push_back(K - 1);
    for (int i : order) {
        const bool gene = i == K - 1;
        const int32_t in0 = (int32_t)(C - p->cum[i]);                      // node of dense column 0
        const int32_t out0 = gene ? (int32_t)C : (int32_t)(C - p->cum[i + 1]);
        const int rep = gene ? 1 : nu;
        p->layer_k0[i] = k;
        int64_t j = lstart[i];
        const int64_t jend = lstart[i + 1];
        for (int32_t r = 0; r < rows[i]; ++r) {
            const int64_t j0 = j;
            while (j < jend && links[j].row == r) ++j;
            for (int a = 0; a < rep; ++a) {
                const int32_t node = out0 + r * rep + a;
                p->rowptr[node] = (int32_t)k;
                for (int64_t q = j0; q < j; ++q) {
                    for (int e = 0; e < nu; ++e) {
                        p->colidx[k] = in0 + links[q].col * nu + e;
                        p->e_layer[k] = i;
                        p->e_row[k] = r * rep + a;
                        p->e_col[k] = links[q].col * nu + e;
                        ++k;
                    }
                }
            }
        }
    }
    // rowptr currently holds row STARTS for nodes with rows; turn it into a proper monotone pointer
    // array.  Node order equals emission order except that level 0 (no rows) sits between level 1
    // and the genes, so a single backward fill is enough.
    {
        std::vector<int32_t> start(p->rowptr);
        std::vector<char> has(NN, 0);
        for (int d = 1; d <= K; ++d)
            for (int32_t n = p->level_begin[d]; n < p->level_end[d]; ++n) has[n] = 1;
        int32_t next = (int32_t)nnz;
        p->rowptr[NN] = next;
        for (int64_t n = NN - 1; n >= 0; --n) {
            if (has[n]) next = start[n];
            p->rowptr[n] = next;
        }
    }
    p->layer_k0[K] = 0;  // unused sentinel (layers are not contiguous in increasing order)

    // CSC by counting sort over column nodes; rows ascend automatically because k ascends by row
    p->cscptr.assign(C + 1, 0);
    for (int64_t q = 0; q < nnz; ++q) p->cscptr[p->colidx[q] + 1]++;
    for (int64_t c = 0; c < C; ++c) p->cscptr[c + 1] += p->cscptr[c];
    p->csc_row.resize(nnz);
    p->csc_k.resize(nnz);
    {
        std::vector<int32_t> fill(p->cscptr.begin(), p->cscptr.end() - 1);
        for (int64_t n = 0; n < NN; ++n)
            for (int32_t q = p->rowptr[n]; q < p->rowptr[n + 1]; ++q) {
                const int32_t c = p->colidx[q];
                const int32_t pos = fill[c]++;
                p->csc_row[pos] = (int32_t)n;
                p->csc_k[pos] = q;
            }
    }
    // Packed position k of every link (= its place in dec_vals and in every buffer laid out like it).  Term layers: the
    // CSR position.  Gene layer: ordered by (panel node, gene), the order in which the gene part of the CSC lists walks the
    // links: the backward gene kernel, which has the D rows of a node's genes in registers and forms the weight gradient
    // of those links there, then reads the weights and writes the gradient partials as contiguous runs per node (by CSR
    // position every one of them was a scattered 4-B access into a 1.9 MB array at GO-BP scale).  The forward gene kernel
    // reads its weights through row_k instead.
    {
        const int32_t nnz_term = p->rowptr[C];
        p->row_k.resize(nnz);
        for (int32_t j = 0; j < nnz_term; ++j) p->row_k[j] = j;
        int32_t next = nnz_term;
        for (int64_t c = 0; c < C; ++c)
            for (int32_t q = p->cscptr[c]; q < p->cscptr[c + 1]; ++q)
                if (p->csc_row[q] >= C) {
                    p->row_k[p->csc_k[q]] = next;
                    p->csc_k[q] = next++;
                }
        std::vector<int32_t> el(p->e_layer), er(p->e_row), ec(p->e_col);
        for (int64_t j = nnz_term; j < nnz; ++j) {
            const int32_t k2 = p->row_k[j];
            p->e_layer[k2] = el[j]; p->e_row[k2] = er[j]; p->e_col[k2] = ec[j];
        }
    }
    p->dims.C = (int32_t)C;
    p->dims.K = K;
    build_sell(p, 0);
    build_sell(p, 1);
    build_gsell(p);
    build_tiles(p, nu);
    p->dims.G = (int32_t)G;
    p->dims.T = (int32_t)T;
    p->dims.K = K;
    p->dims.nu = nu;
    p->dims.L = p->cum[0];
    p->dims.C = (int32_t)C;
    p->dims.nnz = nnz;
    p->dims.nnz_links = (int64_t)links.size();
    p->dims.nnz_term = (int64_t)p->rowptr[C];          // rows [0, C) are the term layers
    *out = p;
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_create(int32_t K, const uint8_t* const* masks, const int32_t* rows, const int32_t* cols,
                                   int32_t nu, ontovae_plan_t** out) {
    if (K < 1 || !masks || !rows || !cols) {
        ontovae_set_error("plan_create: bad arguments");
        return ONTOVAE_EINVAL;
    }
    std::vector<Link> links;
    for (int i = 0; i < K; ++i) {
        if (!masks[i] || rows[i] <= 0 || cols[i] <= 0) {
            ontovae_set_error("plan_create: mask %d is empty", i);
            return ONTOVAE_EINVAL;
        }
        const uint8_t* m = masks[i];
        for (int32_t r = 0; r < rows[i]; ++r)
            for (int32_t c = 0; c < cols[i]; ++c) {
                const uint8_t v = m[(int64_t)r * cols[i] + c];
                if (v > 1) {
                    ontovae_set_error("plan_create: mask %d holds a value other than 0/1 at [%d, %d]", i, r, c);
                    return ONTOVAE_EINVAL;
                }
                if (v) links.push_back({i, r, c});
            }
    }
    return build(K, rows, cols, nu, links, out);
}

extern "C" int ontovae_plan_create_from_links(int32_t K, const int32_t* rows, const int32_t* cols, int32_t nu,
                                              int64_t n_links, const int32_t* layer, const int32_t* row,
                                              const int32_t* col, ontovae_plan_t** out) {
    if (n_links < 0 || (n_links > 0 && (!layer || !row || !col))) {
        ontovae_set_error("plan_create_from_links: bad arguments");
        return ONTOVAE_EINVAL;
    }
    std::vector<Link> links((size_t)n_links);
    for (int64_t i = 0; i < n_links; ++i) links[i] = {layer[i], row[i], col[i]};
    return build(K, rows, cols, nu, links, out);
}

extern "C" void ontovae_plan_destroy(ontovae_plan_t* plan) { delete plan; }

extern "C" int ontovae_plan_dims(const ontovae_plan_t* plan, ontovae_dims_t* out) {
    if (!plan || !out) return ONTOVAE_EINVAL;
    *out = plan->dims;
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_levels(const ontovae_plan_t* plan, int32_t* node_begin, int32_t* node_end) {
    if (!plan || !node_begin || !node_end) return ONTOVAE_EINVAL;
    std::memcpy(node_begin, plan->level_begin.data(), sizeof(int32_t) * plan->level_begin.size());
    std::memcpy(node_end, plan->level_end.data(), sizeof(int32_t) * plan->level_end.size());
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_export(const ontovae_plan_t* plan, int32_t* rowptr, int32_t* colidx, int32_t* cscptr,
                                   int32_t* csc_row, int32_t* csc_k) {
    if (!plan) return ONTOVAE_EINVAL;
    auto cp = [](int32_t* dst, const std::vector<int32_t>& v) {
        if (dst && !v.empty()) std::memcpy(dst, v.data(), sizeof(int32_t) * v.size());
    };
    cp(rowptr, plan->rowptr);
    cp(colidx, plan->colidx);
    cp(cscptr, plan->cscptr);
    cp(csc_row, plan->csc_row);
    cp(csc_k, plan->csc_k);
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_row_k(const ontovae_plan_t* plan, int32_t* row_k) {
    if (!plan || !row_k) return ONTOVAE_EINVAL;
    if (!plan->row_k.empty()) std::memcpy(row_k, plan->row_k.data(), sizeof(int32_t) * plan->row_k.size());
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_sell_sizes(const ontovae_plan_t* plan, int32_t dir, int32_t* n_levels, int32_t* n_slices,
                                       int32_t* n_chunks, int32_t* all_single_chunk) {
    if (!plan || dir < 0 || dir > 1) return ONTOVAE_EINVAL;
    const ontovae_plan::Sell& S = plan->sell[dir];
    if (n_levels) *n_levels = (int32_t)S.level_first.size();
    if (n_slices) *n_slices = (int32_t)(S.node.size() / 64);
    if (n_chunks) *n_chunks = S.chunk0.back();
    if (all_single_chunk) *all_single_chunk = S.all_single;
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_sell_export(const ontovae_plan_t* plan, int32_t dir, int32_t* level_first, int32_t* level_count,
                                        int32_t* node, int32_t* info, int32_t* chunk0, int32_t* idx, int32_t* k) {
    if (!plan || dir < 0 || dir > 1) return ONTOVAE_EINVAL;
    const ontovae_plan::Sell& S = plan->sell[dir];
    auto cp = [](int32_t* dst, const std::vector<int32_t>& v) {
        if (dst && !v.empty()) std::memcpy(dst, v.data(), sizeof(int32_t) * v.size());
    };
    cp(level_first, S.level_first);
    cp(level_count, S.level_count);
    cp(node, S.node);
    cp(info, S.info);
    cp(chunk0, S.chunk0);
    cp(idx, S.idx);
    cp(k, S.k);
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_gsell_sizes(const ontovae_plan_t* plan, int32_t* n_slices, int32_t* n_quads) {
    if (!plan) return ONTOVAE_EINVAL;
    if (n_slices) *n_slices = (int32_t)(plan->gsell.gene.size() / 64);
    if (n_quads) *n_quads = plan->gsell.quad0.back();
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_gsell_export(const ontovae_plan_t* plan, int32_t* gene, int32_t* quad0, uint16_t* col, int32_t* k) {
    if (!plan) return ONTOVAE_EINVAL;
    const ontovae_plan::GSell& S = plan->gsell;
    if (gene && !S.gene.empty()) std::memcpy(gene, S.gene.data(), sizeof(int32_t) * S.gene.size());
    if (quad0) std::memcpy(quad0, S.quad0.data(), sizeof(int32_t) * S.quad0.size());
    if (col && !S.col.empty()) std::memcpy(col, S.col.data(), sizeof(uint16_t) * S.col.size());
    if (k && !S.k.empty()) std::memcpy(k, S.k.data(), sizeof(int32_t) * S.k.size());
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_tile_info(const ontovae_plan_t* plan, int32_t which, int32_t* out) {
    if (!plan || !out || which < 0 || which > 3) return ONTOVAE_EINVAL;
    const std::vector<int32_t>& v = which == 0 ? plan->gene_tiles : (which == 1 ? plan->panel_tiles : (which == 2 ? plan->term_tiles : plan->tpanel_tiles));
    if (which == 3) {          // out[0] = 1 if the structure has term-panel tiles, then the records
        out[0] = v.empty() ? 0 : 1;
        out[1] = 0;                    // the largest sum of list lengths over the tiles (entries a workgroup stages in LDS)
        for (size_t t = 0; t + 127 < v.size(); t += 128) {
            int32_t sum = 0;
            for (int s_ = 0; s_ < 32; ++s_) sum += v[t + (size_t)s_ * 4 + 1];
            out[1] = std::max(out[1], sum);
        }
        if (!v.empty()) std::memcpy(out + 4, v.data(), sizeof(int32_t) * v.size());
        return ONTOVAE_OK;
    }
    if (!v.empty()) std::memcpy(out, v.data(), sizeof(int32_t) * v.size());
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_csc_split(const ontovae_plan_t* plan, int32_t* csc_split) {
    if (!plan || !csc_split) return ONTOVAE_EINVAL;
    const int32_t C = plan->dims.C;
    for (int32_t p = 0; p < C; ++p) {
        const int32_t* b = plan->csc_row.data() + plan->cscptr[p];
        const int32_t* e = plan->csc_row.data() + plan->cscptr[p + 1];
        csc_split[p] = (int32_t)(std::lower_bound(b, e, C) - plan->csc_row.data());
    }
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_entry_coords(const ontovae_plan_t* plan, int32_t* layer, int32_t* row, int32_t* col) {
    if (!plan) return ONTOVAE_EINVAL;
    const size_t n = plan->e_layer.size();
    if (layer && n) std::memcpy(layer, plan->e_layer.data(), n * sizeof(int32_t));
    if (row && n) std::memcpy(row, plan->e_row.data(), n * sizeof(int32_t));
    if (col && n) std::memcpy(col, plan->e_col.data(), n * sizeof(int32_t));
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_layer_node0(const ontovae_plan_t* plan, int32_t layer, int32_t* node0, int32_t* n_out) {
    if (!plan || layer < 0 || layer >= plan->dims.K) {
        ontovae_set_error("layer_node0: bad layer %d", layer);
        return ONTOVAE_EINVAL;
    }
    const int d = layer + 1;  // layer i writes level i+1 (level K = genes)
    if (node0) *node0 = plan->level_begin[d];
    if (n_out) *n_out = plan->level_end[d] - plan->level_begin[d];
    return ONTOVAE_OK;
}

extern "C" int ontovae_plan_unpack_mask(const ontovae_plan_t* plan, int32_t layer, uint8_t* out, int64_t out_len) {
    if (!plan || !out || layer < 0 || layer >= plan->dims.K) {
        ontovae_set_error("unpack_mask: bad arguments");
        return ONTOVAE_EINVAL;
    }
    const int K = plan->dims.K;
    const int32_t C = plan->dims.C;
    const int64_t n_in = plan->cum[layer];
    const int d = layer + 1;
    const int32_t node0 = plan->level_begin[d];
    const int64_t n_out = plan->level_end[d] - node0;
    if (out_len != n_in * n_out) {
        ontovae_set_error("unpack_mask: layer %d is [%lld, %lld], buffer holds %lld", layer, (long long)n_out,
                          (long long)n_in, (long long)out_len);
        return ONTOVAE_EINVAL;
    }
    (void)K;
    std::memset(out, 0, (size_t)out_len);
    const int32_t in0 = C - plan->cum[layer];
    for (int64_t r = 0; r < n_out; ++r)
        for (int32_t q = plan->rowptr[node0 + r]; q < plan->rowptr[node0 + r + 1]; ++q)
            out[r * n_in + (plan->colidx[q] - in0)] = 1;
    return ONTOVAE_OK;
}

extern "C" int ontovae_param_layout(const ontovae_dims_t* d, ontovae_layout_t* out) {
    if (!d || !out) return ONTOVAE_EINVAL;
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };   // 256-B aligned segments
    int64_t o = 0;
    out->W0 = o;       o = up(o + (int64_t)d->L * d->G);
    out->b0 = o;       o = up(o + d->L);
    out->gamma = o;    o = up(o + d->L);
    out->beta = o;     o = up(o + d->L);
    out->Wh = o;       o = up(o + 2LL * d->L * d->L);
    out->bh = o;       o = up(o + 2LL * d->L);
    out->dec_vals = o; o = up(o + d->nnz);
    out->dec_bias = o; o = up(o + (int64_t)d->C + d->G);
    out->total = o;
    return ONTOVAE_OK;
}
