// The plain VAE sibling of OntoVAE (reference onto_vae/vae_model.py:913-1140: class VAE = Encoder + dense Decoder,
// onto_vae/modules.py:15-84, 94-154), on the same kernels as the ontology model: fp32 MFMA GEMMs (ontovae_gemm.h), the
// feature-major BatchNorm / reparameterisation / latent-backward kernels of ontovae_kernels.hip.  Included at the end of
// that file (it uses its internal launchers).
//
//   encoder   x -> [Linear -> BatchNorm -> Dropout -> ReLU] x n_enc -> (mu, log_var) = Dropout(z_drop)(Linear)   (modules.py:43-64)
//   z = mu + eps * exp(log_var / 2)                                                                              (vae_model.py:961-963)
//   decoder   z -> [Linear -> BatchNorm -> Dropout -> ReLU] x n_dec -> Linear -> reconstruction                  (modules.py:116-133)
//   loss = sum (recon - x)^2 + kl_coeff * KL                                                                     (vae_model.py:995-1004)
//   AdamW over every parameter (vae_model.py:1101), no clamp, no masks.
//
// Activations are feature-major [F, ldb] like everywhere in the library.  Layer l of a stack computes
//   H = W a_prev  (split-K slabs)  ->  ov_bn_fwd_kernel (slab sum + bias + BatchNorm + dropout + ReLU)  ->  xhat_l, a_l
// and its backward is ov_bn_bwd_kernel (dh, AdamW of gamma / beta / bias) followed by da_prev = W^T dh and dW = dh a_prev^T
// (+ AdamW in the GEMM epilogue).
#ifndef ONTOVAE_DENSE_H
#define ONTOVAE_DENSE_H

#define OV_VAE_MAXL ONTOVAE_VAE_MAX_LAYERS

struct OvVaeLayer { int in, out; int64_t W, b, gamma, beta; int64_t bn; int bn_index; };    // offsets in floats into P / bn_state
struct OvVaeNet {
    int G, Z, ne, nd;
    OvVaeLayer enc[OV_VAE_MAXL], dec[OV_VAE_MAXL];
    int64_t Wh, bh, Wout, bout, total, bn_total;
    int fmax;                 // widest hidden layer
};

static int ov_vae_net(const ontovae_vae_dims_t& d, OvVaeNet* n) {
    if (d.G < 1 || d.Z < 1 || d.n_enc < 1 || d.n_enc > OV_VAE_MAXL || d.n_dec < 1 || d.n_dec > OV_VAE_MAXL) {
        ontovae_set_error("vae dims: G = %d, Z = %d, n_enc = %d, n_dec = %d (1 .. %d layers each)", d.G, d.Z, d.n_enc, d.n_dec, OV_VAE_MAXL);
        return ONTOVAE_EINVAL;
    }
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };
    int64_t o = 0, bo = 0;
    int bi = 0, fmax = 0;
    n->G = d.G; n->Z = d.Z; n->ne = d.n_enc; n->nd = d.n_dec;
    auto layer = [&](OvVaeLayer& l, int in, int out) {
        l.in = in; l.out = out;
        l.W = o; o = up(o + (int64_t)out * in);
        l.b = o; o = up(o + out);
        l.gamma = o; o = up(o + out);
        l.beta = o; o = up(o + out);
        l.bn = bo; bo += 2 * (int64_t)out;       // running_mean [out] then running_var [out]
        l.bn_index = bi++;
        fmax = std::max(fmax, out);
    };
    int in = d.G;
    for (int i = 0; i < d.n_enc; ++i) {
        if (d.enc[i] < 1) { ontovae_set_error("vae dims: encoder layer %d has width %d", i, d.enc[i]); return ONTOVAE_EINVAL; }
        layer(n->enc[i], in, d.enc[i]);
        in = d.enc[i];
    }
    n->Wh = o; o = up(o + 2LL * d.Z * in);
    n->bh = o; o = up(o + 2LL * d.Z);
    in = d.Z;
    for (int i = 0; i < d.n_dec; ++i) {
        if (d.dec[i] < 1) { ontovae_set_error("vae dims: decoder layer %d has width %d", i, d.dec[i]); return ONTOVAE_EINVAL; }
        layer(n->dec[i], in, d.dec[i]);
        in = d.dec[i];
    }
    n->Wout = o; o = up(o + (int64_t)d.G * in);
    n->bout = o; o = up(o + d.G);
    n->total = o; n->bn_total = bo; n->fmax = std::max(fmax, 2 * d.Z);
    return ONTOVAE_OK;
}

extern "C" int ontovae_vae_layout(const ontovae_vae_dims_t* dims, ontovae_vae_layout_t* out) {
    if (!dims || !out) { ontovae_set_error("vae_layout: NULL argument"); return ONTOVAE_EINVAL; }
    OvVaeNet n;
    int rc = ov_vae_net(*dims, &n);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < n.ne; ++i) { out->enc_W[i] = n.enc[i].W; out->enc_b[i] = n.enc[i].b; out->enc_gamma[i] = n.enc[i].gamma; out->enc_beta[i] = n.enc[i].beta; out->enc_bn[i] = n.enc[i].bn; }
    for (int i = 0; i < n.nd; ++i) { out->dec_W[i] = n.dec[i].W; out->dec_b[i] = n.dec[i].b; out->dec_gamma[i] = n.dec[i].gamma; out->dec_beta[i] = n.dec[i].beta; out->dec_bn[i] = n.dec[i].bn; }
    out->Wh = n.Wh; out->bh = n.bh; out->Wout = n.Wout; out->bout = n.bout; out->total = n.total; out->bn_total = n.bn_total;
    return ONTOVAE_OK;
}

struct OvVaeWs {
    float *slab, *ga, *dh, *mulv, *dmulv, *epsw, *z, *dz, *rec, *bwd_local, *rec_part, *kl_part;
    float *xhat[2 * OV_VAE_MAXL], *a[2 * OV_VAE_MAXL], *mean[2 * OV_VAE_MAXL], *invstd[2 * OV_VAE_MAXL];   // by bn_index
    int64_t slab_floats;
    int npart;
};
#define OV_VAE_SMAX 64
static int ov_vae_split(int tiles, int K, int target) {
    return std::max(1, std::min(std::min((target + tiles - 1) / tiles, OV_VAE_SMAX), std::max(1, K / 256)));
}
static int64_t ov_vae_carve(const OvVaeNet& n, int max_batch, void* base, OvVaeWs* ws) {
    const int64_t ldb = ov_ldb(max_batch);
    int64_t off = 0;
    auto take = [&](int64_t cnt) {
        float* p = base ? reinterpret_cast<float*>(base) + off : nullptr;
        off += (cnt + 63) / 64 * 64;
        return p;
    };
    OvVaeWs w{};
    // split-K slabs: the first encoder layer (K = G) and the backward of the output layer (K = G) are the deep ones
    int64_t slab = 0;
    for (int b = 64; b <= ldb; b += 64) {
        const int t1 = ((n.enc[0].out + 95) / 96) * ((b + 127) / 128);
        const int t2 = ((n.dec[n.nd - 1].out + 31) / 32) * ((b + 31) / 32);
        slab = std::max<int64_t>(slab, (int64_t)ov_vae_split(t1, n.G, 512) * n.enc[0].out * ov_ldb(b));
        slab = std::max<int64_t>(slab, (int64_t)ov_vae_split(t2, n.G, 512) * n.dec[n.nd - 1].out * ov_ldb(b));
    }
    slab = std::max<int64_t>(slab, (int64_t)n.fmax * ldb);
    w.slab = take(slab);
    w.slab_floats = slab;
    w.ga = take(slab);
    w.dh = take((int64_t)n.fmax * ldb);
    w.mulv = take(2LL * n.Z * ldb);
    w.dmulv = take(2LL * n.Z * ldb);
    w.epsw = take((int64_t)n.Z * ldb);
    w.z = take((int64_t)n.Z * ldb);
    w.dz = take((int64_t)n.Z * ldb);
    w.rec = take((int64_t)n.G * ldb);
    w.bwd_local = take(2LL * n.fmax);
    w.npart = (int)(((int64_t)(n.G + 31) / 32) * (ldb / 32) + 4096);
    w.rec_part = take(w.npart);
    w.kl_part = take(w.npart);
    auto per = [&](const OvVaeLayer& l) {
        w.xhat[l.bn_index] = take((int64_t)l.out * ldb);
        w.a[l.bn_index] = take((int64_t)l.out * ldb);
        w.mean[l.bn_index] = take(l.out);
        w.invstd[l.bn_index] = take(l.out);
    };
    for (int i = 0; i < n.ne; ++i) per(n.enc[i]);
    for (int i = 0; i < n.nd; ++i) per(n.dec[i]);
    if (ws) *ws = w;
    return off * (int64_t)sizeof(float);
}

extern "C" int64_t ontovae_vae_workspace_bytes(const ontovae_vae_dims_t* dims, int32_t max_batch) {
    if (!dims || max_batch < 1) return -1;
    OvVaeNet n;
    if (ov_vae_net(*dims, &n)) return -1;
    return ov_vae_carve(n, max_batch, nullptr, nullptr);
}

// recon [G, ldb] feature-major (bias included) against x (sample-major rows, gathered): recon <- dL/drecon = 2 (recon - x)
// in place, block partials of sum (recon - x)^2.  32 x 32 tiles, x transposed through LDS.
__global__ __launch_bounds__(256) void ov_vae_loss_kernel(float* __restrict__ rec, const float* __restrict__ x,
                                                         const int32_t* __restrict__ xidx, int64_t xld, int G, int B, int ldb,
                                                         float* __restrict__ rec_part) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int g0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, g = g0 + tx;
        float v = 0.f;
        if (b < B && g < G) {
            const int64_t r = xidx ? (int64_t)xidx[b] : (int64_t)b;
            v = x[r * xld + g];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    float sq = 0.f;
    for (int j = ty; j < 32; j += 8) {
        const int g = g0 + j, b = b0 + tx;
        if (g < G && b < ldb) {
            float o = 0.f;
            if (b < B) {
                const float dv = rec[(int64_t)g * ldb + b] - tile[tx][j];
                sq += dv * dv;
                o = 2.f * dv;
            }
            rec[(int64_t)g * ldb + b] = o;
        }
    }
    sq = ov_wave_sum(sq);
    if (ov_lane() == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) rec_part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gradient of a bias = row sums of a feature-major gradient matrix; AdamW or store.  One wave per row.
__global__ __launch_bounds__(256) void ov_vae_rowsum_kernel(const float* __restrict__ d, int F, int B, int ldb, float* __restrict__ P,
                                                           float* __restrict__ M, float* __restrict__ V, float* __restrict__ Gr,
                                                           int apply, OvAdam adam) {
    const int f = blockIdx.x * 4 + ov_wave_id();
    if (f >= F) return;
    float s = 0.f;
    for (int b = ov_lane(); b < B; b += 64) s += d[(int64_t)f * ldb + b];
    s = ov_wave_sum(s);
    if (ov_lane() == 0) {
        if (apply) {
            float w = P[f], m = M[f], v = V[f];
            ov_adamw(w, m, v, s, adam, false);
            P[f] = w; M[f] = m; V[f] = v;
        } else {
            Gr[f] = s;
        }
    }
}

// sum of split-K slabs [S][F, ldb] -> dst [F, ldb]
__global__ __launch_bounds__(256) void ov_vae_slabsum_kernel(const float* __restrict__ slab, int S, int64_t stride, int64_t n,
                                                            float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int q = 0; q < S; ++q) v += slab[q * stride + i];
        dst[i] = v;
    }
}

struct OvVaeStep {
    const ontovae_vae_ctx_t* c;
    OvVaeNet n;
    OvVaeWs ws;
    int B, ldb;
    hipStream_t st;
    const float* x; const int32_t* xidx;
    OvAdam adam; int apply;
    uint64_t rng_off;
};

static int ov_vae_check(const ontovae_vae_ctx_t* c, int B, bool need_opt, OvVaeStep* s, const float* x, const int32_t* xidx, void* stream) {
    if (!c || !c->P || !c->bn_state || !c->bn_nbt || !c->workspace) { ontovae_set_error("vae ctx: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    int rc = ov_vae_net(c->dims, &s->n);
    if (rc) return rc;
    if (need_opt && (!c->M || !c->V)) { ontovae_set_error("vae ctx: optimizer state (M, V) is NULL"); return ONTOVAE_EINVAL; }
    if (B < 1 || B > c->max_batch) { ontovae_set_error("vae: batch size %d outside [1, max_batch=%d]", B, c->max_batch); return ONTOVAE_EINVAL; }
    if (c->workspace_bytes < ov_vae_carve(s->n, c->max_batch, nullptr, nullptr) || (reinterpret_cast<uintptr_t>(c->workspace) & 255) ||
        (reinterpret_cast<uintptr_t>(c->P) & 15)) {
        ontovae_set_error("vae ctx: workspace of %lld bytes, 256-B aligned, and a 16-B aligned P required",
                          (long long)ov_vae_carve(s->n, c->max_batch, nullptr, nullptr));
        return ONTOVAE_EINVAL;
    }
    if (!x) { ontovae_set_error("vae: x is NULL"); return ONTOVAE_EINVAL; }
    ov_vae_carve(s->n, c->max_batch, c->workspace, &s->ws);
    s->c = c; s->B = B; s->ldb = ov_ldb(B); s->st = reinterpret_cast<hipStream_t>(stream); s->x = x; s->xidx = xidx;
    s->adam = OvAdam{}; s->apply = 0; s->rng_off = 0;
    return ONTOVAE_OK;
}

static inline uint64_t ov_vae_layer_off(uint64_t rng_off, int bn_index) { return rng_off ^ ((uint64_t)(bn_index + 1) << 48); }

// H = W a_prev as split-K slabs in ws.slab; returns S.  prev == NULL: the first encoder layer, a_prev = the batch rows of x
static int ov_vae_linear_fwd(OvVaeStep& s, const OvVaeLayer& l, const float* W, const float* prev, int* S_out) {
    OvGemm g{};
    g.M = l.out; g.N = s.B; g.K = l.in;
    g.A = W; g.lda = l.in;
    g.C = s.ws.slab; g.ldc = s.ldb; g.slab_stride = (int64_t)l.out * s.ldb;
    int S;
    if (!prev) {
        g.B = s.x; g.ldb = l.in; g.bgather = s.xidx;
        const int tiles = ((l.out + 95) / 96) * ((s.B + 127) / 128);
        S = ov_vae_split(tiles, l.in, 512);
        S = (int)std::min<int64_t>(S, std::max<int64_t>(1, s.ws.slab_floats / g.slab_stride));
        g.kchunk = ((l.in + S - 1) / S + 31) / 32 * 32;
        S = (l.in + g.kchunk - 1) / g.kchunk;
        const bool vec = (l.in % 4 == 0) && ov_al16(g.A) && ov_al16(g.B);
        *S_out = S;
        return ov_launch_gemm<3, 4, 4, 1, true, true, OV_EPI_STORE>(g, S, vec, s.st);
    }
    g.B = prev; g.ldb = s.ldb; g.bgather = nullptr;
    const int tiles = ((l.out + 31) / 32) * ((s.B + 31) / 32);
    S = ov_vae_split(tiles, l.in, 256);
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, s.ws.slab_floats / g.slab_stride));
    g.kchunk = ((l.in + S - 1) / S + 31) / 32 * 32;
    S = (l.in + g.kchunk - 1) / g.kchunk;
    *S_out = S;
    return ov_launch_gemm<1, 1, 4, 4, true, false, OV_EPI_STORE>(g, S, (l.in % 4 == 0) && ov_al16(g.A), s.st);
}

static int ov_vae_block_fwd(OvVaeStep& s, const OvVaeLayer& l, const float* prev, int train) {
    const ontovae_vae_ctx_t* c = s.c;
    int S = 1, rc;
    if ((rc = ov_vae_linear_fwd(s, l, c->P + l.W, prev, &S))) return rc;
    const int k = l.bn_index;
    hipLaunchKernelGGL(ov_bn_fwd_kernel, dim3(l.out), dim3(256), 0, s.st, s.ws.slab, S, (int64_t)l.out * s.ldb, c->P + l.b, c->P + l.gamma,
                       c->P + l.beta, c->bn_state + l.bn, c->bn_state + l.bn + l.out, c->bn_nbt + k, s.ws.xhat[k], s.ws.a[k], s.ws.mean[k],
                       s.ws.invstd[k], (float*)nullptr, l.out, s.B, s.B, s.ldb, train ? 1 : 0, c->drop, c->seed, ov_vae_layer_off(s.rng_off, k));
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// dense Decoder (modules.py:116-153) on ws.z: blocks, then the output layer; leaves recon (with bias) in ws.rec
static int ov_vae_decoder_fwd(OvVaeStep& s, int train) {
    const ontovae_vae_ctx_t* c = s.c;
    const OvVaeNet& n = s.n;
    int rc;
    const float* prev = s.ws.z;
    for (int i = 0; i < n.nd; ++i) {
        OvVaeLayer l = n.dec[i];
        if ((rc = ov_vae_block_fwd(s, l, prev, train))) return rc;
        prev = s.ws.a[l.bn_index];
    }
    // output layer: recon = Wout a + bout (modules.py:127-131), feature-major
    const int Hd = n.dec[n.nd - 1].out;
    OvGemm g{};
    g.M = n.G; g.N = s.B; g.K = Hd;
    g.A = c->P + n.Wout; g.lda = Hd;
    g.B = prev; g.ldb = s.ldb;
    g.C = s.ws.rec; g.ldc = s.ldb; g.slab_stride = 0; g.kchunk = (Hd + 31) / 32 * 32;
    g.bias = c->P + n.bout;
    return ov_launch_gemm<1, 1, 4, 4, true, false, OV_EPI_BIAS>(g, 1, (Hd % 4 == 0) && ov_al16(g.A), s.st);
}

// encoder + heads + reparameterisation + decoder; leaves recon (with bias) in ws.rec, mu / log_var in ws.mulv, z in ws.z
// encoder_only: stop after the reparameterisation (Encoder.forward / get_embedding, modules.py:74-84, vae_model.py:970-981: the
// reference runs no decoder there, so the decoder's BatchNorm buffers must not move in train mode)
static int ov_vae_forward_core(OvVaeStep& s, const float* eps, int train, int write_kl, int* nkl, bool encoder_only = false) {
    const ontovae_vae_ctx_t* c = s.c;
    const OvVaeNet& n = s.n;
    int rc;
    const float* prev = nullptr;
    for (int i = 0; i < n.ne; ++i) {
        if ((rc = ov_vae_block_fwd(s, n.enc[i], prev, train))) return rc;
        prev = s.ws.a[n.enc[i].bn_index];
    }
    // heads: [mu; log_var] = Wh a + bh (modules.py:56-64), bias and z_drop inside ov_reparam_kernel
    const int H = n.enc[n.ne - 1].out;
    {
        OvGemm g{};
        g.M = 2 * n.Z; g.N = s.B; g.K = H;
        g.A = c->P + n.Wh; g.lda = H;
        g.B = prev; g.ldb = s.ldb;
        g.C = s.ws.slab; g.ldc = s.ldb; g.slab_stride = 2LL * n.Z * s.ldb; g.kchunk = (H + 31) / 32 * 32;
        if ((rc = ov_launch_gemm<1, 1, 4, 4, true, false, OV_EPI_STORE>(g, 1, (H % 4 == 0) && ov_al16(g.A), s.st))) return rc;
    }
    {
        const int64_t cnt = (int64_t)n.Z * s.ldb;
        const int blocks = (int)std::min<int64_t>((cnt + 255) / 256, std::min<int64_t>(2048, s.ws.npart));
        OvRefresh rf{};
        hipLaunchKernelGGL(ov_reparam_kernel, dim3(blocks), dim3(256), 0, s.st, s.ws.mulv, eps, train ? s.ws.epsw : (float*)nullptr, s.ws.z,
                           s.ws.kl_part, n.Z, s.B, s.ldb, train, c->z_drop, c->seed, s.rng_off, (uint32_t)OV_RNG_EPS, write_kl, blocks, rf,
                           (const float*)s.ws.slab, 1, 2LL * n.Z * s.ldb, (const float*)(c->P + n.bh));
        OV_LAUNCH_CHECK();
        if (nkl) *nkl = blocks;
    }
    if (encoder_only) return ONTOVAE_OK;
    return ov_vae_decoder_fwd(s, train);
}

static int ov_vae_loss(OvVaeStep& s, int* nrec) {
    const dim3 grid((s.n.G + 31) / 32, s.ldb / 32);
    if ((int64_t)grid.x * grid.y > s.ws.npart) { ontovae_set_error("vae loss: %u x %u partials exceed the workspace", grid.x, grid.y); return ONTOVAE_EINVAL; }
    hipLaunchKernelGGL(ov_vae_loss_kernel, grid, dim3(256), 0, s.st, s.ws.rec, s.x, s.xidx, (int64_t)s.n.G, s.n.G, s.B, s.ldb, s.ws.rec_part);
    OV_LAUNCH_CHECK();
    *nrec = (int)(grid.x * grid.y);
    return ONTOVAE_OK;
}

// da_prev [in, ldb] = W^T dh  (W [out, in], dh [out, ldb]) into dst; deep products through split-K slabs in ws.ga
static int ov_vae_da(OvVaeStep& s, const float* W, int out, int in, const float* dh, float* dst, int* S_out, bool keep_slabs) {
    OvGemm g{};
    g.M = in; g.N = s.B; g.K = out;
    g.A = W; g.lda = in;               // A(m = i, k = j) = W[j * in + i]
    g.B = dh; g.ldb = s.ldb;
    const int tiles = ((in + 31) / 32) * ((s.B + 31) / 32);
    int S = ov_vae_split(tiles, out, 256);
    const int64_t one = (int64_t)in * s.ldb;
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, s.ws.slab_floats / one));
    g.kchunk = ((out + S - 1) / S + 31) / 32 * 32;
    S = (out + g.kchunk - 1) / g.kchunk;
    g.C = (S > 1 || keep_slabs) ? s.ws.ga : dst; g.ldc = s.ldb; g.slab_stride = one;
    int rc = ov_launch_gemm<1, 1, 4, 4, false, false, OV_EPI_STORE>(g, S, false, s.st);
    if (rc) return rc;
    if (S_out) *S_out = S;
    if (!keep_slabs && S > 1) {
        hipLaunchKernelGGL(ov_vae_slabsum_kernel, dim3((unsigned)std::min<int64_t>((one + 255) / 256, 2048)), dim3(256), 0, s.st, s.ws.ga, S, one, one, dst);
        OV_LAUNCH_CHECK();
    }
    return ONTOVAE_OK;
}

// dW [out, in] = dh a_prev^T (+ AdamW); prev == NULL: a_prev = the batch rows of x (first encoder layer)
static int ov_vae_dw(OvVaeStep& s, int64_t Woff, int out, int in, const float* dh, const float* prev) {
    const ontovae_vae_ctx_t* c = s.c;
    OvGemm g{};
    g.M = out; g.N = in; g.K = s.B;
    g.A = dh; g.lda = s.ldb;
    g.ldc = in; g.kchunk = (s.B + 31) / 32 * 32;
    g.adam = s.adam;
    if (s.apply) { g.C = c->P + Woff; g.Pm = c->M + Woff; g.Pv = c->V + Woff; }
    else g.C = c->Gr + Woff;
    if (!prev) {
        g.B = s.x; g.ldb = in; g.bgather = s.xidx;          // B(k = b, n = g) = X[row(b) * G + g]
        if (s.apply) return ov_launch_gemm<1, 4, 1, 2, true, false, OV_EPI_ADAMW>(g, 1, true, s.st);
        return ov_launch_gemm<1, 4, 1, 2, true, false, OV_EPI_STORE>(g, 1, true, s.st);
    }
    g.B = prev; g.ldb = s.ldb;                               // B(k = b, n = i) = a_prev[i * ldb + b]
    if (s.apply) return ov_launch_gemm<1, 1, 4, 2, true, true, OV_EPI_ADAMW>(g, 1, true, s.st);
    return ov_launch_gemm<1, 1, 4, 2, true, true, OV_EPI_STORE>(g, 1, true, s.st);
}

// backward of one [Linear, BatchNorm, Dropout, ReLU] block: ga (daS slabs of the gradient w.r.t. a_l) -> dh in ws.dh,
// AdamW of gamma / beta / bias; then da_prev (into dst, or slabs in ws.ga when keep_slabs) and dW
static int ov_vae_block_bwd(OvVaeStep& s, const OvVaeLayer& l, const float* ga, int daS, const float* prev, float* dprev, int* S_out,
                            bool keep_slabs) {
    const ontovae_vae_ctx_t* c = s.c;
    const int k = l.bn_index;
    float* Gr = c->Gr;
    const OvBnBwdArgs bn{ga, s.ws.a[k], s.ws.xhat[k], s.ws.invstd[k], c->P + l.gamma, c->P + l.beta, c->P + l.b,
                         c->M ? c->M + l.gamma : nullptr, c->M ? c->M + l.beta : nullptr, c->M ? c->M + l.b : nullptr,
                         c->V ? c->V + l.gamma : nullptr, c->V ? c->V + l.beta : nullptr, c->V ? c->V + l.b : nullptr,
                         Gr ? Gr + l.gamma : nullptr, Gr ? Gr + l.beta : nullptr, Gr ? Gr + l.b : nullptr, s.ws.dh, s.ws.bwd_local,
                         (float*)nullptr, l.out, s.B, s.B, s.ldb, 1, s.apply, s.adam, c->drop, c->seed, ov_vae_layer_off(s.rng_off, k), daS,
                         (int64_t)l.out * s.ldb, nullptr, 0};
    hipLaunchKernelGGL(ov_bn_bwd_kernel, dim3(l.out), dim3(256), 0, s.st, bn);
    OV_LAUNCH_CHECK();
    int rc;
    if (dprev && (rc = ov_vae_da(s, c->P + l.W, l.out, l.in, s.ws.dh, dprev, S_out, keep_slabs))) return rc;    // before W is updated
    return ov_vae_dw(s, l.W, l.out, l.in, s.ws.dh, prev);
}

// loss and the backward of the dense Decoder: sum-of-squares partials (nrec of them), every decoder parameter's gradient
// (+ AdamW), dL/dz in ws.dz
static int ov_vae_decoder_bwd(OvVaeStep& s, int* nrec) {
    const ontovae_vae_ctx_t* ctx = s.c;
    const OvVaeNet& n = s.n;
    int rc;
    if ((rc = ov_vae_loss(s, nrec))) return rc;                     // ws.rec now holds dL/drecon
    // output layer: bias gradient, da (before the weights move), dW
    hipLaunchKernelGGL(ov_vae_rowsum_kernel, dim3((n.G + 3) / 4), dim3(256), 0, s.st, s.ws.rec, n.G, s.B, s.ldb, ctx->P + n.bout,
                       ctx->M ? ctx->M + n.bout : nullptr, ctx->V ? ctx->V + n.bout : nullptr, ctx->Gr ? ctx->Gr + n.bout : nullptr, s.apply,
                       s.adam);
    OV_LAUNCH_CHECK();
    const OvVaeLayer& last = n.dec[n.nd - 1];
    int daS = 1;
    if ((rc = ov_vae_da(s, ctx->P + n.Wout, n.G, last.out, s.ws.rec, nullptr, &daS, true))) return rc;
    if ((rc = ov_vae_dw(s, n.Wout, n.G, last.out, s.ws.rec, s.ws.a[last.bn_index]))) return rc;
    // decoder blocks, last to first; the first one hands dz to the latent backward
    for (int i = n.nd - 1; i >= 0; --i) {
        const OvVaeLayer& l = n.dec[i];
        const float* prev = i > 0 ? s.ws.a[n.dec[i - 1].bn_index] : s.ws.z;
        int S2 = 1;
        if ((rc = ov_vae_block_bwd(s, l, s.ws.ga, daS, prev, i > 0 ? s.ws.ga : s.ws.dz, &S2, i > 0))) return rc;
        daS = S2;
    }
    return ONTOVAE_OK;
}

extern "C" int ontovae_vae_train_step(const ontovae_vae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                                      uint64_t rng_offset, float kl_coeff, const ontovae_adamw_t* opt, int32_t apply_update,
                                      float* loss_out, void* stream) {
    OvVaeStep s;
    int rc = ov_vae_check(ctx, B, apply_update != 0, &s, x, xidx, stream);
    if (rc) return rc;
    if (!loss_out) { ontovae_set_error("vae_train_step: loss_out is NULL"); return ONTOVAE_EINVAL; }
    if (apply_update && !opt) { ontovae_set_error("vae_train_step: opt is NULL"); return ONTOVAE_EINVAL; }
    if (!apply_update && !ctx->Gr) { ontovae_set_error("vae_train_step: gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    if (B < 2) { ontovae_set_error("vae_train_step: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    s.adam = ov_adam_from(opt); s.apply = apply_update ? 1 : 0; s.rng_off = rng_offset;
    const OvVaeNet& n = s.n;
    int nkl = 0, nrec = 0;
    if ((rc = ov_vae_forward_core(s, eps, 1, 1, &nkl))) return rc;
    if ((rc = ov_vae_decoder_bwd(s, &nrec))) return rc;
    int daS = 1;
    {
        OvLatentArgs la{s.ws.dz, s.ws.mulv, s.ws.epsw, s.ws.dmulv, n.Z, s.B, s.ldb, kl_coeff, ctx->z_drop, ctx->seed, rng_offset,
                        ctx->P + n.bh, ctx->M ? ctx->M + n.bh : nullptr, ctx->V ? ctx->V + n.bh : nullptr, ctx->Gr ? ctx->Gr + n.bh : nullptr,
                        s.apply, s.adam, OvLossArgs{s.ws.rec_part, nrec, s.ws.kl_part, nkl, kl_coeff, loss_out}, n.Z + 1};
        OvWgradArgs none{};
        OvFinishArgs nofin{};
        hipLaunchKernelGGL(ov_latent_bwd_kernel, dim3(la.nblocks), dim3(256), 0, s.st, la, none, 0, nofin, 0);
        OV_LAUNCH_CHECK();
    }
    // heads: da = Wh^T dmulv (before Wh moves), dWh = dmulv a^T
    const OvVaeLayer& top = n.enc[n.ne - 1];
    if ((rc = ov_vae_da(s, ctx->P + n.Wh, 2 * n.Z, top.out, s.ws.dmulv, nullptr, &daS, true))) return rc;
    if ((rc = ov_vae_dw(s, n.Wh, 2 * n.Z, top.out, s.ws.dmulv, s.ws.a[top.bn_index]))) return rc;
    for (int i = n.ne - 1; i >= 0; --i) {
        const OvVaeLayer& l = n.enc[i];
        const float* prev = i > 0 ? s.ws.a[n.enc[i - 1].bn_index] : nullptr;
        int S2 = 1;
        if ((rc = ov_vae_block_bwd(s, l, s.ws.ga, daS, prev, i > 0 ? s.ws.ga : nullptr, &S2, true))) return rc;
        daS = S2;
    }
    return ONTOVAE_OK;
}

// VAE.forward (vae_model.py:979-990) and, with loss_out, one body of val_round (vae_model.py:1061-1065).
// Outputs (any may be NULL): recon [B, G], mu / log_var [B, Z] sample-major; loss_out: device float[3] {rec, kl, rec + kl_coeff kl}.
extern "C" int ontovae_vae_forward(const ontovae_vae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                                   uint64_t rng_offset, int32_t train, float kl_coeff, float* recon, float* mu, float* log_var,
                                   float* z_out, float* loss_out, void* stream) {
    OvVaeStep s;
    int rc = ov_vae_check(ctx, B, false, &s, x, xidx, stream);
    if (rc) return rc;
    if (train && B < 2) { ontovae_set_error("vae_forward: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    s.rng_off = rng_offset;
    const OvVaeNet& n = s.n;
    int nkl = 0, nrec = 0;
    if ((rc = ov_vae_forward_core(s, eps, train ? 1 : 0, loss_out ? 1 : 0, &nkl, !recon && !loss_out))) return rc;
    OvStep t{};
    t.B = B; t.ldb = s.ldb; t.st = s.st;
    if (recon && (rc = ov_transpose_out(t, s.ws.rec, n.G, recon, n.G))) return rc;
    if (mu && (rc = ov_transpose_out(t, s.ws.mulv, n.Z, mu, n.Z))) return rc;
    if (log_var && (rc = ov_transpose_out(t, s.ws.mulv + (int64_t)n.Z * s.ldb, n.Z, log_var, n.Z))) return rc;
    if (z_out && (rc = ov_transpose_out(t, s.ws.z, n.Z, z_out, n.Z))) return rc;
    if (loss_out) {
        if ((rc = ov_vae_loss(s, &nrec))) return rc;
        hipLaunchKernelGGL(ov_loss_finalize_kernel, dim3(1), dim3(1024), 0, s.st, s.ws.rec_part, nrec, s.ws.kl_part, nkl, kl_coeff, loss_out);
        OV_LAUNCH_CHECK();
    }
    return ONTOVAE_OK;
}

// -------------------------------------------------------------------------------------------------
// OntoEncVAE (reference vae_model.py:468-905): the ontology sits in the ENCODER (OntoEncoder, modules.py:161-238: masked
// linear layers with prepending concatenation and two masked heads over the whole concatenation, no BatchNorm, no ReLU)
// and the decoder is the dense Decoder above.  The masked encoder is the same sparse operator as OntoDecoder with the genes
// as its input: ontovae_decoder_forward / ontovae_decoder_backward on a plan built from the encoder masks give the heads
// [B, 2 Z] = [mu | log_var] and take their gradient.  The two entry points here are everything between those two calls.
// (The dense ctx keeps a one-unit dummy encoder; its parameters are never touched here.)
// -------------------------------------------------------------------------------------------------
// heads [B, 2 Z] sample-major -> ws.mulv (after z_drop in train mode), eps, z, KL partials
static int ov_vae_heads_in(OvVaeStep& s, const float* heads, const float* eps, int train, int write_kl, int* nkl) {
    const ontovae_vae_ctx_t* c = s.c;
    const OvVaeNet& n = s.n;
    dim3 grid((2 * n.Z + 31) / 32, (s.ldb + 31) / 32);
    hipLaunchKernelGGL(ov_transpose_in_kernel, grid, dim3(256), 0, s.st, heads, 2 * n.Z, s.B, s.ldb, s.ws.slab);
    OV_LAUNCH_CHECK();
    OV_HIP_CHECK(hipMemsetAsync(s.ws.dmulv, 0, 2 * (size_t)n.Z * sizeof(float), s.st));      // the heads carry their biases already
    const int64_t cnt = (int64_t)n.Z * s.ldb;
    const int blocks = (int)std::min<int64_t>((cnt + 255) / 256, std::min<int64_t>(2048, s.ws.npart));
    OvRefresh rf{};
    hipLaunchKernelGGL(ov_reparam_kernel, dim3(blocks), dim3(256), 0, s.st, s.ws.mulv, eps, train ? s.ws.epsw : (float*)nullptr, s.ws.z,
                       s.ws.kl_part, n.Z, s.B, s.ldb, train, c->z_drop, c->seed, s.rng_off, (uint32_t)OV_RNG_EPS, write_kl, blocks, rf,
                       (const float*)s.ws.slab, 1, 2LL * n.Z * s.ldb, (const float*)s.ws.dmulv);
    OV_LAUNCH_CHECK();
    if (nkl) *nkl = blocks;
    return ONTOVAE_OK;
}

// OntoEncVAE.forward after the encoder (vae_model.py:557-563) and, with loss_out, the rest of one val_round body (:654-655).
// heads: [B, 2 Z] = [mu | log_var] as OntoEncoder's two Linear layers give them (before their Dropout).  Outputs (any may be
// NULL): recon [B, G], mu / log_var [B, Z] after the heads' Dropout, z [B, Z]; loss_out: device float[3], needs x.
extern "C" int ontovae_vae_decode_forward(const ontovae_vae_ctx_t* ctx, const float* heads, const float* x, const int32_t* xidx,
                                          int32_t B, const float* eps, uint64_t rng_offset, int32_t train, float kl_coeff, float* recon,
                                          float* mu, float* log_var, float* z_out, float* loss_out, void* stream) {
    OvVaeStep s;
    if (!heads) { ontovae_set_error("vae_decode_forward: heads is NULL"); return ONTOVAE_EINVAL; }
    if (loss_out && !x) { ontovae_set_error("vae_decode_forward: the loss needs x"); return ONTOVAE_EINVAL; }
    int rc = ov_vae_check(ctx, B, false, &s, x ? x : heads, xidx, stream);
    if (rc) return rc;
    if (train && B < 2) { ontovae_set_error("vae_decode_forward: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    s.rng_off = rng_offset;
    const OvVaeNet& n = s.n;
    int nkl = 0, nrec = 0;
    if ((rc = ov_vae_heads_in(s, heads, eps, train ? 1 : 0, loss_out ? 1 : 0, &nkl))) return rc;
    if ((recon || loss_out) && (rc = ov_vae_decoder_fwd(s, train ? 1 : 0))) return rc;
    OvStep t{};
    t.B = B; t.ldb = s.ldb; t.st = s.st;
    if (recon && (rc = ov_transpose_out(t, s.ws.rec, n.G, recon, n.G))) return rc;
    if (mu && (rc = ov_transpose_out(t, s.ws.mulv, n.Z, mu, n.Z))) return rc;
    if (log_var && (rc = ov_transpose_out(t, s.ws.mulv + (int64_t)n.Z * s.ldb, n.Z, log_var, n.Z))) return rc;
    if (z_out && (rc = ov_transpose_out(t, s.ws.z, n.Z, z_out, n.Z))) return rc;
    if (loss_out) {
        if ((rc = ov_vae_loss(s, &nrec))) return rc;
        hipLaunchKernelGGL(ov_loss_finalize_kernel, dim3(1), dim3(1024), 0, s.st, s.ws.rec_part, nrec, s.ws.kl_part, nkl, kl_coeff, loss_out);
        OV_LAUNCH_CHECK();
    }
    return ONTOVAE_OK;
}

// The middle of one OntoEncVAE.train_round body (vae_model.py:603-619): reparameterisation, dense Decoder, ELBO, backward down
// to the heads, AdamW of every decoder parameter (apply_update) or their gradients into ctx->Gr.
// dheads_out [B, 2 Z]: dL/d[mu | log_var] at the heads' Linear outputs -- what ontovae_decoder_backward takes.
extern "C" int ontovae_vae_decode_step(const ontovae_vae_ctx_t* ctx, const float* heads, const float* x, const int32_t* xidx, int32_t B,
                                       const float* eps, uint64_t rng_offset, float kl_coeff, const ontovae_adamw_t* opt,
                                       int32_t apply_update, float* dheads_out, float* loss_out, void* stream) {
    OvVaeStep s;
    if (!heads || !dheads_out || !loss_out) { ontovae_set_error("vae_decode_step: heads, dheads_out or loss_out is NULL"); return ONTOVAE_EINVAL; }
    int rc = ov_vae_check(ctx, B, apply_update != 0, &s, x, xidx, stream);
    if (rc) return rc;
    if (apply_update && !opt) { ontovae_set_error("vae_decode_step: opt is NULL"); return ONTOVAE_EINVAL; }
    if (!apply_update && !ctx->Gr) { ontovae_set_error("vae_decode_step: gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    if (B < 2) { ontovae_set_error("vae_decode_step: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    s.adam = ov_adam_from(opt); s.apply = apply_update ? 1 : 0; s.rng_off = rng_offset;
    const OvVaeNet& n = s.n;
    int nkl = 0, nrec = 0;
    if ((rc = ov_vae_heads_in(s, heads, eps, 1, 1, &nkl))) return rc;
    if ((rc = ov_vae_decoder_fwd(s, 1))) return rc;
    if ((rc = ov_vae_decoder_bwd(s, &nrec))) return rc;
    {
        // the heads' biases belong to the masked operator: their gradient (row sums of dmulv) goes to scratch here
        OvLatentArgs la{s.ws.dz, s.ws.mulv, s.ws.epsw, s.ws.dmulv, n.Z, s.B, s.ldb, kl_coeff, ctx->z_drop, ctx->seed, rng_offset,
                        nullptr, nullptr, nullptr, s.ws.bwd_local, 0, s.adam,
                        OvLossArgs{s.ws.rec_part, nrec, s.ws.kl_part, nkl, kl_coeff, loss_out}, n.Z + 1};
        OvWgradArgs none{};
        OvFinishArgs nofin{};
        hipLaunchKernelGGL(ov_latent_bwd_kernel, dim3(la.nblocks), dim3(256), 0, s.st, la, none, 0, nofin, 0);
        OV_LAUNCH_CHECK();
    }
    OvStep t{};
    t.B = B; t.ldb = s.ldb; t.st = s.st;
    return ov_transpose_out(t, s.ws.dmulv, 2 * n.Z, dheads_out, 2 * n.Z);
}

// torch.optim.AdamW.step over the flat buffers (vae_model.py:1101), gradients from ctx->Gr: the unfused path
extern "C" int ontovae_vae_adamw_apply(const ontovae_vae_ctx_t* ctx, const ontovae_adamw_t* opt, void* stream) {
    if (!ctx || !opt || !ctx->P || !ctx->M || !ctx->V || !ctx->Gr) { ontovae_set_error("vae_adamw_apply: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    OvVaeNet n;
    int rc = ov_vae_net(ctx->dims, &n);
    if (rc) return rc;
    const int blocks = (int)std::min<int64_t>((n.total + 255) / 256, 4096);
    hipLaunchKernelGGL(ov_adamw_flat_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ctx->P, ctx->M, ctx->V, ctx->Gr,
                       n.total, (int64_t)-1, (int64_t)-1, ov_adam_from(opt));
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

#endif
