#!/usr/bin/env python3
"""Benchmarks of the OntoVAE hot path on MI355X (GOBP-syn: G=20000, T=3000, K=12, nu=3, fp32, synthetic data resident in HBM).

    python bench.py [--gpus 1] [--steps 200] [--warmup 20]
        BASELINE.json configs[1]: train step (forward + backward + AdamW) at batch 128 on one GPU.  The same JSON line also
        carries (extra keys, measured after the timed region): the inference benchmark (configs[3]), the perturbation
        screen, and the single-GPU step at the global batch of configs[2] (8192 rows).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
        BASELINE.json configs[2]: data-parallel training at GLOBAL batch 8192 (8192 / N rows per GPU, SyncBN + SUM
        exchange over RCCL, onto_vae_amd/parallel.py); total work per step is fixed, so "scaling" is "strong".  Rank 0
        first times the same global batch on its GPU alone (key single_gpu_same_global_batch): the curve's own baseline.
    python bench.py --global-batch 8192          one GPU at configs[2]'s global batch
    python bench.py --batch B [--gpus N]         weak scaling at B rows per GPU (round 1's multi-GPU mode)
    python bench.py --mode infer | screen        configs[3] / the knock-out screen alone

One JSON line on rank 0.  `value` = rows of all ranks / max-over-ranks wall time of exactly --steps steps, bracketed by
barrier + device synchronise.  `roofline` describes the dominant kernel of the step: its time is the mean of HIP-event
pairs recorded by the library around that kernel's launches INSIDE the timed region, on the launch stream
(ontovae_profile_*); `frac` = algorithmic bytes (or flops) / that time / peak.  `roofline_kernels` lists every launch of
the step the same way (events recorded during the second half of the warm-up) with `traffic` = HBM-side bytes per launch
from this round's rocprofv3 --pmc passes (profiles/pmc_traffic.json).  `cpu_baseline` is the dense CPU oracle timed on
this box's host cores (N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
GLOBAL_BATCH = 8192          # BASELINE.json configs[2]


PEAK_BF16_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (the headline 5 PF figure includes 2:1 sparsity)


def step_regime(B, C_=9000):
    """Which launch carries which rider at batch B (ov_train_tail_a in csrc/ontovae_kernels.hip: fused_gw, ride_fin,
    ride_lat; ov_ldb): the gene layer's weight gradient is always formed inside its own two kernels (bias partials in the
    forward one, link partials in the backward one); its FINISH pass (sum of the partials, AdamW, clamp) rides with the
    backward chain while that launch has <= 128 workgroups, with the latent backward from 512 columns on, and with the
    da launch in between.  ncb = 32-column blocks of the padded batch; lds = forward gene layer on the LDS-panel kernel."""
    q = 64 * (1 if B <= 64 else (2 if B <= 128 else 4))
    ldb = (B + q - 1) // q * q
    if ldb >= 4096 and ((ldb >> 8) & 1) == 0:
        ldb += 256
    lds = 160 * 1024 - 2048
    bt = 4 if C_ * 16 <= lds else (2 if C_ * 8 <= lds else 1)          # ov_fused_bt: batch columns of the panel a workgroup holds
    fin = "dec_bwd" if ldb // bt <= 128 else ("latent_bwd" if ldb >= 512 else "da_gemm")
    return {"ldb": ldb, "ncb": ldb // 32, "gene_finish_in": fin, "lds": ldb >= 512, "levels": ldb >= 4096}


def phase_cost(name, d, B, nnz_t, dtype="f32"):
    """Algorithmic (HBM bytes, flops) of the launch(es) of one phase of a training step at batch B; None where a phase has no
    launch of its own.  nnz_t = non-zeros of the term layers, nnz_g of the gene layer.  Every array is counted once per
    launch that must read or write it (the least a launch could move: nothing that an L2 hit or a register could save is
    charged), indices as 4 bytes, riders charged to the launch that carries them in this batch regime (step_regime)."""
    L, G, C_, nnz = d.L, d.G, d.C, int(d.nnz)
    nnz_g = nnz - nnz_t
    r = step_regime(B, C_)
    ncb = r["ncb"]
    ncp = min(ncb, 8)          # partial copies of the gene layer's weight gradient: a workgroup of the backward gene kernel adds up the column blocks of its XCD
    f = 4
    nterm = C_ - L                                                       # term rows of the decoder (their biases too)
    gb_part = G * (B // 4 if r["lds"] else ncb)                          # gene bias partials: per column group of 4 (LDS kernel) or per block
    gene_finish = ((nnz_g * ncp + gb_part) + 6 * (nnz_g + G)) * f        # partials in; w, m, v of links and biases in and out
    gene_finish_fl = 12.0 * (nnz_g + G)
    # the term layers' weight-gradient partials: one copy per 32 columns from riders of the latent backward, or (one launch per level,
    # from 4096 columns) one per 256 columns from the backward level kernels themselves
    lv = r["levels"]
    nct = r["ldb"] // 256 if lv else ncb
    term_finish = ((nnz_t + nterm) * nct + 6 * (nnz_t + nterm)) * f
    cost = {
        "enc_gemm": ((L * G + B * G + L * B) * f, 2.0 * L * G * B),                 # W0, x once, h out
        "bn_fwd": (3 * L * B * f, 10.0 * L * B),                                    # h in, xhat + a out
        "heads": ((2 * L * L + L * B + 2 * L * B) * f, 4.0 * L * L * B),
        "reparam": (6 * L * B * f, 12.0 * L * B),                                   # mu, log_var, eps in; z, eps copy, mu/lv out
        # term levels: links (index + weight), z in, panel out; rider: the forward-ordered copy of the gene weights
        "dec_fwd_terms": ((2 * nnz_t + L * B + C_ * B + 2 * nnz_g) * f, 2.0 * nnz_t * B),
        # gene layer + ELBO: links, panel in, x in, dL/drecon out, gene bias partials out
        "dec_fwd_gene": ((2 * nnz_g + C_ * B + 2 * G * B + gb_part) * f, 2.0 * nnz_g * B),
        # gene part of the backward + the gene layer's weight-gradient partials: links, D genes in, D panel out, partials out
        "dec_bwd_gene": ((2 * nnz_g + (C_ + G) * B + nnz_g * ncp) * f, 4.0 * nnz_g * B),
        # term part of the backward: links, D panel in and out
        "dec_bwd": ((2 * nnz_t + 2 * C_ * B + (C_ * B + (nnz_t + nterm) * nct if lv else 0)) * f, (4.0 if lv else 2.0) * nnz_t * B),
        # d mu / d log_var; rider: weight-gradient partials of the term rows (panel and D rows in, partials out)
        "latent_bwd": ((6 * L * B + (0 if lv else (C_ + nterm) * B + (nnz_t + nterm) * nct + 2 * nnz_t)) * f,
                       16.0 * L * B + (0.0 if lv else 2.0 * nnz_t * B)),
        # da = Wh^T [dmu; dlv]; rider: finish pass of the term layers
        "da_gemm": ((2 * L * L + 3 * L * B) * f + term_finish, 4.0 * L * L * B + 12.0 * (nnz_t + nterm)),
        "dwh": ((6 * 2 * L * L + 3 * L * B + 4 * L * B) * f, 4.0 * L * L * B),      # dWh + AdamW, BatchNorm backward
        "dw0": ((6 * L * G + B * G + L * B) * f, 2.0 * L * G * B),                  # w, m, v read + written, x, dh
    }
    b_, f_ = cost[r["gene_finish_in"]]
    cost[r["gene_finish_in"]] = (b_ + gene_finish, f_ + gene_finish_fl)
    return cost.get(name)


MFMA_PHASES = ("enc_gemm", "heads", "da_gemm", "dwh", "dw0")
BF16_PHASES = ("enc_gemm", "dw0")          # the two products the bf16 variant moves to bf16 MFMA (csrc/ontovae_bf16.h)


def roofline_entry(name, ms, d, B, nnz_t, traffic, dtype="f32", overhead_ms=0.0):
    """One launch (or launch group) against the peak that bounds it.  `ms` is the HIP-event bracket; `us` / `frac` are net of
    the bracket's own cost (overhead_ms, measured by ontovae_profile_overhead on the same stream), `us_events` / `frac_events`
    the raw bracket."""
    cost = phase_cost(name, d, B, nnz_t, dtype)
    if cost is None or ms <= 0:
        return None
    nbytes, flops = cost
    peak_tf = PEAK_BF16_MFMA_TF if (dtype == "bf16" and name in BF16_PHASES) else PEAK_F32_MFMA_TF
    t_hbm, t_mfma = nbytes / (PEAK_HBM_GBS * 1e9), flops / (peak_tf * 1e12)
    bound = "mfma" if (name in MFMA_PHASES and t_mfma > t_hbm) else "hbm"
    t_peak = t_mfma if bound == "mfma" else t_hbm
    sec_raw = ms * 1e-3
    sec = max(sec_raw - overhead_ms * 1e-3, 0.5 * sec_raw)
    e = {"kernel": name, "us": round(sec * 1e6, 2), "us_events": round(sec_raw * 1e6, 2), "bound": bound,
         "algorithmic_bytes": int(nbytes), "algorithmic_flops": flops, "GBps": round(nbytes / sec / 1e9, 1),
         "TFLOPs": round(flops / sec / 1e12, 2), "peak": peak_tf * 1e3 if bound == "mfma" else PEAK_HBM_GBS,
         "peak_unit": "GFLOP/s" if bound == "mfma" else "GB/s",
         "frac": round(t_peak / sec, 4), "frac_events": round(t_peak / sec_raw, 4), "traffic": traffic.get(name)}
    if e["frac"] > 1.0:
        e["exceeds_peak"] = True          # the cost model charges this launch more than it can move: a bug in phase_cost
        print(f"bench.py: roofline fraction {e['frac']} > 1 for {name} at batch {B} ({dtype})", file=sys.stderr)
    return e


ONTOLOGY, DTYPE = "gobp", "f32"          # set from the command line (--ontology, --dtype)


def build_model(device):
    from onto_vae_amd.synthetic import StubOntobj, synthetic_masks, FULLGO_SYN_COUNTS, FULLGO_SYN_SEED
    from onto_vae_amd.vae_model import OntoVAE
    if ONTOLOGY == "fullgo":                                     # Full-GO-syn: T = 4500, K = 14
        masks = synthetic_masks(FULLGO_SYN_COUNTS, 20000, FULLGO_SYN_SEED)
    else:
        masks = synthetic_masks()                                # GOBP-syn, seed 20250117
    onto = StubOntobj(masks, data={"d": np.zeros((2, masks[-1].shape[0]))}, description=ONTOLOGY + "-syn")
    torch.manual_seed(0)
    model = OntoVAE(onto, "d", neuronnum=3, drop=0.0, z_drop=0.0)   # timing model: dropout off
    if DTYPE == "bf16":
        model.set_precision("bf16")                              # configs[4]: bf16 MFMA for x W0^T and dW0, the rest fp32
    return model, masks


def synth_X(n, G, device, seed=1234):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.empty(n, G, dtype=torch.float32, device=device)
    for i in range(0, n, 8192):
        r = torch.full((min(8192, n - i), G), 0.3, device=device)
        X[i:i + r.shape[0]] = torch.log1p(torch.poisson(r, generator=g))
    return X


def cpu_baseline(masks, B, budget_s=20.0):
    """The dense CPU oracle (the reference's algorithm: dense masks, dense Linear per depth + concat, dense gradient mask, dense
    AdamW over all 223 M parameters, dense clamp) on the host cores, BASELINE.md's protocol: 1 warm-up step, median of >= 5
    steps, the five-phase split, thread count stated.  Bounded sample (~20 s).  numpy (OpenBLAS) rather than torch on purpose:
    the oracle is the checker of the tests, and it is what is timed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from ontovae_oracle import OracleOntoVAE
    from onto_vae_amd.synthetic import random_state_dict
    cores = os.cpu_count() or 1
    use = min(cores, 16)                       # the GPU box gives this job 16 CPUs
    torch.set_num_threads(use)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=use)
    except Exception:
        limiter = None
    o = OracleOntoVAE(masks, 3)
    o.load_state_dict(random_state_dict(masks, 3, 77, dense=False))
    rng = np.random.RandomState(0)
    G, L = masks[-1].shape[0], o.latent_dim
    x = np.log1p(rng.poisson(0.3, size=(B, G))).astype(np.float32)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    o.train_step(x, eps, 1e-4, 1e-4)           # warm-up
    times, phases = [], []
    t_end = time.time() + budget_s
    while len(times) < 5 or (time.time() < t_end and len(times) < 12):
        o.phase_seconds = {}
        t0 = time.time()
        o.train_step(x, eps, 1e-4, 1e-4)
        times.append(time.time() - t0)
        phases.append(o.phase_seconds)
        if len(times) >= 5 and time.time() > t_end:
            break
    o.phase_seconds = None
    med = float(np.median(times))
    split = {k: round(float(np.median([p[k] for p in phases])), 4) for k in ("forward", "backward", "grad_mask", "adamw", "clamp")}
    del limiter
    return {"value": B / med, "unit": "samples/s", "cores": use, "host_cpus": cores, "kind": "port",
            "phase_seconds": split,
            "sample": f"{len(times)} train steps of GOBP-syn at batch {B} after 1 warm-up, median {med:.3f} s/step, "
                      f"dense numpy oracle (oracle/ontovae_oracle.py; numpy/OpenBLAS, not torch) on {use} threads of {cores} host CPUs; "
                      "phase_seconds = median seconds per step of forward / backward / gradient mask / AdamW / clamp"}


def _pmc_traffic():
    """HBM-side bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: separate rocprofv3 --pmc runs of
    profiling/one_step.py, not of this process; `_meta` says on which commit), {} if absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}


def event_overhead_ms(dev, n=200):
    """(null bracket, empty-kernel bracket) in ms: what two events cost by themselves on the launch stream."""
    from onto_vae_amd import _lib
    a, b = C.c_float(), C.c_float()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.lib().ontovae_profile_overhead(st, n, C.byref(a), C.byref(b)), "profile_overhead")
    return float(a.value), float(b.value)


def _timed(fn, steps, world, dev):
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


INFER_ROWS = 1_000_000       # BASELINE.json configs[3]: get_pathway_activities over 1 M synthetic samples


def infer_bench(a, model, dev, world, rank, steps=None, warmup=None, rows=None):
    """Inference throughput (BASELINE.json configs[3]): every rank streams its own `rows` HBM-resident samples (1 000 000 by
    default: 80 GB of X generated on the device in chunks, no collective) through ontovae_pass_data in batches, activities
    written to a resident [rows, T] output; the kernels are the eval-mode encoder, the term levels and the nu-mean (no gene
    layer).  steps=None: ONE pass over the whole dataset (ceil(rows / batch) batches, the last one partial) is the timed
    region and a step is one batch; otherwise `steps` batches, cycling."""
    e = model._engine
    d = e.plan.dims
    warmup = a.warmup if warmup is None else warmup
    B = a.batch if a.batch not in (None, 128) else 16384          # 5.56 / 5.73 / 5.84 M samples/s at 4096 / 8192 / 16384 rows per batch
    rows = int(rows if rows is not None else (a.rows if a.rows is not None else INFER_ROWS))
    rows = max(rows, B)
    model.eval()
    X = synth_X(rows, d.G, dev, seed=1234 + rank)
    out = torch.empty(rows, e.plan.T, dtype=torch.float32, device=dev)
    nb = (rows + B - 1) // B
    whole = steps is None
    steps = nb if whole else steps
    it = [0]

    def step():
        i = it[0] % nb
        e.pass_data(X[i * B:(i + 1) * B], None, None, None, act_out=out[i * B:(i + 1) * B])
        it[0] += 1

    for _ in range(warmup):
        step()
    it[0] = 0
    dt = _timed(step, steps, world, dev)
    assert torch.isfinite(out[:B]).all() and torch.isfinite(out[-B:]).all()
    done = sum(min(B, rows - (i % nb) * B) for i in range(steps))          # rows the timed region processed on this rank
    value = done * world / dt
    # per sample: encoder 2 G L + heads 4 L^2 flops (MFMA-bound); 4 G bytes in, 4 T out
    flops = (2.0 * d.G * d.L + 4.0 * d.L * d.L) * done
    peak = PEAK_BF16_MFMA_TF if DTYPE == "bf16" else PEAK_F32_MFMA_TF
    # bf16: the encoder product at the bf16 peak is shorter than the stream of x through HBM, which then bounds the pass
    t_mfma, t_hbm = flops / (peak * 1e12), 4.0 * (d.G + e.plan.T) * done / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    del X, out
    torch.cuda.empty_cache()
    return {
        "metric": "inference samples/sec (GO-BP OntoVAE get_pathway_activities)", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[3]: GOBP-syn get_pathway_activities ('act' output, [rows, 3000] fp32) over "
                               f"{rows} HBM-resident rows per GPU in batches of {B}" + (", one pass over the whole dataset" if whole else ""),
                   "batch_per_gpu": B, "dataset_rows": rows, "rows_timed": done, "seconds": dt,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        "roofline": {"bound": bound, "achieved": (flops / dt / 1e12) if bound == "mfma" else 4.0 * (d.G + e.plan.T) * done / dt / 1e9,
                     "peak": peak if bound == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                     "frac": max(t_mfma, t_hbm) / dt, "traffic": None, "kernel": "whole pass (encoder GEMM dominates)",
                     "algorithmic_flops": flops}}


def host_infer_bench(model, dev, N=32768):
    """get_pathway_activities as a user calls it: the dataset is a float64 numpy array in HOST memory (what Ontobj.data
    holds, onto_vae/ontobj.py:495) and the result a numpy array; OntoVAE._pass_data pipelines conversion to float32 into
    pinned memory, the copy up, the kernels and the copy down.  PCIe-inclusive: never the headline `value`."""
    e = model._engine
    X = synth_X(N, e.plan.G, dev, seed=4321).cpu().double().numpy()
    model._pass_data(X[:4096], 'act')                     # warm-up (streams, pinned buffers of the allocator)
    t0 = time.perf_counter()
    act = model._pass_data(X, 'act')
    dt = time.perf_counter() - t0
    assert act.shape == (N, e.plan.T) and np.isfinite(act[:16]).all()
    return {"metric": "inference samples/sec from a host-resident float64 dataset (PCIe-inclusive)", "value": N / dt, "unit": "samples/s",
            "rows": N, "seconds": dt, "batch": int(model.inference_batch),
            "host_to_device_GBps": N * e.plan.G * 4 / dt / 1e9}


def screen_bench(a, model, dev, world, rank, steps=None, warmup=None):
    """In-silico knock-out screen: a step is ONE single-gene perturbation of N resident samples:
    the perturbed pass from the cached first layer (ontovae_pass_data_perturbed), the move to feature-major, and the
    paired Wilcoxon test of all T terms against the control activities; the Benjamini-Hochberg pass over all steps'
    p-values closes the timed region, as OntoVAE.perturbation_screen does.  Ranks screen disjoint gene ranges."""
    from onto_vae_amd import stats as dstats
    e = model._engine
    d = e.plan.dims
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    N = a.batch if a.batch not in (None, 128) else 4096
    model.eval()
    X = synth_X(N, d.G, dev, seed=77)
    T = e.plan.T
    hpre = torch.empty(d.L, N, dtype=torch.float32, device=dev)
    stage = torch.empty(N, T, dtype=torch.float32, device=dev)
    ctrl = torch.zeros(T, N, dtype=torch.float32, device=dev)
    pert = torch.zeros(T, N, dtype=torch.float32, device=dev)
    none_g, none_v = np.zeros(0, np.int32), np.zeros(0, np.float32)
    e.encode_pre(X, None, hpre, 0)
    e.pass_data_perturbed(X, None, hpre, 0, none_g, none_v, None, None, act_out=stage)
    dstats.to_feature_major(stage, ctrl)
    P = warmup + steps
    stat = torch.empty(P, T, dtype=torch.float64, device=dev)
    pval = torch.empty(P, T, dtype=torch.float64, device=dev)
    zero = np.zeros(1, np.float32)
    it = [0]
    q = [None]

    def step(full=False):
        i = it[0]
        g = (rank * 2500 + i) % d.G
        if full:          # the reference's recipe: overwrite the column, run the whole pass
            col = X[:, g].clone()
            X[:, g] = 0.0
            e.pass_data(X, None, None, None, act_out=stage)
            X[:, g] = col
            dstats.to_feature_major(stage, pert)
        else:
            e.pass_data_perturbed(X, None, hpre, 0, np.array([g], np.int32), zero, None, None, act_fm=pert)
        dstats.wilcoxon_zsplit(pert, ctrl, N, "greater", out=(stat[i % P], pval[i % P]))
        it[0] += 1

    def run(n, full):
        def body():
            step(full)
            if it[0] == n:
                q[0] = dstats.fdr_bh(pval[:max(n, 1)])
        it[0] = 0
        return _timed(body, n, world, dev)

    for _ in range(warmup):
        step()
    dt = run(steps, False)
    assert torch.isfinite(q[0]).all() and torch.isfinite(stat[:steps]).all()
    nfull = max(min(steps // 4, 50), 1)
    step(True)
    dt_full = run(nfull, True)
    # Wilcoxon kernel alone, under events on the launch stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        dstats.wilcoxon_zsplit(pert, ctrl, N, "greater", out=(stat[0], pval[0]))
    ev[1].record()
    torch.cuda.synchronize()
    wx_ms = ev[0].elapsed_time(ev[1]) / 10
    sec = dt / steps
    wx_bytes = 8.0 * N * T          # the Wilcoxon kernel reads both [T, N] matrices once
    return {
        "metric": "perturbations/sec (GO-BP single-gene knock-out screen: perturbed pass + paired Wilcoxon of all terms + FDR)",
        "value": world * steps / dt, "unit": "perturbations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"GOBP-syn, {N} HBM-resident samples, one knocked-out gene per step, {T} terms tested per step, "
                               "Benjamini-Hochberg over all steps at the end", "samples": N,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "full_pass_ms_per_step": dt_full / nfull * 1e3, "wilcoxon_kernel_ms": wx_ms},
        "roofline": {"bound": "hbm", "achieved": wx_bytes / (wx_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": wx_bytes / (wx_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": _pmc_traffic().get("wilcoxon"),
                     "kernel": "ov_wilcoxon_sort_kernel (bitonic sort in registers: bound by its vector instructions, not HBM)",
                     "algorithmic_bytes": wx_bytes}}


def train_bench(model, dev, world, rank, B, steps, warmup, rows, profile=True, single=False, want_launches=False):
    """`steps` training steps at B rows per rank (global batch B * world; single=True: this rank alone, no collectives).
    Returns (seconds of the timed region [max over ranks], per-phase ms per step, dominant phase, its ms per step inside
    the timed region, its launches per step, final loss)."""
    from onto_vae_amd import _lib
    from onto_vae_amd.vae_model import FusedAdamW
    from onto_vae_amd.parallel import dp_train_step
    e = model._engine
    d = e.plan.dims
    w = 1 if single else world
    Bg = B * w
    rows = max(rows, Bg)
    X = synth_X(rows, d.G, dev)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(1)).to(dev).to(torch.int32)
    nb = rows // Bg
    opt = FusedAdamW(model, lr=1e-4)
    kc = 1e-4
    loss = torch.zeros(warmup + steps + 64, 3, device=dev)
    L = _lib.lib()
    it = [0]

    def step():
        i = it[0]
        idx = perm[(i % nb) * Bg:(i % nb + 1) * Bg]
        if w > 1:
            dp_train_step(model, X, idx, None, kc, opt, loss[i % loss.shape[0]], reduce_loss=False)
        else:
            e.train_step(X, idx, None, kc, opt.next_hyper(), loss[i % loss.shape[0]])
        it[0] += 1

    nprobe = min(10, max(warmup // 2, 1))
    for _ in range(warmup - nprobe):
        step()
    phase_ms, phase_launches, dom, dom_ms_step, launches = {}, {}, None, 0.0, 0
    cnt = C.c_int32()
    if profile:
        # the second part of the warm-up runs with every phase under HIP events: per-kernel times and the dominant phase
        nph = len(_lib.PHASES)
        _lib.check(L.ontovae_profile_begin((1 << nph) - 1, 64 * nprobe), "profile_begin")
    for _ in range(nprobe):
        step()
    torch.cuda.synchronize()
    if profile:
        buf = (C.c_float * (64 * nprobe))()
        for ph, name in enumerate(_lib.PHASES):
            _lib.check(L.ontovae_profile_read(ph, buf, 64 * nprobe, C.byref(cnt)), "profile_read")
            phase_ms[name] = sum(buf[i] for i in range(cnt.value)) / nprobe
            phase_launches[name] = cnt.value / nprobe
        dom = max(phase_ms, key=phase_ms.get)
        # dW0 on one GPU: differential timing (include/ontovae_hip.h, ONTOVAE_PHASE_PAIR_A) -- alternate steps bracket the
        # launch in front of dW0 alone and that launch plus dW0, with no event between the two kernels; the difference of the
        # means is dW0 as it runs in an un-instrumented step.  Other kernels: one bracket per launch.
        pair = dom == "dw0" and w == 1 and steps >= 8
        mask = (1 << _lib.PHASE_PAIR_A) if pair else (1 << _lib.PHASES.index(dom))
        _lib.check(L.ontovae_profile_begin(mask, 16 * steps), "profile_begin")
    dt = _timed(step, steps, w, dev)
    pair_info = None
    if profile:
        kbuf = (C.c_float * (16 * steps))()
        if pair:
            means = []
            for ph in (_lib.PHASE_PAIR_A, _lib.PHASE_PAIR_AB):
                _lib.check(L.ontovae_profile_read(ph, kbuf, 16 * steps, C.byref(cnt)), "profile_read")
                v = np.array([kbuf[i] for i in range(cnt.value)], np.float64)
                means.append((float(v.mean()), float(v.std() / np.sqrt(max(len(v), 1))), len(v)))
            launches = steps
            dom_ms_step = means[1][0] - means[0][0]
            pair_info = {"front_launch_ms": means[0][0], "front_launch_plus_dw0_ms": means[1][0],
                         "stderr_ms": float(np.hypot(means[0][1], means[1][1])), "brackets": [means[0][2], means[1][2]]}
        else:
            _lib.check(L.ontovae_profile_read(_lib.PHASES.index(dom), kbuf, 16 * steps, C.byref(cnt)), "profile_read")
            launches = cnt.value
            dom_ms_step = sum(kbuf[i] for i in range(launches)) / steps
        L.ontovae_profile_end()
    final = loss[:min(it[0], loss.shape[0])].cpu().numpy()
    assert np.isfinite(final).all(), "non-finite loss in the benchmark run"
    del X
    res = (dt, phase_ms, dom, dom_ms_step, launches / max(steps, 1), float(final[-1, 2]))
    return res + (phase_launches, pair_info) if want_launches else res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default: 128 on one GPU, 8192 / N on N)")
    ap.add_argument("--global-batch", type=int, default=None, help="global batch (one GPU: the whole batch on it)")
    ap.add_argument("--rows", type=int, default=None, help="synthetic dataset rows resident in HBM (default: 65536 for training, "
                                                           "1 000 000 for --mode infer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="train mode, one GPU: skip the inference / screen / global-batch legs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-phase times to stderr")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: BASELINE.json configs[4], the two large dense products on bf16 MFMA (fp32 accumulation, fp32 weights)")
    ap.add_argument("--ontology", choices=["gobp", "fullgo"], default="gobp", help="fullgo: the deeper Full-GO-syn of configs[4]")
    ap.add_argument("--mode", choices=["train", "infer", "screen"], default="train",
                    help="infer: BASELINE.json configs[3], get_pathway_activities-equivalent (ontovae_pass_data, 'act' output) over "
                         "the HBM-resident rows in batches of --batch (default 16384 there); a step is one batch")
    a = ap.parse_args()
    global ONTOLOGY, DTYPE
    ONTOLOGY, DTYPE = a.ontology, a.dtype

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the OntoVAE hot path has no CPU fallback")
    # ONTOVAE_BENCH_SHARED_GPU=1: rehearsal of the multi-rank code path on a box with ONE GPU (all ranks on cuda:0, gloo with
    # host-staged collectives).  Its numbers mean nothing; the line says so.
    rehearsal = world > 1 and os.environ.get("ONTOVAE_BENCH_SHARED_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    model, masks = build_model(dev)
    e = model._engine
    d = e.plan.dims
    if a.mode in ("infer", "screen"):
        if a.mode == "infer":
            res = infer_bench(a, model, dev, world, rank, steps=None if a.steps == 200 else a.steps)
        else:
            res = screen_bench(a, model, dev, world, rank)
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            dist.destroy_process_group()
        return

    # which workload: configs[1] on one GPU, configs[2] on several (or with --global-batch), --batch overrides both
    if a.batch is not None:
        B, scaling = a.batch, "weak"
        workload = (f"GOBP-syn OntoVAE fp32 train step, {B} rows per GPU (weak scaling)" if (world > 1 or B != 128) else None)
    elif a.global_batch is not None or world > 1:
        gb = a.global_batch or GLOBAL_BATCH
        if gb % world:
            raise SystemExit(f"global batch {gb} is not a multiple of {world} ranks")
        B, scaling = gb // world, "strong"
        workload = (f"BASELINE.json configs[2]: GOBP-syn OntoVAE (G=20000, T=3000, K=12, nu=3, L=360) fp32 data-parallel train "
                    f"step, global batch {gb} = {world} x {B} rows, SyncBN + SUM gradient exchange, lr 1e-4, kl_coeff 1e-4, dropout off")
    else:
        B, scaling, workload = 128, "weak", None
    if workload is None:
        workload = ("BASELINE.json configs[1]: GOBP-syn OntoVAE (G=20000, T=3000, K=12, nu=3, L=360) "
                    "fp32 train step, batch 128 per GPU, lr 1e-4, kl_coeff 1e-4, dropout off")
    if a.dtype == "bf16" or a.ontology == "fullgo":
        workload = (f"BASELINE.json configs[4]: {'Full-GO-syn (T=4500, K=14, L=600)' if a.ontology == 'fullgo' else 'GOBP-syn'} OntoVAE, "
                    f"{'bf16 MFMA variant (x W0^T and dW0 on bf16 inputs, fp32 accumulation, fp32 master weights and AdamW)' if a.dtype == 'bf16' else 'fp32'}"
                    f", global batch {B * world} = {world} x {B} rows")
    Bg = B * world
    nnz_t = int(e.plan.rowptr[d.C])
    train_rows = a.rows if a.rows is not None else 65536

    single_same = None
    if world > 1 and scaling == "strong":
        # the curve's own baseline: the same global batch on ONE GPU (rank 0 alone; the other ranks wait at the barrier)
        if rank == 0:
            n1 = max(min(a.steps // 4, 50), 10)
            dt1, *_ = train_bench(model, dev, world, rank, Bg, n1, 5, train_rows, profile=False, single=True)
            single_same = {"value": Bg * n1 / dt1, "unit": "samples/s", "ms_per_step": dt1 / n1 * 1e3, "steps": n1, "n_gpus": 1,
                           "global_batch": Bg}
        model, masks = build_model(dev)          # fresh weights and optimizer state on every rank
        e = model._engine
        dist.barrier()

    dt, phase_ms, dom, dom_ms_step, lps, final_loss, phase_launches, pair_info = train_bench(model, dev, world, rank, B, a.steps, a.warmup, train_rows,
                                                                                  want_launches=True)
    if a.breakdown and rank == 0:
        print("per-phase ms/step (event-bracketed, includes launch gaps):", {k: round(v, 4) for k, v in phase_ms.items()},
              "sum", round(sum(phase_ms.values()), 4), file=sys.stderr)

    if rank == 0:
        value = Bg * a.steps / dt
        traffic = _pmc_traffic() if (B == 128 and world == 1 and a.dtype == "f32" and a.ontology == "gobp") else {}
        tmeta = traffic.get("_meta") if traffic else None
        # What a bracket adds to the kernel inside it.  Launches are back to back on one stream (rocprofv3 traces: no gaps), so
        # the un-bracketed step of the timed region is the sum of its kernels, and the brackets of the probe steps add up to
        # that sum plus one bracket cost per launch: cost = (sum of brackets - wall step) / launches.  A bracket that holds
        # nothing (ontovae_profile_overhead: ~3.4 us on this pool) only bounds it from above -- most of it overlaps the kernel.
        ov_null, ov_empty = event_overhead_ms(dev)
        n_launch = sum(max(v, 0.0) for v in phase_launches.values()) or 1.0
        ov_est = min(max((sum(phase_ms.values()) - dt / a.steps * 1e3) / n_launch, 0.0), ov_null) if world == 1 else 0.0
        # data parallel in the gathered-dh form: every rank forms dW0 of the GLOBAL batch (parallel.py)
        from onto_vae_amd.parallel import use_gathered_dh
        dw0_rows = Bg if (world > 1 and use_gathered_dh(Bg, world, d.L, d.G)) else B
        rows_of = lambda name: dw0_rows if name == "dw0" else B
        # launches per step of a phase (term levels run as one launch per level from 2048 columns on): brackets to subtract
        launches_of = lambda name: max(phase_launches.get(name, 1.0), 1.0)
        kernels = [r for r in (roofline_entry(n, ms, d, rows_of(n), nnz_t, traffic, a.dtype, ov_est * launches_of(n))
                               for n, ms in phase_ms.items()) if r]
        roof = None
        if pair_info is not None:
            # differential timing: nothing to subtract; the raw single-launch bracket of the probe steps beside it
            domr = roofline_entry(dom, dom_ms_step, d, rows_of(dom), nnz_t, traffic, a.dtype, 0.0)
            ev = phase_ms[dom]
            domr["us_events"], domr["frac_events"] = round(ev * 1e3, 2), round(domr["frac"] * dom_ms_step / ev, 4)
            timer = ("HIP events on the launch stream inside the timed region, differential: even steps bracket the launch in "
                     "front of dW0 (dWh + BatchNorm backward) alone, odd steps that launch and dW0 in one bracket with no event "
                     "between the kernels; kernel_ms_per_step = mean(odd) - mean(even) and frac follow from it (dW0 as it runs in "
                     "an un-instrumented step, its launch gap included); *_events = one bracket around dW0 alone in the probe "
                     "steps, which reads 1-3 us more (the events hold the launch back)")
        else:
            domr = roofline_entry(dom, dom_ms_step, d, rows_of(dom), nnz_t, traffic, a.dtype, ov_est * lps)
            timer = ("HIP events on the launch stream around every launch of the kernel inside the timed region; "
                     "kernel_ms_per_step and frac are net of the bracket cost per launch, estimated live as (sum of all "
                     "phases' brackets in the probe steps - wall time of an un-bracketed step) / launches per step; "
                     "*_events are the raw brackets; null / empty-kernel brackets (ontovae_profile_overhead) are upper bounds")
        if domr is not None:
            unit = "TFLOP/s" if domr["bound"] == "mfma" else "GB/s"
            ach = domr["TFLOPs"] if domr["bound"] == "mfma" else domr["GBps"]
            roof = {"bound": domr["bound"], "achieved": ach, "peak": domr["peak"] / (1e3 if domr["bound"] == "mfma" else 1.0),
                    "unit": unit, "frac": domr["frac"], "frac_events": domr["frac_events"], "traffic": domr["traffic"],
                    "traffic_source": tmeta, "kernel": dom,
                    "kernel_ms_per_step": domr["us"] * 1e-3, "kernel_ms_per_step_events": domr["us_events"] * 1e-3,
                    "differential": pair_info,
                    "event_bracket_overhead_ms": {"per_launch_estimate": ov_est, "null_bracket": ov_null, "empty_kernel_bracket": ov_empty},
                    "launches_per_step": lps, "timer": timer,
                    "algorithmic_bytes": domr["algorithmic_bytes"], "algorithmic_flops": domr["algorithmic_flops"]}
        # whole-step roofline of BASELINE.md: max(F_s*B / mfma, (40 P + 4 G B) / hbm) / t_step
        P_real = d.L * d.G + 2 * d.L * d.L + int(d.nnz) + 6 * d.L + (d.C - d.L) + d.G
        # (bf16 variant: the two first-layer products at the bf16 peak, the rest at the fp32 one)
        enc_peak = PEAK_BF16_MFMA_TF if a.dtype == "bf16" else PEAK_F32_MFMA_TF
        t_bound = max(2.0 * 2 * d.L * d.G * B / (enc_peak * 1e12) + (3 * 4.0 * d.L * d.L + 3 * 2.0 * int(d.nnz)) * B / (PEAK_F32_MFMA_TF * 1e12),
                      (40.0 * P_real + 4.0 * d.G * B) / (PEAK_HBM_GBS * 1e9))
        out = {"metric": "train samples/sec (GO-BP OntoVAE fwd+bwd+step)", "value": value, "unit": "samples/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": workload, "batch_per_gpu": B, "global_batch": Bg, "dataset_rows": max(train_rows, Bg),
                          "parallelism": f"dp{world}" if world > 1 else "single"},
               "roofline": roof, "step_roofline_frac": t_bound / (dt / a.steps), "step_roofline_bound_ms": t_bound * 1e3,
               "roofline_kernels": kernels,
               "phase_ms_per_step": {k: round(v, 5) for k, v in phase_ms.items()},
               "final_loss": final_loss}
        if single_same is not None:
            out["single_gpu_same_global_batch"] = single_same
        if rehearsal:
            out["rehearsal"] = "all ranks share cuda:0, gloo with host-staged collectives: code-path check only, not a measurement"
        if world == 1 and not a.no_extras and B == 128 and a.global_batch is None and a.dtype == "f32" and a.ontology == "gobp":
            # the other headline workloads, measured after the timed region so that the driver's run covers them too
            xa = argparse.Namespace(**vars(a))
            xa.batch, xa.rows = None, None
            inf = infer_bench(xa, model, dev, 1, 0, steps=None, warmup=3, rows=INFER_ROWS)
            out["inference"] = {k: inf[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline")}
            out["inference_from_host"] = host_infer_bench(model, dev)
            scr = screen_bench(xa, model, dev, 1, 0, steps=100, warmup=10)
            out["screen"] = {k: scr[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline")}
            m2, _ = build_model(dev)
            n1 = 30
            dt1, ph1, _, _, _, _, pl1, _ = train_bench(m2, dev, 1, 0, GLOBAL_BATCH, n1, 6, train_rows, want_launches=True)
            out["global_batch_8192_single_gpu"] = {
                "metric": "train samples/sec at configs[2]'s global batch on one GPU", "value": GLOBAL_BATCH * n1 / dt1,
                "unit": "samples/s", "ms_per_step": dt1 / n1 * 1e3, "steps": n1,
                "roofline_kernels": [r for r in (roofline_entry(n, ms, d, GLOBAL_BATCH, nnz_t, {}, "f32", 0.0)
                                                 for n, ms in ph1.items()) if r]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(masks, 128)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
