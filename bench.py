#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the GO-BP OntoVAE step (forward + backward + AdamW)
on N MI355X, BASELINE.json configs[1]: GOBP-syn (G=20000, T=3000, K=12, nu=3), fp32, batch 128
per GPU, synthetic data, inputs resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  N > 1 is data parallel, weak scaling: every rank takes `--batch` rows
of a global batch of N * batch (SyncBN + SUM all-reduce over RCCL, onto_vae_amd/parallel.py).
`roofline` describes the dominant kernel of the step (HIP events around its launches inside the
timed region); `cpu_baseline` is the dense CPU oracle timed on this box's host cores (N=1 only).
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


def phase_cost(name, d, B, nnz_t=0):
    """Algorithmic (bytes, flops) of one launch-set of a phase per step; None where it is not modelled.
    nnz_t = non-zeros of the term layers (rows of the node-space operator below C)."""
    L, G, C_, nnz = d.L, d.G, d.C, int(d.nnz)
    f = 4
    if name == "dw0":       # read dh + x, read+write w, m, v of W0 (fused AdamW); 2*L*G*B flops
        return (6 * L * G + B * G + L * B) * f, 2.0 * L * G * B
    if name == "enc_gemm":  # read W0 and x once, write h
        return (L * G + B * G + L * B) * f, 2.0 * L * G * B
    if name == "dec_wgrad":  # per non-zero: w, m, v read+write, index; D and ACT rows once
        return (6 * nnz + nnz + (2 * C_ + G) * B) * f, 2.0 * nnz * B
    if name == "dec_fwd_gene":
        return (2 * nnz + C_ * B + 2 * G * B) * f, 2.0 * nnz * B
    if name == "dec_bwd_gene":   # gene-layer links: value + row index + k; D gene rows read, D panel rows written
        return (3 * (nnz - nnz_t) + (C_ + G) * B) * f, 2.0 * (nnz - nnz_t) * B
    if name == "dec_bwd":        # term levels: packed link records once per 4-column panel pass, D panel read + written
        return (2 * nnz_t * (B // 4) + 2 * C_ * B) * f, 2.0 * nnz_t * B
    return None


def build_model(device):
    from onto_vae_amd.synthetic import StubOntobj, synthetic_masks
    from onto_vae_amd.vae_model import OntoVAE
    masks = synthetic_masks()                                    # GOBP-syn, seed 20250117
    onto = StubOntobj(masks, data={"d": np.zeros((2, masks[-1].shape[0]))}, description="GOBP-syn")
    torch.manual_seed(0)
    model = OntoVAE(onto, "d", neuronnum=3, drop=0.0, z_drop=0.0)   # timing model: dropout off
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
    """The dense CPU oracle (same algorithm as the reference) on the host cores: bounded sample."""
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
    times = []
    t_end = time.time() + budget_s
    while len(times) < 5 or (time.time() < t_end and len(times) < 12):
        t0 = time.time()
        o.train_step(x, eps, 1e-4, 1e-4)
        times.append(time.time() - t0)
        if len(times) >= 3 and time.time() > t_end:
            break
    med = float(np.median(times))
    del limiter
    return {"value": B / med, "unit": "samples/s", "cores": use, "kind": "port",
            "sample": f"{len(times)} train steps of GOBP-syn at batch {B} after 1 warm-up, median {med:.3f} s/step, "
                      f"dense numpy oracle (oracle/ontovae_oracle.py) on {use} threads"}


def infer_bench(a, model, dev, world, rank):
    """Inference throughput: every rank streams its own rows (no collective), activities written to a resident
    [rows, T] output; the kernels are the eval-mode encoder, the term chain and the nu-mean (no gene layer)."""
    import torch.distributed as dist
    e = model._engine
    d = e.plan.dims
    B = a.batch if a.batch != 128 else 16384          # 5.56 / 5.73 / 5.84 M samples/s at 4096 / 8192 / 16384 rows per batch
    rows = max(a.rows // B, 1) * B
    model.eval()
    X = synth_X(rows, d.G, dev, seed=1234 + rank)
    out = torch.empty(rows, e.plan.T, dtype=torch.float32, device=dev)
    nb = rows // B
    it = [0]

    def step():
        i = it[0] % nb
        e.pass_data(X[i * B:(i + 1) * B], None, None, None, act_out=out[i * B:(i + 1) * B])
        it[0] += 1

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out[:B]).all()
    if rank == 0:
        value = B * world * a.steps / dt
        # per sample: encoder 2 G L + heads 4 L^2 flops (MFMA-bound); 4 G bytes in, 4 T out
        flops = (2.0 * d.G * d.L + 4.0 * d.L * d.L) * B
        t_mfma = flops / (PEAK_F32_MFMA_TF * 1e12)
        sec = dt / a.steps
        print(json.dumps({
            "metric": "inference samples/sec (GO-BP OntoVAE get_pathway_activities)", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[3]: GOBP-syn get_pathway_activities ('act' output, [rows, 3000] fp32), "
                                   f"batches of {B} rows from {rows} HBM-resident rows per GPU", "batch_per_gpu": B,
                       "dataset_rows": rows, "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                         "frac": t_mfma / sec, "traffic": None, "kernel": "whole step (encoder GEMM dominates)",
                         "algorithmic_flops": flops}}))
    if world > 1:
        dist.destroy_process_group()


def _pmc_traffic(key):
    """HBM-side bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def screen_bench(a, model, dev, world, rank):
    """In-silico knock-out screen: a step is ONE single-gene perturbation of N resident samples:
    the perturbed pass from the cached first layer (ontovae_pass_data_perturbed), the move to feature-major, and the
    paired Wilcoxon test of all T terms against the control activities; the Benjamini-Hochberg pass over all steps'
    p-values closes the timed region, as OntoVAE.perturbation_screen does.  Ranks screen disjoint gene ranges."""
    import torch.distributed as dist
    from onto_vae_amd import stats as dstats
    e = model._engine
    d = e.plan.dims
    N = a.batch if a.batch != 128 else 4096
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
    P = a.warmup + a.steps
    stat = torch.empty(P, T, dtype=torch.float64, device=dev)
    pval = torch.empty(P, T, dtype=torch.float64, device=dev)
    zero = np.zeros(1, np.float32)
    it = [0]

    def step(full=False):
        i = it[0]
        g = (rank * 2500 + i) % d.G
        if full:          # the reference's recipe: overwrite the column, run the whole pass
            col = X[:, g].clone()
            X[:, g] = 0.0
            e.pass_data(X, None, None, None, act_out=stage)
            X[:, g] = col
        else:
            e.pass_data_perturbed(X, None, hpre, 0, np.array([g], np.int32), zero, None, None, act_fm=pert)
        if full:
            dstats.to_feature_major(stage, pert)
        dstats.wilcoxon_zsplit(pert, ctrl, N, "greater", out=(stat[i % P], pval[i % P]))
        it[0] += 1

    def timed(n, full):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step(full)
        q = dstats.fdr_bh(pval[:max(n, 1)])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, q

    for _ in range(a.warmup):
        step()
    it[0] = 0
    dt, q = timed(a.steps, False)
    assert torch.isfinite(q).all() and torch.isfinite(stat[:a.steps]).all()
    it[0] = 0
    nfull = max(min(a.steps // 4, 50), 1)
    step(True)
    dt_full, _ = timed(nfull, True)
    # Wilcoxon kernel alone, under events on the launch stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        dstats.wilcoxon_zsplit(pert, ctrl, N, "greater", out=(stat[0], pval[0]))
    ev[1].record()
    torch.cuda.synchronize()
    wx_ms = ev[0].elapsed_time(ev[1]) / 10
    if rank == 0:
        sec = dt / a.steps
        # the Wilcoxon kernel reads both [T, N] matrices once: 8 N T bytes
        wx_bytes = 8.0 * N * T
        print(json.dumps({
            "metric": "perturbations/sec (GO-BP single-gene knock-out screen: perturbed pass + paired Wilcoxon of all terms + FDR)",
            "value": world * a.steps / dt, "unit": "perturbations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GOBP-syn, {N} HBM-resident samples, one knocked-out gene per step, {T} terms tested per step, "
                                   "Benjamini-Hochberg over all steps at the end", "samples": N,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "full_pass_ms_per_step": dt_full / nfull * 1e3, "wilcoxon_kernel_ms": wx_ms},
            "roofline": {"bound": "hbm", "achieved": wx_bytes / (wx_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": wx_bytes / (wx_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": _pmc_traffic("wilcoxon"),
                         "kernel": "ov_wilcoxon_sort_kernel (bitonic sort in registers: bound by its vector instructions, not HBM)",
                         "algorithmic_bytes": wx_bytes}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="rows per GPU")
    ap.add_argument("--rows", type=int, default=65536, help="synthetic dataset rows resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-phase times to stderr")
    ap.add_argument("--mode", choices=["train", "infer", "screen"], default="train",
                    help="infer: BASELINE.json configs[3], get_pathway_activities-equivalent (ontovae_pass_data, 'act' output) over "
                         "the HBM-resident rows in batches of --batch (default 16384 there); a step is one batch")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the OntoVAE hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from onto_vae_amd import _lib
    from onto_vae_amd.vae_model import FusedAdamW
    from onto_vae_amd.parallel import dp_train_step
    model, masks = build_model(dev)
    e = model._engine
    d = e.plan.dims
    if a.mode == "infer":
        return infer_bench(a, model, dev, world, rank)
    if a.mode == "screen":
        return screen_bench(a, model, dev, world, rank)
    B = a.batch
    Bg = B * world
    X = synth_X(a.rows, d.G, dev)
    perm = torch.randperm(a.rows, generator=torch.Generator().manual_seed(1)).to(dev).to(torch.int32)
    nb = a.rows // Bg
    opt = FusedAdamW(model, lr=1e-4)
    kc = 1e-4
    loss = torch.zeros(a.warmup + a.steps + 64, 3, device=dev)
    L = _lib.lib()
    it = [0]

    def step():
        i = it[0]
        idx = perm[(i % nb) * Bg:(i % nb + 1) * Bg]
        if world > 1:
            dp_train_step(model, X, idx, None, kc, opt, loss[i % loss.shape[0]], reduce_loss=False)
        else:
            e.train_step(X, idx, None, kc, opt.next_hyper(), loss[i % loss.shape[0]])
        it[0] += 1

    # warm-up; its first part also finds the dominant phase with all phases under HIP events
    nprobe = min(10, max(a.warmup // 2, 1))
    for _ in range(a.warmup - nprobe):
        step()
    nph = len(_lib.PHASES)
    _lib.check(L.ontovae_profile_begin((1 << nph) - 1, 64 * nprobe), "profile_begin")
    for _ in range(nprobe):
        step()
    torch.cuda.synchronize()
    buf = (C.c_float * (64 * nprobe))()
    cnt = C.c_int32()
    phase_ms = {}
    for ph, name in enumerate(_lib.PHASES):
        _lib.check(L.ontovae_profile_read(ph, buf, 64 * nprobe, C.byref(cnt)), "profile_read")
        phase_ms[name] = sum(buf[i] for i in range(cnt.value)) / nprobe
    dom = max(phase_ms, key=phase_ms.get)
    if a.breakdown and rank == 0:
        print("per-phase ms/step (event-bracketed, includes launch gaps):", {k: round(v, 4) for k, v in phase_ms.items()},
              "sum", round(sum(phase_ms.values()), 4), file=sys.stderr)
    _lib.check(L.ontovae_profile_begin(1 << _lib.PHASES.index(dom), 16 * a.steps), "profile_begin")

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kbuf = (C.c_float * (16 * a.steps))()
    _lib.check(L.ontovae_profile_read(_lib.PHASES.index(dom), kbuf, 16 * a.steps, C.byref(cnt)), "profile_read")
    launches = cnt.value
    dom_ms_step = sum(kbuf[i] for i in range(launches)) / a.steps
    L.ontovae_profile_end()
    final = loss[:min(it[0], loss.shape[0])].cpu().numpy()
    assert np.isfinite(final).all(), "non-finite loss in the benchmark run"

    if rank == 0:
        value = Bg * a.steps / dt
        # data parallel: every rank forms dW0 of the GLOBAL batch (parallel.py), the other kernels see the local rows
        cost = phase_cost(dom, d, Bg if dom == "dw0" else B, int(e.plan.rowptr[d.C]))
        roof = None
        if cost is not None:
            nbytes, flops = cost
            t_hbm = nbytes / (PEAK_HBM_GBS * 1e9)
            t_mfma = flops / (PEAK_F32_MFMA_TF * 1e12)
            sec = dom_ms_step * 1e-3
            if t_hbm >= t_mfma:
                ach = nbytes / sec / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
            else:
                ach = flops / sec / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TF}
            roof.update({"traffic": None, "kernel": dom, "kernel_ms_per_step": dom_ms_step, "launches_per_step": launches / a.steps,
                         "algorithmic_bytes": nbytes, "algorithmic_flops": flops})
            tr = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tr):
                try:
                    roof["traffic"] = json.load(open(tr)).get(dom)
                except Exception:
                    pass
        # whole-step roofline of BASELINE.md: max(F_s*B / mfma, (40 P + 4 G B) / hbm) / t_step
        P_real = d.L * d.G + 2 * d.L * d.L + int(d.nnz) + 6 * d.L + (d.C - d.L) + d.G
        t_bound = max((2.0 * 2 * d.L * d.G + 3 * 4.0 * d.L * d.L + 3 * 2.0 * int(d.nnz)) * B / (PEAK_F32_MFMA_TF * 1e12),
                      (40.0 * P_real + 4.0 * d.G * B) / (PEAK_HBM_GBS * 1e9))
        out = {"metric": "train samples/sec (GO-BP OntoVAE fwd+bwd+step)", "value": value, "unit": "samples/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]: GOBP-syn OntoVAE (G=20000, T=3000, K=12, nu=3, L=360) "
                                      "fp32 train step, batch 128 per GPU, lr 1e-4, kl_coeff 1e-4, dropout off",
                          "batch_per_gpu": B, "global_batch": Bg, "dataset_rows": a.rows,
                          "parallelism": f"dp{world}" if world > 1 else "single"},
               "roofline": roof, "step_roofline_frac": t_bound / (dt / a.steps),
               "phase_ms_per_step": {k: round(v, 5) for k, v in phase_ms.items()},
               "final_loss": float(final[-1, 2])}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(masks, B)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
