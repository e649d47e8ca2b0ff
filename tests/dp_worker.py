"""Worker of tests/test_gpu_parity.py::test_dp_two_processes_one_gpu: rank `argv[1]` of `argv[2]` processes that share
GPU 0 and talk through gloo (NCCL/RCCL refuses two ranks on one device); runs `argv[5]` data-parallel steps on the
inputs in `argv[4]` and saves its parameters.  The code path is the one bench.py --gpus N takes, collectives included."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # a hung collective ends the worker with a traceback, not the box's limit
    rank, world, port, path, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    backend = os.environ.get("ONTOVAE_DP_WORKER_BACKEND", "gloo")      # "nccl" (RCCL) only with one rank per GPU: world size 1 here
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    elif os.environ.get("ONTOVAE_DP_WORKER_GLOO_THREADS"):         # diagnosis (profiling/dp_hang_probe.sh): gloo's worker-thread count
        opts = dist.ProcessGroupGloo._Options()
        opts._threads = int(os.environ["ONTOVAE_DP_WORKER_GLOO_THREADS"])
        opts._devices = [dist.ProcessGroupGloo.create_default_device()]
        dist.init_process_group("gloo", rank=rank, world_size=world, pg_options=opts)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from onto_vae_amd.synthetic import StubOntobj
    from onto_vae_amd.vae_model import OntoVAE, FusedAdamW
    from onto_vae_amd.parallel import dp_train_step
    z = np.load(path, allow_pickle=False)
    gobp = "gobp" in z.files          # the full-size model: masks, weights and data regenerated from their seeds, not shipped
    if gobp:
        from onto_vae_amd.synthetic import synthetic_masks, synthetic_expression, random_state_dict
        masks, rng = synthetic_masks(return_rng=True)
        Xh = synthetic_expression(rng, 2048, 20000).astype(np.float32)
        sd = random_state_dict(masks, 3, 77, dense=False)
    else:
        masks = [z[f"mask{i}"] for i in range(int(z["nmask"]))]
        Xh = z["X"]
        sd = {k[3:]: z[k] for k in z.files if k.startswith("sd:")}
    onto = StubOntobj(masks, data={"d": np.zeros((2, masks[-1].shape[0]))})
    torch.manual_seed(0)
    model = OntoVAE(onto, "d", neuronnum=int(z["nu"]), drop=0.0, z_drop=0.0)
    model.to("cuda:0")
    if os.environ.get("ONTOVAE_DP_WORKER_PRECISION") == "bf16":      # BASELINE.json configs[4]: the bf16-MFMA variant, data parallel
        model.set_precision("bf16")
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    del sd
    X = torch.as_tensor(Xh).cuda()
    idx = torch.as_tensor(z["idx"]).cuda()
    eps = torch.as_tensor(z["eps"])
    B = int(z["B"])
    opt = FusedAdamW(model, lr=float(z["lr"]))
    # start the ranks together on an idle device, as a torchrun-launched training loop does after building its model
    torch.cuda.synchronize()
    dist.barrier()
    trace = os.environ.get("ONTOVAE_DP_WORKER_TRACE") == "1"
    if trace:
        import time
        print(f"[dp_worker {rank}] model ready at {time.time():.1f}", file=sys.stderr, flush=True)
    calls = []
    if os.environ.get("ONTOVAE_DP_WORKER_CALLS") == "1":
        # the order in which a step issues collectives, launches the row panels of dW0 and waits: what must stay outstanding
        # under which launches is a property of this sequence (tests/test_gpu_configs.py::test_dp_async_handles_*)
        from onto_vae_amd import parallel
        from onto_vae_amd.engine import Engine
        staged = parallel._HostStaged.get(torch.device("cuda", 0))
        submit0, dw0_rows0, apply0 = staged.submit, Engine.dw0_rows, Engine.adamw_apply_range

        def submit(kind, t, out, group):
            h = submit0(kind, t, out, group)
            tag = f"{kind}:{t.numel()}"
            calls.append("issue " + tag)
            wait0 = h.wait

            def wait():
                calls.append(("wait " if h.done.is_set() else "wait(pending) ") + tag)
                wait0()
            h.wait = wait
            return h

        def dw0_rows(self, x, xidx, l0, l1):
            calls.append(f"dw0_rows {l0}:{l1}")
            return dw0_rows0(self, x, xidx, l0, l1)

        def apply_range(self, adam, begin, end):
            calls.append(f"adamw {begin}:{end}")
            return apply0(self, adam, begin, end)
        staged.submit, Engine.dw0_rows, Engine.adamw_apply_range = submit, dw0_rows, apply_range
    if os.environ.get("ONTOVAE_DP_WORKER_WATCHDOG"):
        # diagnosis (profiling/dp_hang_probe.sh): after N seconds report, without blocking, whether everything launched on the
        # compute stream has finished (hipStreamQuery) -- a stalled collective with an idle compute stream is not a stalled kernel
        import threading
        import time as _t

        def watch():
            _t.sleep(float(os.environ["ONTOVAE_DP_WORKER_WATCHDOG"]))
            torch.cuda.set_device(0)
            for i in range(3):
                print(f"[dp_worker {rank}] watchdog: compute stream idle = {torch.cuda.default_stream().query()}", file=sys.stderr, flush=True)
                _t.sleep(5)
        threading.Thread(target=watch, daemon=True).start()
    losses = torch.zeros(steps, 3, device="cuda")
    for s in range(steps):
        ii = idx[s * B:(s + 1) * B]
        if int(z["use_idx"]):
            dp_train_step(model, X, ii, eps[s], 1e-4, opt, losses[s])
        else:
            dp_train_step(model, X[ii.long()], None, eps[s], 1e-4, opt, losses[s])
        if trace:
            torch.cuda.synchronize()
            print(f"[dp_worker {rank}] step {s} done at {time.time():.1f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    np.savez(path + f".rank{rank}.npz", P=model._engine.P.cpu().numpy(), losses=losses.cpu().numpy(), calls=np.array(calls, dtype=str))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
