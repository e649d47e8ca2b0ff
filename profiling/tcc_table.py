"""Per-kernel L2 table of the training step (profiles/r02/tcc_table.txt) from the counter passes of profiling/r02_profile.sh and
profiling/r02_tcc.sh and the kernel trace's stats:
    python profiling/tcc_table.py profiles/r02 > profiles/r02/tcc_table.txt
Mean of the second half of the launches of a 6-step run; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950)."""
import csv, sys
from pmc_traffic import per_kernel

d = sys.argv[1]
us = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f"{d}/train_b128_kernel_stats.csv"))}
c = lambda f, n: per_kernel(f"{d}/{f}", n)
hit, miss, req = (c("pmc_tcc_hit_miss_req.csv", n) for n in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))
rd = c("pmc_tcc_ea_rdreq.csv", "TCC_EA0_RDREQ_sum")
wr, wr64 = (c("pmc_tcc_ea_wrreq.csv", n) for n in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
fs, ws = c("pmc_fetch_size.csv", "FETCH_SIZE"), c("pmc_write_size.csv", "WRITE_SIZE")
print("training step, GOBP-syn, batch 128, per launch (mean of the second half of a 6-step run); us from train_b128_kernel_stats.csv")
print(f"{'kernel':44s} {'us':>6s} {'L2 hit%':>7s} {'L2 req':>8s} {'EA rd req':>9s} {'EA wr req':>9s} {'wr 64B':>8s} {'read MB':>8s} {'write MB':>8s}")
for k in sorted((k for k in us if "ov_" in k), key=lambda k: -us[k]):
    g = lambda m: m.get(k, 0.0)
    h = 100 * g(hit) / max(g(hit) + g(miss), 1)
    print(f"{k.split('(')[0].replace('void ', '')[:44]:44s} {us[k]:6.1f} {h:7.1f} {g(req):8.0f} {g(rd):9.0f} {g(wr):9.0f} {g(wr64):8.0f} "
          f"{2 * g(fs) * 1024 / 1e6:8.1f} {g(ws) * 1024 / 1e6:8.1f}")
