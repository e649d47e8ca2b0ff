"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one run each, no trace flags) into
profiles/pmc_traffic.json: HBM-side bytes per launch for every kernel and for bench.py's phases.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is taken as is.
    python profiling/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv
"""
import csv, collections, datetime, json, subprocess, sys

PHASE_OF = {"ov_dw0_kernel": "dw0", "ov_gemm_lds_kernel": "enc_gemm", "ov_gemm_lds2_kernel": "enc_gemm", "ov_gene_cb_kernel": "dec_fwd_gene",
            "ov_bwd_gene_cb_kernel": "dec_bwd_gene", "ov_bwd_gene_term": "dec_bwd_gene", "ov_chain_bwd_finish_kernel": "dec_bwd",
            "ov_dec_chain_kernel": "dec_fwd_terms", "ov_chain_fwd_refresh_kernel": "dec_fwd_terms", "ov_latent_bwd_kernel": "latent_bwd", "ov_da_gemm_kernel": "da_gemm",
            "ov_dwh_bn_kernel": "dwh", "ov_bn_fwd_kernel": "bn_fwd", "ov_reparam_kernel": "reparam", "ov_gemm_kernel<1, 1, 4, 4": "heads"}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # last launches only: the first ones include cold caches
    return {k: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for k, v in acc.items()}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "onto_vae_amd/csrc"], capture_output=True, text=True).stdout.strip())
    except OSError:
        commit, dirty = "unknown", False
    out = {"_meta": {"commit": commit + ("+uncommitted kernel changes" if dirty else ""), "date": datetime.date.today().isoformat(),
                     "sources": [sys.argv[1], sys.argv[2]],
                     "what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiling/one_step.py 6 128 train, NOT of the bench process"},
           "_note": "bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950 correction: FETCH_SIZE doubled); "
                    "mean of the second half of the launches of a 6-step run; Infinity-Cache hits are counted",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if "ov_" not in k:
            continue
        rd, wr = 2 * f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
        short = k.split("(")[0].replace("void ", "")
        out["kernels"][short] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
        for stem, phase in PHASE_OF.items():
            if short.startswith(stem) and phase not in out:
                out[phase] = rd + wr
    # keys this script does not measure (the screen's Wilcoxon kernel, a pass of its own: profiles/r01) are carried over
    prev = {}
    for path in ("profiles/r01/pmc_traffic.json", "profiles/pmc_traffic.json"):
        try:
            prev.update({k: v for k, v in json.load(open(path)).items() if k == "wilcoxon"})
        except (OSError, ValueError):
            pass
    if "wilcoxon" in prev:
        out["wilcoxon"] = prev["wilcoxon"]
        out["_note"] += "; 'wilcoxon' from the round-1 passes (profiles/r01/r01_v8_pmc_wilcoxon_*.csv; that kernel has not changed since)"
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:60]:60s} read {v['read_bytes']/1e6:8.1f} MB write {v['write_bytes']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
