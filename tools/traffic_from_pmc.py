"""HBM bytes per launch per bench stage from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB and gfx950 reports half of wide streaming reads
(MI355X_MICROARCH.md, HBM / rocprofv3 section).   usage: traffic_from_pmc.py <fetch_dir> <write_dir> <config> <out.json>"""
import csv, glob, json, re, sqlite3, sys, collections

STAGE = [  # (regex on the kernel name, stage)
    (r"gate_fwd_kernel", "fwd.gate"),
    (r"gemm_ws_kernel<\d+, [12],", "fwd.vproj"), (r"gate_stats_kernel", "fwd.vproj"), (r"vproj_modal_kernel", "fwd.vproj"),
    (r"vproj_slab_kernel", "fwd.vproj"),
    (r"gemm_ws_kernel<\d+, 0,", "plain_nt"), (r"gemm_nt_kernel", "plain_nt"),
    (r"gemm_tn_tr_kernel<1, false", "bwd.dw_out"), (r"gemm_tn_ring_kernel", "bwd.dw_out"), (r"gemm_tn_hilo_pooled_kernel", "bwd.dw_v"),
    (r"dscore_v_kernel", "bwd.dscore"), (r"dsu_ws_kernel|dsu_slab_kernel", "bwd.dscore"), (r"row_fwd_kernel", "fwd.vproj"),
    (r"dx_ws2?_kernel", "bwd.dx"), (r"bwd_g_kernel", "bwd.dx"),
    (r"gemm_tn_tr_kernel<\d+, true", "bwd.dw_v"), (r"gemm_tn_tr_wide_kernel", "bwd.dw_v"), (r"gemm_tn_u_kernel|u_stream_kernel", "bwd.u"), (r"gemm_tn_kernel", "bwd.dw_v"),
    (r"reduce_segments_kernel|fin_outer_kernel|fin_dquery_kernel|finalize_all_kernel|dqp_kernel", "bwd.finalize"),
    (r"prep_all_kernel", "prep"),
]


def per_kernel(d, counter):
    """average counter value per kernel name, from rocprofv3's csv output or its rocpd sqlite output"""
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(f)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            acc[name].append(float(value))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fdir, wdir, cfg, out = sys.argv[1:5]
    fetch, nf = per_kernel(fdir, "FETCH_SIZE")
    write, _ = per_kernel(wdir, "WRITE_SIZE")
    stages = collections.defaultdict(float)
    kernels = {}
    for k in fetch:
        b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
        for rx, st in STAGE:
            if re.search(rx, k):
                # kernels launched more than once per step under one name (the two plain NT GEMMs) are averaged per launch
                stages[st] += b
                short = re.sub(r"\(anonymous namespace\)::", "", k).replace("void aecf::", "").replace("aecf::", "")
                kernels[re.sub(r"\(.*", "", short)[:80]] = dict(stage=st, bytes_per_launch=b, fetch_kb=fetch[k],
                                                           write_kb=write.get(k, 0.0), launches=nf[k])
                break
    if "plain_nt" in stages:
        stages["fwd.outproj"] = stages["bwd.dout"] = stages.pop("plain_nt")
    total = sum(v for k, v in stages.items())
    json.dump({"config": cfg,
               "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB*1024, rocprofv3 --pmc, separate passes; "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide streaming reads)",
               "bytes_per_launch": dict(stages), "kernels": kernels, "total_bytes_per_step": total}, open(out, "w"), indent=1)
    print(json.dumps(dict(stages), indent=1))
    print("total GB/step", total / 1e9)


if __name__ == "__main__":
    main()
