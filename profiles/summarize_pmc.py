"""Summarise the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs of
profiles/pmc_gathers.py) into per-launch HBM traffic of the gather kernels.

Corrections (MI355X_MICROARCH.md, section HBM): counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, i.e. reports 1/2 of the bytes of wide reads -- calibrated here on
the 1 GiB device copy contained in the same run (expected 1 GiB read, 1 GiB written).

    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_pmc_kernels.json
"""
import csv
import json
import sys


def load(d):
    return list(csv.DictReader(open(f"{d}/g_counter_collection.csv")))


def mean_of(rows, pred):
    v = [float(r["Counter_Value"]) for r in rows if pred(r["Kernel_Name"])]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main(fetch_dir, write_dir):
    f, w = load(fetch_dir), load(write_dir)
    GiB_KiB = float(1 << 20)
    # calibration = the LAST device copy of the run (the 1 GiB clone at the end of pmc_gathers.py); other
    # copyBuffer launches (e.g. re-tiling the correlation pyramid) are larger
    last = lambda rows: [r for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])) if "copyBuffer" in r["Kernel_Name"]][-1]
    cal_f = float(last(f)["Counter_Value"])
    cal_w = float(last(w)["Counter_Value"])
    fetch_corr = GiB_KiB / cal_f     # ~2.0 on gfx950
    write_corr = GiB_KiB / cal_w     # ~1.0
    out = {"calibration": {"copy_bytes": 1 << 30, "FETCH_SIZE_KiB": cal_f, "WRITE_SIZE_KiB": cal_w,
                           "fetch_correction": fetch_corr, "write_correction": write_corr}}
    for key, pred in (("conv_igemm_gru_zr", lambda n: "conv_igemm_kernel<1" in n or "conv_igemm_kernelILi1" in n or "conv_pp_kernel<1" in n
                       or "conv_pp_kernelILi1" in n),
                      ("corr_lookup", lambda n: "corr_lookup" in n), ("knn_query", lambda n: "knn_query" in n),
                      ("idw_gather", lambda n: "idw_gather" in n), ("mlp_geo", lambda n: "mlp_geo" in n),
                      ("mlp_nb", lambda n: "mlp_nb" in n), ("mlp_col", lambda n: "mlp_col" in n),
                      ("corr_dm_lookup", lambda n: "corr_dm_lookup" in n)):
        fv, nf = mean_of(f, pred)
        wv, nw = mean_of(w, pred)
        if fv is None or wv is None:
            continue
        out[key] = {"launches": nf, "FETCH_SIZE_KiB": fv, "WRITE_SIZE_KiB": wv,
                    "hbm_read_bytes": fv * 1024 * fetch_corr, "hbm_write_bytes": wv * 1024 * write_corr,
                    "hbm_bytes": fv * 1024 * fetch_corr + wv * 1024 * write_corr}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
