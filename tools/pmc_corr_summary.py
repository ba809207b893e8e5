"""gpurun_out/pmc_corr/{fetch,write}.csv (tools/pmc_corr.sh) -> per-launch HBM traffic of the correlation lookups (JSON).
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is calibrated twice: on the 1 GiB streaming copy (16 B per lane) and on the
uniform-flow launch of the displacement-major kernel (2 B per lane, every line read once: the analytic byte count is
printed by tools/pmc_corr.py)."""
import csv
import json
import re
import sys

d = sys.argv[1]
rows = {k: sorted(csv.DictReader(open(f"{d}/{k}.csv")), key=lambda r: int(r["Dispatch_Id"])) for k in ("fetch", "write")}
uni_bytes = int(re.search(r"uniform_flow_lines \d+ bytes (\d+)", open(f"{d}/fetch.log").read()).group(1))


def groups(rs):
    copy = [r for r in rs if "copyBuffer" in r["Kernel_Name"]][-1]
    dm_enc = [r for r in rs if "corr_dm_encode_kernel" in r["Kernel_Name"]]
    dm_cl = [r for r in rs if "corr_dm_lookup_kernel" in r["Kernel_Name"]]
    tiled = [r for r in rs if "corr_lookup_r3_tiled" in r["Kernel_Name"]]
    v = lambda r: float(r["Counter_Value"])
    m = lambda q: sum(map(v, q)) / len(q)
    return {"copy": v(copy), "dm_uniform_enc": m(dm_enc[:3]), "dm_enc": m(dm_enc[3:6]), "dm_cl": m(dm_cl[-3:]), "tiled_cl": m(tiled[-3:])}


f, w = groups(rows["fetch"]), groups(rows["write"])
GiB_KiB = float(1 << 20)
corr_stream = GiB_KiB / f["copy"]
corr_pattern = (uni_bytes / 1024.0) / f["dm_uniform_enc"]
wc = GiB_KiB / w["copy"]
alg_read = 36 * 4800 * 4 * (64 * 2 + 8)
out = {"calibration": {"copy_FETCH_KiB": f["copy"], "copy_WRITE_KiB": w["copy"], "fetch_correction_streaming": corr_stream,
                       "uniform_flow_bytes_analytic": uni_bytes, "uniform_flow_FETCH_KiB": f["dm_uniform_enc"],
                       "fetch_correction_this_pattern": corr_pattern, "write_correction": wc},
       "algorithmic_read_bytes": alg_read}
for k in ("dm_uniform_enc", "dm_enc", "dm_cl", "tiled_cl"):
    rd = f[k] * 1024 * corr_stream
    out[k] = {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w[k], "hbm_read_bytes": rd, "hbm_read_bytes_pattern_cal": f[k] * 1024 * corr_pattern,
              "hbm_write_bytes": w[k] * 1024 * wc, "read_over_algorithmic": rd / alg_read}
json.dump(out, sys.stdout, indent=1)
