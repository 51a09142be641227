"""HBM bytes per launch of every Llama-step kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC slots),
summarised by tools/pmc_summary.py:  python tools/pmc_hbm_json.py fetch.csv write.csv out.csv out_gate_up.json [round label]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
read requests of wide coalesced streams at 64 bytes (MI355X_MICROARCH.md, HBM section), hence the factor 2 for these streaming kernels."""
import csv
import json
import sys


def load(path, counter):
    out = {}
    for row in csv.DictReader(open(path)):
        if row["counter"] == counter:
            out[row["kernel"]] = (int(row["dispatches"]), float(row["avg_value"]))
    return out


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k, (n, f) in fetch.items():
    w = write.get(k, (0, 0.0))[1]
    rows.append((k, n, f, w, (2 * f + w) * 1024))
rows.sort(key=lambda r: -r[4] * r[1])
with open(sys.argv[3], "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)*1024\n")
    for k, n, f, w, b in rows:
        fh.write(f"\"{k}\",{n},{f:.1f},{w:.1f},{b:.0f}\n")
gu = [r for r in rows if "gemv16_kernel<16, 8, 1, 3" in r[0]]
if gu:
    k, n, f, w, b = gu[0]
    json.dump({"kernel": k, "launches": n, "round": (sys.argv[5] if len(sys.argv) > 5 else "round ?"), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": b,
               "algorithmic_bytes_per_launch": 235286528,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --output-format csv) over "
                         "tools/probe_llm.py --frames 24; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at "
                         "64 B for wide coalesced streams); WRITE_SIZE uncalibrated (<0.1 % of the total)"}, open(sys.argv[4], "w"), indent=1)
    print(f"gate/up: {b / 1e6:.2f} MB per launch over {n} launches (algorithmic 235.29 MB)")
