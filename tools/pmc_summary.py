"""Turn rocprofv3 --pmc counter_collection CSVs into the per-kernel summary committed under profiles/.

    python tools/pmc_summary.py <tag> <out.json> <dir_with_FETCH_SIZE> <dir_with_WRITE_SIZE> [<dir_with_SQ counters>]

FETCH_SIZE / WRITE_SIZE are KiB per dispatch, collected in separate passes (TCC slots).  On gfx950 FETCH_SIZE
reports half of the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section): doubled here.
Calibration in this code base: adamw_kernel reads 16 B and writes 14 B per parameter (53.76 M parameters):
expected 860 MB / 753 MB, counters give 2 x 430 MB / 753 MB.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, by_grid=False):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = r["Kernel_Name"] + ("|grid=" + r["Grid_Size"] if by_grid else "")
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)


def row_of(cs):
    row = {"dispatches": max(n for _, n in cs.values())}
    if "FETCH_SIZE" in cs:
        row["fetch_bytes_avg"] = round(cs["FETCH_SIZE"][0] * 1024 * 2)
    if "WRITE_SIZE" in cs:
        row["write_bytes_avg"] = round(cs["WRITE_SIZE"][0] * 1024)
    if "fetch_bytes_avg" in row and "write_bytes_avg" in row:
        row["hbm_bytes_avg"] = row["fetch_bytes_avg"] + row["write_bytes_avg"]
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"):
        if c in cs:
            row[c] = round(cs[c][0])
    return row


def main():
    """extra args NAME=SUBSTRING:GRID name a bench kernel class: dispatches whose symbol contains SUBSTRING and
    whose total grid size (threads) is GRID, i.e. one problem shape."""
    args = sys.argv[1:]
    classes = [a for a in args if "=" in a and ":" in a.split("=", 1)[1] and not a.startswith("/")]
    args = [a for a in args if a not in classes]
    tag, out_path, dirs = args[0], args[1], args[2:]
    merged = collections.defaultdict(dict)
    for d in dirs:
        for k, cs in load(d).items():
            for c, v in cs.items():
                merged[short(k)][c] = (sum(v) / len(v), len(v))
    rows = {k: row_of(cs) for k, cs in merged.items()}
    cls_out = {}
    if classes:
        bygrid = collections.defaultdict(dict)
        for d in dirs:
            for k, cs in load(d, True).items():
                for c, v in cs.items():
                    bygrid[short(k.split("|grid=")[0]) + "|grid=" + k.split("|grid=")[1]][c] = (sum(v) / len(v), len(v))
        for spec in classes:
            name, rest = spec.split("=", 1)
            sub, grid = rest.rsplit(":", 1)
            acc = collections.defaultdict(lambda: [0.0, 0])
            for k, cs in bygrid.items():
                if sub in k and k.endswith("|grid=" + grid):
                    for c, (avg, n) in cs.items():
                        acc[c][0] += avg * n
                        acc[c][1] += n
            if acc:
                cls_out[name] = row_of({c: (t / n, n) for c, (t, n) in acc.items()})
                cls_out[name]["match"] = spec
    json.dump({"tag": tag, "classes": cls_out, "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024; averages per dispatch",
               "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1].get("hbm_bytes_avg", 0) * kv[1]["dispatches"]))},
              open(out_path, "w"), indent=1)
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main()
