"""profiles/traffic_latest.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

  python tools/make_traffic.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> > profiles/traffic_latest.json

Counter unit KB; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 — on gfx950 FETCH_SIZE counts half of wide
coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 "
                  "--warmup 1`; counter unit KB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half of wide "
                  "coalesced reads, MI355X_MICROARCH.md HBM section); tools/make_traffic.py"}
alias = {"k_bcd_w<0>": "k_bcd", "k_bcd_w<1>": "k_bcd_first", "k_bcd<0, 8>": "k_bcd_wg", "k_bcd<1, 8>": "k_bcd_wg_first"}
for k in sorted(fetch):
    if k.startswith("at::") or "elementwise" in k:
        continue
    f, w = fetch[k], write.get(k, [0.0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    name = alias.get(k, k.split("<")[0])
    out[name] = {"kernel": k, "launches_sampled": len(f), "fetch_size_kb": round(fm, 1), "write_size_kb": round(wm, 1),
                 "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
print(json.dumps(out, indent=1))
