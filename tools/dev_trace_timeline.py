"""Time line of ONE step from a rocprofv3 --kernel-trace CSV: per kernel its stream / queue, start and end relative to the step's
first kernel (us).  usage: python tools/dev_trace_timeline.py <run_kernel_trace.csv> [step index among the un-profiled steps]
A step starts at each k_planes* launch; steps whose launches carry profiling events run serialised and are listed like any other."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
want = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if nm.startswith("at::") or "rocclr" in nm:
        continue
    if nm.startswith("k_planes"):
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((nm, r["Queue_Id"], r["Stream_Id"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
st = steps[want]
t0 = st[0][3]
print(f"step {want} of {len(steps)}: {len(st)} launches, {(max(s[4] for s in st) - t0) / 1e3:.1f} us")
for nm, q, s, a, b, vg, lds, gx, wx in st:
    print(f"  q{q} s{s} {(a - t0) / 1e3:9.1f} .. {(b - t0) / 1e3:9.1f}  ({(b - a) / 1e3:7.1f} us)  {nm}  vgpr {vg} lds {lds} grid {int(gx)//int(wx)}x{wx}")
