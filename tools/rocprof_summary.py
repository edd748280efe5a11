"""Summarise a rocprofv3 run (rocpd sqlite .db or *kernel_trace.csv): per (kernel, grid) count /
total / average duration.  Usage: python tools/rocprof_summary.py <db-or-dir> [top]"""
import collections, csv, glob, os, re, sqlite3, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for f in dbs:
    cur = sqlite3.connect(f).cursor()
    rows += list(cur.execute("select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), duration, vgpr_count, accum_vgpr_count, lds_size from kernels"))
if not dbs:
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], r.get("Grid_Size", "?"), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), 0, 0, 0))
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0, 0])
tot = 0.0
for name, grid, d, vg, ag, lds in rows:
    name = re.sub(r"\(.*", "", name).replace("void slv::", "").replace("slv::", "")
    a = agg[(name, grid)]
    a[0] += 1; a[1] += d / 1e3; a[2:] = [vg, ag, lds]; tot += d / 1e3
# the library the trace was taken on: selavi_amd/build.py's digest of csrc/ + include/ + flags (bench.py reports a committed
# summary's in-step figures only when it matches the library it runs on)
try:
    stamp = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "selavi_amd", "libselavi_hip.so.stamp")).read().strip()
except OSError:
    stamp = "unknown"
print(f"library digest {stamp}")
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} launches")
print(f"{'us_total':>12} {'%':>6} {'count':>6} {'us_avg':>10} {'vgpr':>5} {'agpr':>5} {'lds':>6}  kernel [workgroups]")
for (name, grid), (c, d, vg, ag, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{d:12.1f} {100*d/tot:6.2f} {c:6d} {d/c:10.1f} {vg:5d} {ag:5d} {lds:6d}  {name[:80]} [{grid}]")
# per-kernel-name roll-up
byname = collections.defaultdict(lambda: [0, 0.0])
for (name, grid), (c, d, *_ ) in agg.items():
    byname[name][0] += c; byname[name][1] += d
print("\nby kernel name:")
for name, (c, d) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{d:12.1f} {100*d/tot:6.2f} {c:6d} {d/c:10.1f}  {name[:100]}")
