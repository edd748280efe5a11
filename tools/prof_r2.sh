#!/bin/bash
# Round-2 kernel-time profiles (rocprofv3 --kernel-trace --stats, no counters): the bench command (fp32 cfg2 step + SK
# + the 16-bit cfg5 leg) and the 16-bit step at cfg2's shape.  Summaries -> gpurun_out/$1/ (copy to profiles/).
out=gpurun_out/${1:-prof}
mkdir -p $out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cfg5-steps 3 > $out/bench_under_rocprof.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/step16 -o p -- python tools/step16_bench.py 16 16 10 bf16 > $out/step16.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for tag in ("bench", "step16"):
    fs = glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True)
    if not fs:
        print("no stats for", tag); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"{out}/{tag}_kernel_summary.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats, all kernels ({len(rows)}), total {tot/1e6:.2f} ms\n")
        f.write(f"{'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}  name\n")
        for r in rows:
            f.write(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                    f"{100*float(r['TotalDurationNs'])/tot:6.2f}  {r['Name']}\n")
PY
