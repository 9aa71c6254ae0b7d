#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/micro/fetchcal.hip); writes gpurun_out/fetchcal.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fetchcal
mkdir -p $O $R/scripts/micro/bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/scripts/micro/fetchcal.hip -o $R/scripts/micro/bin/fetchcal || exit 1
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $c --output-format csv -d $O/$n -- $R/scripts/micro/bin/fetchcal > $O/$n.log 2>&1
done
python3 - $O <<'PY' | tee $R/gpurun_out/fetchcal.txt
import csv, glob, sys, collections
known = 12288 * 12288 * 8
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        key = k + (" rmw" if False else "")
        acc[(k, r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
# second repetition only (warm instruction caches); dispatch order: stream16, stream8, seg<2>, seg<1>, seg<2> rmw
names = ["stream16 (contiguous, 16 B/lane)", "stream8 (contiguous, 8 B/lane)", "seg1k (1 KiB row segments, 16 B/lane)",
         "seg512 (512 B row segments, 8 B/lane)", "seg1k_rmw (read v + read/write hv)"]
by_disp = collections.defaultdict(dict)
for (k, d), cs in acc.items():
    for c, v in cs.items():
        by_disp[int(d)][c] = sum(v) / len(v)
disp = sorted(by_disp)
print("# gfx950 rocprofv3 counter calibration; every kernel touches each byte of a %.3f GB matrix exactly once" % (known / 1e9))
print("# %-44s %14s %14s %10s %10s" % ("shape", "FETCH_SIZE B", "known read B", "ratio", "WRITE/known"))
for i, d in enumerate(disp[-5:]):
    c = by_disp[d]
    rd_known = known * (2 if i == 4 else 1)
    wr_known = known if i == 4 else 0
    fs = c.get("FETCH_SIZE", 0) * 1024
    ws = c.get("WRITE_SIZE", 0) * 1024
    print("%-46s %14.4g %14.4g %10.3f %10s   RDREQ=%.4g WRREQ=%.4g TCC_MISS=%.4g TCC_REQ=%.4g" % (
        names[i], fs, rd_known, fs / rd_known, ("%.3f" % (ws / wr_known)) if wr_known else "-",
        c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0), c.get("TCC_MISS_sum", 0), c.get("TCC_REQ_sum", 0)))
PY
