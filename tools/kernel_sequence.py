"""Last N kernels of a rocprofv3 --kernel-trace run in launch order, with start offsets and durations: shows launch-bound
chains (many 1-2 us kernels spaced ~10 us apart).  python tools/kernel_sequence.py <rocprof output dir> [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -100:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
