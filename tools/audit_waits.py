"""Which kernels serialise their loads?  Compiles every csrc/*.hip to gfx950 assembly and counts, per kernel, the global / buffer loads
whose NEXT vm-counter wait is `s_waitcnt vmcnt(0)` with no other load issued in between ("lonely" loads: one round trip each).
Round 3 found three kernels that way whose source looked pipelined (run-time flags or lambdas around the loads made the wait-count
pass conservative, or the compiler sank a load under a select): the crop + resize, the window sums, the finalize kernel.
python tools/audit_waits.py [min_loads]"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="eqa_audit_")
srcs = sorted(glob.glob(os.path.join(ROOT, "equiadapt_amd", "csrc", "*.hip")))


def asm(src):
    out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-I",
                    os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
    return out


with ThreadPoolExecutor(6) as pool:
    files = list(pool.map(asm, srcs))
min_loads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for path in files:
    name, loads, lonely, pending, rows = None, 0, 0, 0, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name and loads >= min_loads:
                rows.append((name, loads, lonely))
            name, loads, lonely, pending = m.group(1), 0, 0, 0
            continue
        t = line.strip()
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.startswith(("global_load", "buffer_load")) and "lds" not in t:
            loads += 1
            pending += 1
        elif op == "s_waitcnt" and "vmcnt(0)" in t:
            lonely += pending == 1
            pending = 0
        elif op == "s_waitcnt" and "vmcnt" in t:
            pending = 0 if pending <= 1 else pending
        elif op == "s_endpgm":
            if name and loads >= min_loads:
                rows.append((name, loads, lonely))
            name = None
    for n, lo, ly in rows:
        if 2 * ly >= lo:
            try:
                n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
            except OSError:
                pass
            print(f"{os.path.basename(path):16s} {lo:4d} loads, {ly:4d} lonely   {n[:110]}")
