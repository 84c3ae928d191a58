# sw_term: fabric reads / writes per launch for both block orders (separate counter passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/swt_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/swt_$c -- python3 tools/r04_swterm_ab.py > gpurun_out/swt_$c.log 2>&1
  python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob("gpurun_out/swt_%s/*/*counter_collection.csv" % c)[0]
rows = [r for r in csv.DictReader(open(f)) if "sw_term_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
# 4 x 21 launches: order 0, 1, 0, 1
vals = {}
for r in rows:
    vals.setdefault(r["Dispatch_Id"], 0.0)
    vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = [vals[k] for k in sorted(vals, key=int)]
n = len(v) // 4
for j in range(4):
    seg = v[j * n:(j + 1) * n]
    kb = sum(seg) / len(seg)
    print("%s swterm_order %d: %.1f MB per launch (x1024%s)" % (c, j & 1, kb * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, " x2, the gfx950 wide-read correction" if c == "FETCH_SIZE" else ""))
PY
done
