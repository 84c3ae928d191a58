# sw_all_gather_kernel: fabric reads / writes per launch for the block orders of tools/r04_swall_ab.py (separate counter passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWALL_ORDERS=${SWALL_ORDERS:-0,1,2,8}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/swa_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/swa_$c -- python3 tools/r04_swall_ab.py > gpurun_out/swa_$c.log 2>&1
  python3 - $c <<'PY'
import csv, glob, os, sys
c = sys.argv[1]
orders = os.environ["SWALL_ORDERS"].split(",")
f = glob.glob("gpurun_out/swa_%s/*/*counter_collection.csv" % c)[0]
rows = [r for r in csv.DictReader(open(f)) if "sw_all_gather_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
vals = {}
for r in rows:
    vals.setdefault(r["Dispatch_Id"], 0.0)
    vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = [vals[k] for k in sorted(vals, key=int)]
n = len(v) // len(orders)       # 11 launches per order
for j, o in enumerate(orders):
    seg = v[j * n:(j + 1) * n]
    kb = sum(seg) / len(seg)
    print("%s swall_order %s: %.1f MB per launch (x1024%s)" % (c, o, kb * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, " x2, the gfx950 wide-read correction" if c == "FETCH_SIZE" else ""))
PY
done
