# the driver's N-rank command rehearsed on ONE GPU as real processes (the box admits six processes on its GPU):
#   5 ranks (with the torchrun agent the six processes the box admits): configs[3] as 40 x 32^3 (T_local 8), headline 32^4 per rank; 4 ranks: configs[3] T_local 16, strong_32 T_local 8
#   default (faces auto: host-staged ring first, then the direct carrier checked against it), HIP events instead of flags (4 ranks), direct carrier from the start
mkdir -p gpurun_out
export TMLQCD_BENCH_TRACE=1
run() {   # tag nranks env... -- args...
  tag=$1; n=$2; shift 2
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  t0=$(date +%s)
  env TMLQCD_BENCH_TRANSPORT=shm $envs timeout -k 10 560 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu --no-rows "$@" > gpurun_out/r04_bench_${tag}.json 2> gpurun_out/r04_bench_${tag}.err
  echo "== $tag: rc $? wall $(( $(date +%s) - t0 )) s"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04_bench_${tag}.json"))
    st = d.get("strong") or {}
    fd = d.get("faces_direct") or {}
    print("  value %.3e  n_ranks %s n_gpus %s rehearsal %s transport %s faces %s ring %s rccl %s" % (d["value"] or 0, d.get("n_ranks"), d.get("n_gpus"), d.get("rehearsal"), d.get("transport"), d.get("faces"), d.get("ring_nranks"), d.get("rccl_nranks")))
    print("  rank_check ok %s (%s)  strong %.3e  strong_32 check %s" % ((d.get("rank_check") or {}).get("ok"), (d.get("rank_check") or {}).get("lattice"), st.get("value") or 0, ((d.get("strong_32") or {}).get("rank_check") or {}).get("ok")))
    if fd:
        print("  faces_direct ok %s  strong check %s value %.3e  headline dev %s speedup %s" % (fd.get("ok"), ((fd.get("strong") or {}).get("rank_check") or {}).get("ok"), (fd.get("strong") or {}).get("value") or 0, (fd.get("headline") or {}).get("max_rel_dev_vs_communicator"), (fd.get("headline") or {}).get("speedup_vs_communicator")))
    print("  wall_s", {k: round(v, 1) for k, v in d["wall_s"].items()})
except Exception as e:
    print("  no line:", e)
PY
  grep -c "gave up" gpurun_out/r04_bench_${tag}.err | sed 's/^/  give-ups: /'
}
run 5ranks_default 5 --
run 4ranks_events 4 -- --opt split_sync=1
run 5ranks_ipc 5 TMLQCD_BENCH_TRANSPORT=ipc --
