# The forced one-rank RCCL step (RR_DP_FORCE=1: every SyncBN exchange and gradient bucket issued for real on a one-rank nccl group)
# beside the plain step, back to back in one call, + a kernel trace of the forced step split into device idle around the exchanges.
#   bash tools/prof_dp_force.sh r06   -> gpurun_out/profiles_r06/r06_bench_dp_force.json, r06_bench_dp_plain.json, r06_dp_force_gaps.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r06}; OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-host-fed"
for rep in 1 2; do
  timeout 600 python3 bench.py $ARGS > gpurun_out/dp_plain_$rep.log 2>&1; grep '^{"metric"' gpurun_out/dp_plain_$rep.log | tail -1 > $OUT/${TAG}_bench_dp_plain_$rep.json
  RR_DP_FORCE=1 timeout 600 python3 bench.py $ARGS > gpurun_out/dp_force_$rep.log 2>&1; grep '^{"metric"' gpurun_out/dp_force_$rep.log | tail -1 > $OUT/${TAG}_bench_dp_force_$rep.json
done
RR_DP_FORCE=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}_dp -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-host-fed --no-kernel-timing > gpurun_out/prof_${TAG}_dp.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_${TAG}_dp > $OUT/${TAG}_dp_force_gaps.txt 2>&1
rm -rf gpurun_out/prof_${TAG}_dp
python3 - $OUT $TAG <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
for kind in ("plain", "force"):
    for rep in (1, 2):
        try:
            d = json.load(open("%s/%s_bench_dp_%s_%d.json" % (out, tag, kind, rep)))
            print(kind, rep, d["ms_per_step"], "ms  host enqueue", d.get("host_enqueue_ms_per_step"), d.get("dp_host_ms_per_step"), d.get("collectives_per_step"))
        except Exception as e:
            print(kind, rep, "failed", e)
PY
head -30 $OUT/${TAG}_dp_force_gaps.txt
