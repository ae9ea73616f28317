# loss-curve A/B of the bf16-only activations (RR_BF16_ONLY_ACT, default on under cfg.Model.bf16) over N steps of the
# config-4 plain model at full size, same seed, same batches; the fp32 model as the yardstick of what "the same" means.
#   bash tools/ab_only_act.sh r06 300     -> profiles/<tag>_only_act_ab.json (summary) ; full curves under gpurun_out/
TAG=${1:-r06}; N=${2:-300}
mkdir -p gpurun_out
RR_BF16_ONLY_ACT=1 timeout 900 python3 tools/bench_config4.py --plain --bf16 --steps $N --curve > gpurun_out/curve_${TAG}_only1.json 2> gpurun_out/curve_${TAG}_only1.err
RR_BF16_ONLY_ACT=0 timeout 900 python3 tools/bench_config4.py --plain --bf16 --steps $N --curve > gpurun_out/curve_${TAG}_only0.json 2> gpurun_out/curve_${TAG}_only0.err
timeout 900 python3 tools/bench_config4.py --plain --fp32 --steps $((N/2)) --curve > gpurun_out/curve_${TAG}_fp32.json 2> gpurun_out/curve_${TAG}_fp32.err
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
def load(n):
    try:
        return json.loads(open("gpurun_out/curve_%s_%s.json" % (tag, n)).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
arms = {n: load(n) for n in ("only1", "only0", "fp32")}
out = {"_commit": (open("profiles/.commit").read().strip() if __import__("os").path.exists("profiles/.commit") else "unrecorded"),
       "what": "total loss / heat-map focal loss of the config-4 plain train step (hourglass-104, B=8, 1024x1024, seed 219, the loader's resident "
               "batches in order), mean over windows of 25 steps; only1 = bf16-only activations (default), only0 = every activation keeps "
               "its fp32 tensor (RR_BF16_ONLY_ACT=0), fp32 = the fp32 model"}
for n, r in arms.items():
    c = r.get("loss_curve")
    if not c:
        out[n] = r
        continue
    w = 25
    out[n] = {"steps": len(c), "finite": r.get("finite_after_timed_steps"),
              "total_by_window": [round(sum(x[0] for x in c[i:i + w]) / len(c[i:i + w]), 4) for i in range(0, len(c), w)],
              "hm_by_window": [round(sum(x[1] for x in c[i:i + w]) / len(c[i:i + w]), 4) for i in range(0, len(c), w)]}
json.dump(out, open("profiles/%s_only_act_ab.json" % tag, "w"), indent=1)
import os
os.makedirs("gpurun_out/profiles_%s" % tag, exist_ok=True)
json.dump(out, open("gpurun_out/profiles_%s/%s_only_act_ab.json" % (tag, tag), "w"), indent=1)
print(json.dumps(out)[:3000])
PY
