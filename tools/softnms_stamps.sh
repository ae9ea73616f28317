# Development: where an outer step of the register-resident Soft-NMS kernel spends its cycles.  Builds a second library with
# -DRR_SNMS_STAMP (s_memtime stamps of wave 0 accumulated per phase), runs single segments of N boxes through it.
set -e
cd "$(dirname "$0")/.."
OBJS=$(ls rrnet_amd/csrc/_build/*.o | grep -v softnms.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I rrnet_amd/csrc -Wno-unused-result -fno-gpu-rdc -ffp-contract=off -DRR_SNMS_STAMP -c rrnet_amd/csrc/softnms.hip -o /tmp/softnms_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/librrnet_hip_stamp.so $OBJS /tmp/softnms_stamp.o
RRNET_HIP_LIB=$PWD/gpurun_out/librrnet_hip_stamp.so python3 - <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from rrnet_amd import _C
from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
import bench_softnms
lib = _C.lib()
lib.rr_snms_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(219)
names = ["pass 1: swap + overlap filter", "decay loop (fp64 union / div / exp)", "pass 2: lane's best", "block_best (DPP + LDS exchange + barrier)", "-", "-", "renumbering (death steps) + loop"]
for n in (150, 1500, 9000):
    one = bench_softnms.gen_boxes(n, rng)
    base = torch.from_numpy(one).cuda()
    seg_off = torch.tensor([0, n], dtype=torch.int32, device="cuda")
    for rep in range(2):
        work = base.clone()
        lib.rr_snms_stamps(None, 1)
        n_out, _ = soft_nms_segments(work, seg_off, n, 0.5, 0.7, 0.1, 2, check=False)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.rr_snms_stamps(buf, 0)
    steps = int(n_out[0])
    tot = sum(buf[i] for i in range(7))
    print("N = %d: %d outer steps, %d cycles per step (wave 0)" % (n, steps, tot // max(steps, 1)))
    for i in (0, 1, 2, 3, 6):
        print("    %-44s %7d cycles per step  (%4.1f %%)" % (names[i], buf[i] // max(steps, 1), 100.0 * buf[i] / max(tot, 1)))
PY
