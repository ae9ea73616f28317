# A/B of the conv16 forward kernel's main loop: ping-pong with PM of a sub-phase's four DMA pieces issued among the MFMAs
# (RR_CONV16_PM=0/1/2/4) against the round-5 loop (RR_CONV16_PP=0); stamps = in-kernel timeline of one workgroup (tools/conv16_stamps.py)
mkdir -p gpurun_out
for rep in 1 2; do
for v in "RR_CONV16_PP=0" "RR_CONV16_PM=0" "RR_CONV16_PM=1" "RR_CONV16_PM=2" "RR_CONV16_PM=4"; do
  for sh in 8,256,256,256,256,3,1 8,256,128,128,256,3,1 8,128,512,512,256,3,2; do
    echo "== $v $(env $v timeout 300 python3 tools/bench_conv16.py --reps 20 --shape $sh 2>&1 | tail -1)"
  done
done
done
for v in 0 1 2 4; do echo "== stamps PM=$v"; RR_CONV16_PM=$v timeout 300 python3 tools/conv16_stamps.py 2>&1 | grep -A7 "^wave 0\|^wave 4\|^total\|^stagger" | grep -v "^--"; done
