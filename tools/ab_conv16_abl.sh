for pm in 0 2; do for abl in 0 1 2 3; do
 echo "== PM=$pm ABL=$abl $(RR_CONV16_PM=$pm RR_CONV16_ABL=$abl timeout 300 python3 tools/bench_conv16.py --reps 20 --shape 8,256,256,256,256,3,1 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fprop_nostats_ms'], d['fprop_nostats_tflops'], d['dgrad_ms'])")"
done; done
