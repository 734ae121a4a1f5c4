cd navlab-dpe-sdr_amd
cp libdpe_hip.so /tmp/orig.so
for v in p4u2 p8u2 p4u4 p8u4 p2u4; do
  cp libdpe_hip_$v.so libdpe_hip.so
  cd ..; echo -n "$v "; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['kernels_ms_per_step']['bcm_scan_pos'], d['kernels_ms_per_step']['bcm_scan_vel'])"; cd navlab-dpe-sdr_amd
done
cp /tmp/orig.so libdpe_hip.so
