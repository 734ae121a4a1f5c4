"""Soak of the device-resident loop: 2 000 windows (40 distinct, cycled), fixes against the host-driven loop window by window."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import navlab_dpe_sdr_amd as dpe
fs, S, K, W0, REP = 2.5e6, 50000, 8, 40, 50
iq, _, _, _ = dpe.workload.build_windows(W0, fs, S, K, seed=71, amp=200.0)
iq = np.concatenate([iq] * REP)
W = iq.shape[0]
ho = dpe.workload.extend_handoff(dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV), K)
pos = dpe.synth.uniform_grid(9, 1.0); vel = dpe.synth.uniform_grid(9, 2.0)
tg = np.unique(pos[:, 3])
t0 = time.time(); fixes_h, res_h = dpe.pipeline.run_closed_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K); t1 = time.time()
fixes_d, res_d, status = dpe.pipeline.run_device_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K, ring_depth=8); t2 = time.time()
bad = sum(1 for w in range(W) if res_d[w]["posIndex"] != res_h[w]["posIndex"] or res_d[w]["velIndex"] != res_h[w]["velIndex"])
print("device-loop soak: %d windows, status %d, %d windows with a different grid point, max |fix difference| %.3g m (host-driven %.1f s, device loop %.1f s in Python)"
      % (W, status, bad, np.abs(fixes_d - fixes_h).max(), t1 - t0, t2 - t1))
assert status == 0 and bad == 0
# ... and with cuEKF's filter inside the measurement kernel: the two closed loops agree to the managers' own sin / cos (~1e-15 relative); the
# filter's arithmetic itself is compared bit for bit in tests/test_gpu_chm_dev.py
for couple in (True, False):
    fh, _ = dpe.pipeline.run_closed_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K, enable_ekf=True, couple_velocity=couple)
    fd, _, st = dpe.pipeline.run_device_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K, ring_depth=8, enable_ekf=True, couple_velocity=couple)
    rel = np.abs(fd - fh).max(0) / np.maximum(np.abs(fh).max(0), 1.0)
    print("device-loop soak with the filter (velocity coupling %s): %d windows, status %d, max relative state difference %.3g" % (couple, W, st, rel.max()))
    assert st == 0 and rel.max() < 1e-12
