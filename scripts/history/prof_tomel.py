"""Profiling target for the forward path and package phase: a batch of 10 s utterances through gomel_to_mel_batch, one through
ToPhase / FromPhase.  Run under `rocprofv3 --kernel-trace --stats`; prints the algorithmic bytes per launch for the roofline lines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut = 192, 16000.0, 1280, 4096
rng = np.random.default_rng(0)
sigs = [rng.standard_normal(441000) * 0.1 for _ in range(U)]
m.ToMelBatch(sigs)
t = time.perf_counter(); out = m.ToMelBatch(sigs); dt = time.perf_counter() - t
F = sum(len(o) for o in out) // 192
print("to_mel_batch: %d utterances, %d frames, %.2f ms host-to-host (%.2f M frames/s)" % (U, F, dt * 1e3, F / dt / 1e6))
print("k_stft_mag algorithmic bytes per launch: %d (hop*8 in + (N/2+1)*8 out per frame)" % (F * (1280 * 8 + 2049 * 8)))
print("k_domel_log algorithmic bytes per launch: %d ((N/2+1)*8 in + K*16 out per frame)" % (F * (2049 * 8 + 192 * 16)))
p = gomel_amd.NewPhase()
p.Window, p.Resolut, p.NumFreqs = 1280, 4096, 2048
ph = p.ToPhase(sigs[0]); w = p.FromPhase(ph)
ph = p.ToPhase(sigs[0]); w = p.FromPhase(ph)
print("phase: %d pairs -> %d samples" % (len(ph), len(w)))
