"""Times the persistent small-batch kernel against the per-iteration kernels on BASELINE configs[1] (one 342-frame utterance,
Resolut 4096, hop 1280, 60 iterations) and on small batches of the TTS geometry; checks bit equality on the way."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gomel_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import gomel_oracle as oracle


def go(N, hop, K, fmax, frames, I, path):
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, fmax, hop, N, I
    e = m.engine(); e.set_path(path)
    mels = [oracle.random_logmel(f, K, seed=i) for i, f in enumerate(frames)]
    b = gomel_amd.Batch(e, [x.shape[0] for x in mels])
    b.upload(mels, None, list(range(len(frames))))
    for _ in range(3): tm = b.run_timed()
    ts = []
    for _ in range(20):
        tm = b.run_timed(); ts.append(tm["total_ms"])
    out = b.download()
    t0 = time.perf_counter()
    for _ in range(20): o2 = e.from_mel_batch(mels, None, list(range(len(frames))))
    hh = (time.perf_counter() - t0) / 20 * 1e3
    b.close(); e.close()
    return out, float(np.median(ts)), tm, hh


for name, (N, hop, K, fmax, frames, I) in {"cfg2 single 10 s utterance": (4096, 1280, 192, 16000.0, [342], 60),
                                           "cfg2 two utterances 200+150": (4096, 1280, 192, 16000.0, [200, 150], 60),
                                           "TTS 8 x 100 frames": (1024, 256, 80, 8000.0, [100] * 8, 60),
                                           "TTS 1 x 400 frames": (1024, 256, 80, 8000.0, [400], 60),
                                           "N 2048 hop 256 1 x 500": (2048, 256, 160, 8000.0, [500], 60), "N 2048 hop 128 1 x 500": (2048, 128, 160, 8000.0, [500], 60), "N 1024 hop 64 4 x 500": (1024, 64, 80, 8000.0, [500] * 4, 60)}.items():
    og, tg, tmg, hg = go(N, hop, K, fmax, frames, I, 1)
    op, tp, tmp, hp = go(N, hop, K, fmax, frames, I, 0)
    same = all(np.array_equal(a, b) for a, b in zip(og, op))
    print("%-30s generic %.3f ms (host-to-host %.3f)   auto[path %d] %.3f ms (host-to-host %.3f)  kernel %.3f ms  bit-identical %s" %
          (name, tg, hg, tmp["fast_path"], tp, hp, tmp["dom_ms"], same), flush=True)
