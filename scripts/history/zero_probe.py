import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go
for fields in (dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)):
    N = fields["Resolut"]
    cfg = go.MelConfig(GriffinLimIterations=1, **fields)
    m = gomel_amd.NewMel()
    for k, v in dict(fields, GriffinLimIterations=1).items(): setattr(m, k, v)
    mel = go.random_logmel(1, cfg.NumMels, seed=2)
    x0 = np.zeros(N)
    want = go.from_mel(cfg, mel.copy(), x0); got = m.FromMel(mel.copy(), x0=x0)
    err = got - want
    w = go.hann(N); w[0] = w[-1] = 1.0
    E = np.fft.rfft(err / w); W = np.fft.rfft(want / w)
    top = np.argsort(np.abs(E))[-6:][::-1]
    print(N, "rms err %.3e rms want %.3e" % (np.sqrt((err**2).mean()), np.sqrt((want**2).mean())), "worst bins", [(int(k), "%.2e" % abs(E[k]), "%.2e" % abs(W[k])) for k in top])
    M = go.target_magnitudes(cfg, np.exp(mel))[0]; Mg = m.engine().target_magnitudes(mel)[0] * N
    print("   mag max rel diff %.2e" % np.max(np.abs(Mg - M) / (M + 1e-30)), "M[top]", [("%.3e" % M[k], "%.3e" % Mg[k]) for k in top[:3]])
