import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go
fields = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096)
for F, I in ((2, 1), (7, 1), (7, 3)):
    cfg = go.MelConfig(GriffinLimIterations=I, **fields)
    m = gomel_amd.NewMel()
    for k, v in dict(fields, GriffinLimIterations=I).items(): setattr(m, k, v)
    mel = go.random_logmel(F, cfg.NumMels, seed=2)
    L = go.out_len(cfg, len(mel)); x0 = np.zeros(L)
    want = go.from_mel(cfg, mel.copy(), x0); got = m.FromMel(mel.copy(), x0=x0)
    err = got - want
    nb = L // 1280
    print("F", F, "I", I, "rms err %.3e want %.3e" % (np.sqrt((err**2).mean()), np.sqrt((want**2).mean())))
    print("   per hop-block rms err:", ["%.1e" % np.sqrt((err[i*1280:(i+1)*1280]**2).mean()) for i in range(nb)])
    print("   per hop-block rms want:", ["%.1e" % np.sqrt((want[i*1280:(i+1)*1280]**2).mean()) for i in range(nb)])
    i = int(np.argmax(np.abs(err))); print("   max err at", i, err[i], want[i], got[i])
