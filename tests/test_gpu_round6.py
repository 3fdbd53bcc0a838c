"""GPU tests added in round 6: the bounded partner wait of the fp64 fused kernel (StreamFlagSync's poll cap, gl_kernels.hpp) and how the
host reports a give-up.  (The two-frames-in-flight form of that kernel was built, found bit-exact and 8-12 % slower, and dropped:
profiles/r06/two_frames_in_flight/.)"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
T = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096)     # the towav geometry: the one two-wave streams (N 4096) exist at


def mel_obj(fields, **kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(fields, **kw).items():
        setattr(m, k, v)
    return m


def inputs(U, F, K, first=0):
    return ([np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)],
            [1000 + first + u for u in range(U)])


def test_a_partner_wait_that_gives_up_is_reported_and_the_grid_drains(monkeypatch):
    """GOMEL_SYNC_CAP=0 (read at gomel_create): every wait of a two-wave stream that does not find its partner wave there at the first
    poll gives up -- what a wave does when its partner has stopped running.  The launch must still END (every later wait runs through
    after one poll), the call must come back as GOMEL_ERR_HIP with hipErrorLaunchTimeOut instead of a waveform, the handle must stay
    usable, and a handle created without the knob must be untouched by all of this."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(T, GriffinLimIterations=4)
    U, F, K = 24, 120, 192
    mels, seeds = inputs(U, F, K)
    good = mel_obj(fields).engine()
    good.set_precision(64)
    good.set_path(2)                                                   # the fused kernels (k_gl_fused_d at N 4096: two waves per stream)
    ref = good.from_mel_batch(mels, None, seeds)
    monkeypatch.setenv("GOMEL_SYNC_CAP", "0")
    bad = mel_obj(fields).engine()
    monkeypatch.delenv("GOMEL_SYNC_CAP")
    bad.set_precision(64)
    bad.set_path(2)
    statuses = []
    for entry in ("batch", "host_to_host", "job"):
        try:
            if entry == "batch":
                b = gomel_amd.Batch(bad, [F * K] * U)
                b.upload(mels, None, seeds)
                try:
                    b.run()
                    b.download()
                finally:
                    b.close()
            elif entry == "host_to_host":
                bad.from_mel_batch(mels, None, seeds)
            else:
                outs = [np.zeros(bad.out_len(F * K)) for _ in range(U)]
                bad.submit_bound(bad.bind_buffers([np.array(m) for m in mels], outs), seeds).wait()
            statuses.append((entry, 0, None))
        except gomel_amd.GomelError as ex:
            statuses.append((entry, ex.status, gomel_amd._lib.lib().gomel_last_hip_error()))
    # with 4 x 24 x 120 frame-iterations x 8 waits per frame some wait misses its first poll in every launch: all three entry points report
    assert all(st == 8 for _, st, _ in statuses), statuses
    assert all(hip == 702 for _, _, hip in statuses), statuses        # hipErrorLaunchTimeOut
    bad.set_precision(32)                                             # the handle itself is fine (the fp32 kernel's two-wave streams use s_barrier: no event waits to give up)
    got32 = bad.from_mel_batch(mels[:3], None, seeds[:3])
    assert all(np.isfinite(o).all() for o in got32)
    bad.close()
    again = good.from_mel_batch(mels, None, seeds)                     # the other handle never noticed
    assert all(np.array_equal(a, b) for a, b in zip(again, ref))
    good.close()
