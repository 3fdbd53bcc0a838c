"""Soak: several host threads drive two handles of one device for a while -- jobs in flight (whole-batch launches), single utterances (the
persistent kernels, one and two launches), mid-size batches (generic pair), ToMel -- and every result is compared bit for bit with what the
same call returned when it ran alone.  Exercises the device gate, the pin registry, the buffer pool and the job queue together.
usage: soak.py [seconds]   (exit code 0: every comparison held, no call failed)"""
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=12)
S = dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024, GriffinLimIterations=12)


def mel(fields):
    m = gomel_amd.NewMel()
    for k, v in fields.items():
        setattr(m, k, v)
    return m


def logmels(frames, K, first):
    return [np.random.default_rng(first + i).uniform(math.log(1e-5), math.log(50.0), size=(f * K, 2)) for i, f in enumerate(frames)]


def main():
    mt, ms = mel(T), mel(S)
    et, es = mt.engine(), ms.engine()
    work = {
        "jobs_T": (et, logmels([200] * 24, 192, 0)),            # fused kernels, submitted
        "single_T": (et, logmels([342], 192, 100)),             # persistent: one fp64 launch + one fp32
        "pair_T": (et, logmels([342, 300], 192, 200)),          # persistent: two fp64 launches + one fp32
        "mid_T": (et, logmels([342] * 4, 192, 300)),            # generic pair
        "small_S": (es, logmels([100] * 8, 80, 400)),           # persistent
        "mid_S": (es, logmels([400] * 8, 80, 500)),             # two fp64 persistent launches
    }
    wave = np.random.default_rng(9).standard_normal(44100 * 3) * 0.1
    want = {k: e.from_mel_batch([m.copy() for m in ms_], None, list(range(len(ms_)))) for k, (e, ms_) in work.items()}
    want_job = [o.copy() for o in et.submit_f32([m.astype(np.float32) for m in work["jobs_T"][1]], None, list(range(24))).wait()]
    want_mel = et.to_mel(wave)
    stop = time.time() + SECONDS
    bad, calls = [], {}
    lock = threading.Lock()

    def note(name, ok):
        with lock:
            calls[name] = calls.get(name, 0) + 1
            if not ok:
                bad.append(name)

    def sync_worker(name):
        e, ms_ = work[name]
        while time.time() < stop and not bad:
            got = e.from_mel_batch([m.copy() for m in ms_], None, list(range(len(ms_))))
            note(name, all(np.array_equal(a, b) for a, b in zip(got, want[name])))

    def job_worker():
        ms32 = [m.astype(np.float32) for m in work["jobs_T"][1]]
        while time.time() < stop and not bad:
            j1 = et.submit_f32(ms32, None, list(range(24)))
            j2 = et.submit_f32(ms32, None, list(range(24)))
            for j in (j2, j1):
                got = j.wait()
                note("jobs_f32", all(np.array_equal(a, b) for a, b in zip(got, want_job)))

    def tomel_worker():
        while time.time() < stop and not bad:
            note("to_mel", np.array_equal(et.to_mel(wave), want_mel))

    def guarded(fn, *a):
        def run():
            try:
                fn(*a)
            except Exception as ex:      # a failed call ends the soak as a mismatch does
                with lock:
                    bad.append("%s: %r" % (fn.__name__, ex))
        return run

    threads = [threading.Thread(target=guarded(sync_worker, k)) for k in ("single_T", "pair_T", "mid_T", "small_S", "mid_S")]
    threads += [threading.Thread(target=guarded(job_worker)), threading.Thread(target=guarded(tomel_worker))]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print("soak %.0f s: calls %s; mismatches or failures: %s" % (time.time() - t0, calls, bad or "none"), flush=True)
    et.close(); es.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
