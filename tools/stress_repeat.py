"""Stress: the same evaluations over and over on several sizes, several handles alive at once, single and batched entry points
interleaved -- every repeat must return the first result's bits (races in the cross-stream edges, the operand-order side
buffers or the in-kernel polls show up as differences).   python tools/stress_repeat.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # > 1: the handles are spread over this many host threads that run at once
sizes = [(1000, 3, "RBF"), (1536, 4, "Matern52"), (2500, 5, "RBF"), (3100, 4, "Matern32"), (4200, 6, "RBF"), (5200, 5, "Matern52"), (8320, 8, "RBF")]
gps, ref = [], []
for (N, d, k) in sizes:
    X, y = synth_problem(N, d, seed=N)
    gp = MiGP(X, y, k)
    th = np.array(theta_sequence(d, 3, seed=N))
    gps.append((gp, th))
    ref.append(([gp.lml(t) for t in th], [gp.lml_grad(t) for t in th], gp.lml_batch(th)))
t0 = time.time()
bad = 0
if threads > 1:
    import threading
    errs = []
    def work(mine):
        for r in range(rounds):
            for i in mine:
                gp, th = gps[i]
                l = [gp.lml(t) for t in th]
                g = [gp.lml_grad(t) for t in th]
                b = gp.lml_batch(th)
                if not (l == ref[i][0] and all(a[0] == c[0] and np.array_equal(a[1], c[1]) for a, c in zip(g, ref[i][1])) and np.array_equal(b, ref[i][2])):
                    errs.append((r, sizes[i]))
    ts = [threading.Thread(target=work, args=(list(range(k, len(gps), threads)),)) for k in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print(f"{threads} threads, {rounds} rounds: {time.time() - t0:.0f} s, mismatches {len(errs)} {errs[:5]}")
    print("stress_repeat", "ok" if not errs else "FAILED")
    sys.exit(1 if errs else 0)
for r in range(rounds):
    for i, (gp, th) in enumerate(gps):
        l = [gp.lml(t) for t in th]
        g = [gp.lml_grad(t) for t in th]
        b = gp.lml_batch(th)
        ok = l == ref[i][0] and all(a[0] == c[0] and np.array_equal(a[1], c[1]) for a, c in zip(g, ref[i][1])) and np.array_equal(b, ref[i][2]) \
            and np.array_equal(b, np.array(l))
        if not ok:
            bad += 1
            print("MISMATCH", r, sizes[i], l, ref[i][0], flush=True)
    if r % 10 == 9:
        print(f"round {r + 1}: {time.time() - t0:.0f} s, mismatches {bad}", flush=True)
print("stress_repeat", "ok" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
