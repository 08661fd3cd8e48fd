"""Runs the LOAM pipeline and the ICP batch under different GPSCAL_POISON fill bytes in child processes: any
difference means some kernel reads device memory nobody wrote."""
import os, pickle, subprocess, sys
import numpy as np
sys.path.insert(0, "/root/repo")
tmp = "/tmp/poisonp"
os.makedirs(tmp, exist_ok=True)
if len(sys.argv) > 1:
    from gpscalibration_amd import Context, synth
    ctx = Context(0)
    W = synth.lidar_world(0, length=600.0)
    bag, st, truth = synth.drive(W, 60, seed=1, n_az=900)
    a = ctx.input_data_run([bag], [st], 50.0, 22.0, 8.0)
    segs = [bag[:30], bag[30:60]]
    b = ctx.loam_run(segs, [st[:30], st[30:60]])
    tg, to, sr, so, _ = synth.scan_batch(3, 20000)
    sb = ctx.scan_batch(tg, to, sr, so)
    T, err, _ = sb.icp(12)
    idx, sqd = sb.correspondences()
    ix = ctx.knn_index(tg[:20000])
    ki, kd = ix.search(sr[:5000] + np.float32(0.4), 5)
    sreg = ctx.scan_registration(bag[:4])
    pickle.dump({"id": a, "loam": b, "T": T, "err": err, "idx": idx, "sqd": sqd, "ki": ki, "kd": kd, "sr": sreg},
                open(tmp + "/%s.pkl" % sys.argv[1], "wb"))
    sys.exit(0)


def same(x, y):
    if isinstance(x, dict):
        return x.keys() == y.keys() and all(same(x[k], y[k]) for k in x)
    if isinstance(x, (list, tuple)):
        return len(x) == len(y) and all(same(a, b) for a, b in zip(x, y))
    if isinstance(x, np.ndarray):
        return x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
    return x == y


tags = {"none": None, "ab": "0xAB", "p41": "0x41", "p7f": "0x7f", "pff": "0xff"}
if os.environ.get("PROBE_TAGS"):
    tags = {t: v for t, v in tags.items() if t == "none" or t in os.environ["PROBE_TAGS"].split(",")}
for t in tags:
    if os.path.exists(tmp + "/%s.pkl" % t):
        os.remove(tmp + "/%s.pkl" % t)
for t, v in tags.items():
    env = dict(os.environ)
    env.pop("GPSCAL_POISON", None)
    if v:
        env["GPSCAL_POISON"] = v
    r = subprocess.run([sys.executable, __file__, t], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        print(t, "FAILED", r.stdout.decode()[-800:])
R = {t: pickle.load(open(tmp + "/%s.pkl" % t, "rb")) for t in tags if os.path.exists(tmp + "/%s.pkl" % t)}
bad = False
for t in R:
    for k in R["none"]:
        ok = same(R[t][k], R["none"][k])
        print(t, k, "same as unpoisoned:", ok, flush=True)
        bad = bad or not ok
sys.exit(1 if bad or len(R) != len(tags) else 0)
