import sys, os, subprocess, numpy as np, pickle
sys.path.insert(0, "/root/repo")
tmp = "/tmp/detp"
os.makedirs(tmp, exist_ok=True)
if len(sys.argv) > 1:
    from gpscalibration_amd import Context, synth
    W = synth.lidar_world(0, length=600.0)
    bag, st, truth = synth.drive(W, 150, seed=1, n_az=900)
    ctx = Context(0)
    a = ctx.input_data_run([bag], [st], 50.0, 22.0, 8.0)
    pickle.dump(a, open(tmp + "/%s.pkl" % sys.argv[1], "wb"))
    sys.exit(0)
for tag in ("t1", "d1", "d2", "e1", "e2", "e3"):
    env = dict(os.environ)
    if tag[0] in "abcd":
        env["GPSCAL_NO_TORCH"] = "1"
        env["GPSCAL_POOL_MASK"] = "8"
    if tag[0] == "e":
        env["GPSCAL_NO_TORCH"] = "1"
        env["GPSCAL_POOL_MASK"] = "8"
        env["GPSCAL_SORTED_PAD"] = "65536"
    subprocess.run([sys.executable, __file__, tag], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
R = {t: pickle.load(open(tmp + "/%s.pkl" % t, "rb")) for t in ("t1", "d1", "d2", "e1", "e2", "e3")}
ref = [(x["flag"], x["first"], x["last"], len(x["track"])) for x in R["t1"]]
for t in R:
    cuts = [(x["flag"], x["first"], x["last"], len(x["track"])) for x in R[t]]
    same = cuts == ref and all(np.array_equal(x["track"], y["track"]) for x, y in zip(R[t], R["t1"]))
    print(t, "identical to t1:", same)
a, b = R["t1"], R["n1"]
for k, (x, y) in enumerate(zip(a, b)):
    n = min(len(x["track"]), len(y["track"]))
    d = np.abs(x["track"][:n] - y["track"][:n]).max(axis=1)
    nz = np.flatnonzero(d > 0)
    print("track", k, "first differing row", nz[:1], "max diff %.3e" % (d.max() if n else 0))
