"""Synthetic inputs of the shapes BASELINE.json / SURVEY.md section 8(d) name.

No dataset ships with the reference besides data/original_gps_data.txt (the demo
bags are an external download), so benchmarks and parity tests use generated
data: lidar-like scan pairs (ground plane + walls + poles) and GPS/SLAM track
segments with GPRMC logs.  numpy only; seeded, deterministic for a numpy version.
"""
import math

import numpy as np


def rot_zyx(yaw_deg=0.0, pitch_deg=0.0, roll_deg=0.0):
    y, p, r = (math.radians(a) for a in (yaw_deg, pitch_deg, roll_deg))
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    return Rz @ Ry @ Rx


def scan_scene(n, seed, half=50.0, sigma=0.01):
    """n x 3 float32: 60 % ground (|x|,|y| < half), 30 % on 8 walls, 10 % on 32 poles."""
    rng = np.random.default_rng(seed)
    ng = int(round(0.6 * n))
    nw = int(round(0.3 * n))
    npole = n - ng - nw
    g = np.empty((ng, 3))
    g[:, 0:2] = rng.uniform(-half, half, size=(ng, 2))
    g[:, 2] = 0.0
    # 8 vertical walls, 20 m long x 5 m high, fixed layout from the seed-independent table
    wall_rng = np.random.default_rng(12345)
    wc = wall_rng.uniform(-0.8 * half, 0.8 * half, size=(8, 2))
    wa = wall_rng.uniform(0, math.pi, size=8)
    wi = rng.integers(0, 8, size=nw)
    s = rng.uniform(-10.0, 10.0, size=nw)
    w = np.empty((nw, 3))
    w[:, 0] = wc[wi, 0] + s * np.cos(wa[wi])
    w[:, 1] = wc[wi, 1] + s * np.sin(wa[wi])
    w[:, 2] = rng.uniform(0.0, 5.0, size=nw)
    pc = wall_rng.uniform(-0.9 * half, 0.9 * half, size=(32, 2))
    pi = rng.integers(0, 32, size=npole)
    ang = rng.uniform(0, 2 * math.pi, size=npole)
    p = np.empty((npole, 3))
    p[:, 0] = pc[pi, 0] + 0.1 * np.cos(ang)
    p[:, 1] = pc[pi, 1] + 0.1 * np.sin(ang)
    p[:, 2] = rng.uniform(0.0, 6.0, size=npole)
    pts = np.concatenate([g, w, p], axis=0)
    pts += rng.normal(0.0, sigma, size=pts.shape)
    return pts.astype(np.float32)


def scan_pair(n, pair_id=0, yaw_deg=None, pitch_deg=0.3, t=(0.50, 0.20, 0.05), sigma=0.01):
    """(target, source, T_true) with source = T_true^-1-ish: source = R*target + t + noise, shuffled.

    ICP recovers the transform that maps source back onto target, i.e. inverse(R, t).
    SURVEY 8d cfg 2: yaw 2 deg; cfg 4: pair p uses seeds (2p+1, 2p+2), yaw 0.5 + 0.003 p.
    """
    if yaw_deg is None:
        yaw_deg = 2.0 if pair_id == 0 else 0.5 + 0.003 * pair_id
    tgt = scan_scene(n, 2 * pair_id + 1, sigma=sigma)
    rng = np.random.default_rng(2 * pair_id + 2)
    R = rot_zyx(yaw_deg, pitch_deg, 0.0)
    src = tgt.astype(np.float64) @ R.T + np.asarray(t)
    src += rng.normal(0.0, sigma, size=src.shape)
    src = src[rng.permutation(len(src))].astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ np.asarray(t)
    return tgt, src, T


def scan_batch(npairs, n, first_pair=0):
    """Packed arrays for ScanBatch: (tgt[npairs*n,3], off, src[npairs*n,3], off, T_true[npairs,4,4])."""
    tg, sr, Ts = [], [], []
    for p in range(first_pair, first_pair + npairs):
        a, b, T = scan_pair(n, p)
        tg.append(a)
        sr.append(b)
        Ts.append(T)
    off = np.arange(npairs + 1, dtype=np.int64) * n
    return np.concatenate(tg), off, np.concatenate(sr), off.copy(), np.stack(Ts)


# ------------------------------------------------------------------ tracks
def _nmea_ddmm(v, is_lat):
    d = int(abs(v))
    m = (abs(v) - d) * 60.0
    return ("%02d%08.5f" if is_lat else "%03d%08.5f") % (d, m)


def gprmc_line(t, lat, lon, valid=True):
    """One line in the shipped log's format (data/original_gps_data.txt:1)."""
    tm = int(t) % 86400
    hms = "%02d%02d%02d.00" % (tm // 3600, (tm // 60) % 60, tm % 60)
    body = "$GPRMC,%s,%s,%s,%s,%s,%s,0.447,,130517,,,A" % (
        hms, "A" if valid else "V", _nmea_ddmm(lat, True), "N" if lat >= 0 else "S",
        _nmea_ddmm(lon, False), "E" if lon >= 0 else "W")
    cs = 0
    for ch in body[1:]:
        cs ^= ord(ch)
    return "%.8f,%s*%02X" % (t, body, cs)


def smooth_path(n, dt, seed, v_max=12.0):
    """Smooth random-curvature path with stops: returns (xy[n,2] metres, speed[n])."""
    rng = np.random.default_rng(seed)
    kappa = np.cumsum(rng.normal(0, 0.002, size=n))
    kappa -= np.linspace(0, kappa[-1], n)
    kappa = np.clip(kappa, -0.03, 0.03)
    v = np.clip(6.0 + np.cumsum(rng.normal(0, 0.15, size=n)), 0.0, v_max)
    stop = rng.random(n) < 0.002
    for i in np.flatnonzero(stop):
        v[i:i + int(3.0 / dt)] = 0.0
    heading = np.cumsum(kappa * v * dt) + rng.uniform(0, 2 * math.pi)
    xy = np.cumsum(np.c_[v * np.cos(heading), v * np.sin(heading)] * dt, axis=0)
    return xy, v


def track_segments(nseg, poses, seed=7, rate_hz=10.0, gps_sigma=3.0, dropout=0.0, t0=1494650700.0,
                   lat0=31.1779, lon0=121.3983):
    """Synthetic GPS/SLAM inputs for the track path.

    Returns dict(slam[N,4], enu_true[N,4] (only for sanity), seg_off[nseg+1], gprmc_text, gps(lat,lon,t,valid)).
    Every segment's SLAM track starts at the origin with an unknown heading (LOAM is reset per
    segment: laserOdometry.cpp:519-563), z = 10 (transformMaintenance.cpp:149).
    """
    rng = np.random.default_rng(seed)
    dt = 1.0 / rate_hz
    N = nseg * poses
    xy, _ = smooth_path(N, dt, seed)
    t = t0 + np.arange(N) * dt
    # metres -> degrees around (lat0, lon0); northing = x, easting = y in the reference's convention
    mlat = 111132.0
    mlon = 111320.0 * math.cos(math.radians(lat0))
    # GPS fixes at 1 Hz bracketing the SLAM span, true path + AR(1) noise
    span = (N - 1) * dt
    K = int(math.ceil(span + 0.7)) + 2
    gt = t0 - 0.7 + np.arange(K)
    gx = np.interp(gt, t, xy[:, 0])
    gy = np.interp(gt, t, xy[:, 1])
    noise = np.zeros((K, 2))
    e = rng.normal(0, gps_sigma * math.sqrt(1 - 0.95 ** 2), size=(K, 2))
    for k in range(1, K):
        noise[k] = 0.95 * noise[k - 1] + e[k]
    glat = lat0 + (gx + noise[:, 0]) / mlat
    glon = lon0 + (gy + noise[:, 1]) / mlon
    valid = np.ones(K, dtype=bool)
    if dropout > 0 and K > 6:
        k = 2
        while k < K - 2:
            if rng.random() < dropout / 6.0:
                L = int(rng.integers(2, 11))
                valid[k:min(k + L, K - 2)] = False
                k += L
            else:
                k += 1
    lines = []
    for k in range(K):
        lines.append(gprmc_line(gt[k], glat[k], glon[k], bool(valid[k])))
        lines.append("")
    text = "\n".join(lines) + "\n"
    slam = np.empty((N, 4))
    seg_off = np.arange(nseg + 1, dtype=np.int32) * poses
    for s in range(nseg):
        a, b = seg_off[s], seg_off[s + 1]
        th = rng.uniform(0, 2 * math.pi)
        c, sn = math.cos(th), math.sin(th)
        loc = xy[a:b] - xy[a]
        drift = np.cumsum(rng.normal(0, 0.002, size=(b - a, 2)), axis=0)
        loc = loc + drift
        slam[a:b, 0] = c * loc[:, 0] - sn * loc[:, 1]
        slam[a:b, 1] = sn * loc[:, 0] + c * loc[:, 1]
    slam[:, 2] = 10.0
    slam[:, 3] = t
    return {"slam": slam, "seg_off": seg_off, "gprmc": text, "xy_true": xy, "t": t,
            "gps": (glat, glon, gt, valid)}


def segmented_run(total_poses=3000, long_len=1000, short_len=300, overlap=100, seed=11, dropout=0.1):
    """A whole synthetic run as input_data would publish it: long segments (flag 0) and
    short overlapping segments (flag 1) cut from ONE drive, each re-based to its first pose
    with its own unknown heading (LOAM restarts per segment).  Returns
    (longs, shorts, gprmc_text) with segments as (n,4) arrays."""
    d = track_segments(1, total_poses, seed=seed, dropout=dropout)
    xy, t = d["xy_true"], d["t"]
    rng = np.random.default_rng(seed + 1)

    def make(a, b):
        th = rng.uniform(0, 2 * math.pi)
        c, s = math.cos(th), math.sin(th)
        loc = xy[a:b] - xy[a] + np.cumsum(rng.normal(0, 0.002, size=(b - a, 2)), axis=0)
        seg = np.empty((b - a, 4))
        seg[:, 0] = c * loc[:, 0] - s * loc[:, 1]
        seg[:, 1] = s * loc[:, 0] + c * loc[:, 1]
        seg[:, 2] = 10.0
        seg[:, 3] = t[a:b]
        return seg

    longs = [make(a, min(a + long_len, total_poses)) for a in range(0, total_poses, long_len)]
    shorts = []
    a = 0
    while a < total_poses:
        b = min(a + short_len, total_poses)
        shorts.append(make(a, b))
        if b == total_poses:
            break
        a = b - overlap
    return longs, shorts, d["gprmc"]


def large_demo_like(gprmc_text, rate_hz=10.0, long_m=1000.0, short_m=300.0, overlap_m=100.0, seed=17, smooth_s=9):
    """BASELINE configs[2] substitute (SURVEY 8d "large-demo-like"): the bags of large_size_demo_data are an external
    download, the GPRMC log the reference ships (data/original_gps_data.txt: 2 490 fixes, 41.5 min) is not.  The SLAM
    side is derived from that log: the fixes are smoothed (moving average over `smooth_s` s), sampled at the sweep rate
    and cut by travelled distance the way input_data does with run.sh's defaults (long 1000 m; short 300 m with 100 m
    overlap; a tail shorter than a third of the distance joins the previous segment, input_data.cpp:367-424); every
    segment is re-based to its first pose with its own unknown heading (LOAM restarts per segment) and carries a little
    odometry drift.  Returns (longs, shorts): lists of [n,4] {x, y, z = 10, t} arrays."""
    fixes = []
    for line in gprmc_text.replace("\r", "\n").split("\n"):
        f = line.split(",")
        if len(f) < 8 or f[1] != "$GPRMC" or f[3] != "A":
            continue
        la, lo = float(f[4]), float(f[6])
        lat = int(la / 100) + (la - 100 * int(la / 100)) / 60.0
        lon = int(lo / 100) + (lo - 100 * int(lo / 100)) / 60.0
        fixes.append((float(f[0]), lat if f[5] == "N" else -lat, lon if f[7] == "E" else -lon))
    g = np.array(fixes)
    x = (g[:, 1] - g[0, 1]) * 111132.0  # northing, easting in metres (the shape matters, not the datum)
    y = (g[:, 2] - g[0, 2]) * 111320.0 * math.cos(math.radians(g[0, 1]))
    k = np.ones(smooth_s) / smooth_s
    pad = smooth_s // 2
    xs = np.convolve(np.pad(x, pad, mode="edge"), k, mode="valid")
    ys = np.convolve(np.pad(y, pad, mode="edge"), k, mode="valid")
    t = np.arange(g[2, 0], g[-3, 0], 1.0 / rate_hz)  # strictly inside the log: interPolate drops later stamps
    xy = np.c_[np.interp(t, g[:, 0], xs), np.interp(t, g[:, 0], ys)]
    dist = np.r_[0.0, np.cumsum(np.hypot(np.diff(xy[:, 0]), np.diff(xy[:, 1])))]
    rng = np.random.default_rng(seed)

    def make(a, b):
        th = rng.uniform(0, 2 * math.pi)
        c, s_ = math.cos(th), math.sin(th)
        loc = xy[a:b] - xy[a] + np.cumsum(rng.normal(0, 0.002, size=(b - a, 2)), axis=0)
        seg = np.empty((b - a, 4))
        seg[:, 0] = c * loc[:, 0] - s_ * loc[:, 1]
        seg[:, 1] = s_ * loc[:, 0] + c * loc[:, 1]
        seg[:, 2] = 10.0
        seg[:, 3] = t[a:b]
        return seg

    def cut(D, ov):
        out, a, n = [], 0, len(t)
        while a < n - 1:
            b = int(np.searchsorted(dist, dist[a] + D, side="right"))
            if b >= n or dist[-1] - dist[min(b, n - 1)] < D / 3.0:  # a short rest joins this segment
                b = n
            out.append((a, b))
            if b == n:
                break
            a = int(np.searchsorted(dist, dist[b - 1] - ov, side="left"))
        return out

    longs = [make(a, b) for a, b in cut(long_m, 0.0)]
    shorts = [make(a, b) for a, b in cut(short_m, overlap_m)]
    return longs, shorts


def write_track_file(path, longs, shorts):
    """The driver's track-file format: '<flag> <n>' then n lines 'x y z t'."""
    with open(path, "w") as f:
        for flag, segs in ((0, longs), (1, shorts)):
            for s in segs:
                f.write("%d %d\n" % (flag, len(s)))
                for r in s:
                    f.write("%.17g %.17g %.17g %.17g\n" % tuple(r))


# ------------------------------------------------------------ LOAM feature sweeps
def _small_rot(r):
    """LOAM's rotation order of TransformToStart's inverse: Rz(rz) Rx(rx) Ry(ry) on (x,y,z)."""
    rx, ry, rz = r
    cz, sz, cx, sx, cy, sy = math.cos(rz), math.sin(rz), math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    return Rz @ Rx @ Ry


def loam_sweep(motion, seed=0, n_az=900, sensor_h=1.8):
    """One VLP-16-like sweep of a box room with poles, as LOAM feature clouds.

    `motion` = (rx,ry,rz,tx,ty,tz) of the sensor over the sweep (start frame -> end frame, constant
    rate).  A point captured at relative time s is seen from the pose s*motion.  Returns
    dict(sharp, less_sharp, flat, less_flat) of float32 [n,4] {x,y,z,intensity} in the frame of
    capture; intensity = ring + 0.1*s (scanRegistration.cpp:340-362); clouds are ordered by ring,
    then azimuth.  Axes: LOAM's (y up): ground is the plane y = -sensor_h.
    """
    rng = np.random.default_rng(seed)
    motion = np.asarray(motion, dtype=np.float64)
    elev = np.radians(np.linspace(-15, 15, 16))
    az = np.linspace(-math.pi, math.pi, n_az, endpoint=False)
    poles = np.array([[6, 4], [-7, 5], [9, -6], [-5, -8], [12, 9], [-11, -3], [3, -11], [-2, 12]], dtype=np.float64)
    half = np.array([20.0, 15.0])  # room half sizes in (z, x)
    out = {"sharp": [], "less_sharp": [], "flat": [], "less_flat": []}
    for ring in range(16):
        ce, se = math.cos(elev[ring]), math.sin(elev[ring])
        pts, kind = [], []
        for k in range(n_az):
            s = k / n_az
            R = _small_rot(s * motion[:3])
            t = s * motion[3:]
            d_s = np.array([ce * math.sin(az[k]), se, ce * math.cos(az[k])])  # sensor frame (x left, y up, z fwd)
            d = R @ d_s
            o = t.copy()
            best, bk = 1e9, -1
            if d[1] < -1e-6:  # ground y = -h
                lam = (-sensor_h - o[1]) / d[1]
                if 0 < lam < best:
                    best, bk = lam, 0
            for axis, sign in ((2, 1), (2, -1), (0, 1), (0, -1)):  # walls z = +-20, x = +-15
                lim = half[0] if axis == 2 else half[1]
                if abs(d[axis]) > 1e-9:
                    lam = (sign * lim - o[axis]) / d[axis]
                    if 0 < lam < best:
                        best, bk = lam, 1
            for pc in poles:  # vertical cylinders r = 0.15 at (z, x) = pc
                oz, ox = o[2] - pc[0], o[0] - pc[1]
                a = d[2] * d[2] + d[0] * d[0]
                b = 2 * (oz * d[2] + ox * d[0])
                c = oz * oz + ox * ox - 0.15 * 0.15
                disc = b * b - 4 * a * c
                if disc > 0 and a > 1e-12:
                    lam = (-b - math.sqrt(disc)) / (2 * a)
                    if 0 < lam < best:
                        best, bk = lam, 2
            if bk < 0:
                continue
            pw = o + best * d + rng.normal(0, 0.005, 3)
            ps = R.T @ (pw - t)  # back into the frame of capture
            pts.append([ps[0], ps[1], ps[2], ring + 0.1 * s])
            kind.append(bk)
        pts = np.array(pts, dtype=np.float32)
        kind = np.array(kind)
        pole_idx = np.flatnonzero(kind == 2)
        plane_idx = np.flatnonzero(kind != 2)
        # runs of consecutive pole hits: middle sample = sharp, all = less sharp
        if len(pole_idx):
            runs = np.split(pole_idx, np.flatnonzero(np.diff(pole_idx) > 1) + 1)
            for r_ in runs:
                out["sharp"].append(pts[r_[len(r_) // 2]][None])
                out["less_sharp"].append(pts[r_])
        out["flat"].append(pts[plane_idx[::12]])
        out["less_flat"].append(pts[plane_idx[::3]])
    return {k: np.concatenate(v).astype(np.float32) for k, v in out.items()}


# ------------------------------------------------------------------ raw LiDAR sweeps
# beam elevations scanRegistration.cpp:307-325 maps to ring ids 0..15
SR_RING_DEG = (-15, -13, -11, -9, -7, -5, -4, -3, -2, -1, 0, 1, 3, 5, 7, 9)


def lidar_world(seed=0, length=400.0, half_width=9.0):
    """A street along +x: ground z = 0, building boxes on both sides with random set-backs and gaps,
    poles along the kerbs.  Returns dict(boxes[nb,5] = xmin,xmax,ymin,ymax,height; poles[np,3] = x,y,r)."""
    rng = np.random.default_rng(seed)
    boxes, poles = [], []
    for side in (-1.0, 1.0):
        x = -40.0
        while x < length + 40.0:
            w = rng.uniform(6.0, 18.0)
            setback = rng.uniform(0.0, 4.0)
            depth = rng.uniform(6.0, 12.0)
            y0 = side * (half_width + setback)
            y1 = side * (half_width + setback + depth)
            boxes.append([x, x + w, min(y0, y1), max(y0, y1), rng.uniform(4.0, 12.0)])
            x += w + rng.uniform(1.0, 5.0)
        x = -35.0
        while x < length + 35.0:
            poles.append([x, side * (half_width - 1.5 + rng.uniform(-0.3, 0.3)), rng.uniform(0.08, 0.2)])
            x += rng.uniform(9.0, 22.0)
    # the street is closed at both ends so every azimuth returns something
    boxes.append([-60.0, -45.0, -40.0, 40.0, 10.0])
    boxes.append([length + 45.0, length + 60.0, -40.0, 40.0, 10.0])
    return {"boxes": np.array(boxes), "poles": np.array(poles)}


def raw_sweep(world, pos=(0.0, 0.0), yaw=0.0, vel=(0.0, 0.0), yaw_rate=0.0, seed=0, n_az=1800, sensor_h=1.8,
              sigma=0.004, max_range=80.0, nan_every=0, period=0.1):
    """One 16-ring sweep in the sensor frame (x forward, y left, z up), in firing order: azimuth
    step by step (clockwise, as scanRegistration's ori = -atan2(y, x) assumes), 16 beams per step.
    The sensor moves with `vel` (m/s, world) and turns with `yaw_rate` (rad/s) during the sweep.
    Rays without a return inside max_range are dropped; `nan_every` > 0 replaces every such-th
    point by NaNs (pcl::removeNaNFromPointCloud, SR:266).  float32 [n,3]."""
    rng = np.random.default_rng(seed)
    k = np.repeat(np.arange(n_az), 16)
    ring = np.tile(np.arange(16), n_az)
    s = k / n_az
    phi = math.pi - 2 * math.pi * s  # sensor-frame azimuth, decreasing = clockwise
    el = np.radians(np.array(SR_RING_DEG, dtype=np.float64))[ring]
    ds = np.stack([np.cos(el) * np.cos(phi), np.cos(el) * np.sin(phi), np.sin(el)], axis=1)
    psi = yaw + yaw_rate * period * s
    c, sn = np.cos(psi), np.sin(psi)
    d = np.stack([c * ds[:, 0] - sn * ds[:, 1], sn * ds[:, 0] + c * ds[:, 1], ds[:, 2]], axis=1)
    o = np.stack([pos[0] + vel[0] * period * s, pos[1] + vel[1] * period * s, np.full(len(s), sensor_h)], axis=1)
    best = np.full(len(s), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        lam = -o[:, 2] / d[:, 2]  # ground
        best = np.where((lam > 0.3) & (lam < best), lam, best)
        bx = world["boxes"]
        near = bx[(bx[:, 1] > pos[0] - max_range) & (bx[:, 0] < pos[0] + max_range)]
        for b in near:
            t1 = (b[0] - o[:, 0]) / d[:, 0]
            t2 = (b[1] - o[:, 0]) / d[:, 0]
            t3 = (b[2] - o[:, 1]) / d[:, 1]
            t4 = (b[3] - o[:, 1]) / d[:, 1]
            t5 = (0.0 - o[:, 2]) / d[:, 2]
            t6 = (b[4] - o[:, 2]) / d[:, 2]
            tmin = np.maximum(np.maximum(np.minimum(t1, t2), np.minimum(t3, t4)), np.minimum(t5, t6))
            tmax = np.minimum(np.minimum(np.maximum(t1, t2), np.maximum(t3, t4)), np.maximum(t5, t6))
            hit = (tmax >= tmin) & (tmin > 0.3)
            best = np.where(hit & (tmin < best), tmin, best)
        pl = world["poles"]
        for p in pl[(pl[:, 0] > pos[0] - max_range) & (pl[:, 0] < pos[0] + max_range)]:
            ox, oy = o[:, 0] - p[0], o[:, 1] - p[1]
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            bq = 2 * (ox * d[:, 0] + oy * d[:, 1])
            cq = ox * ox + oy * oy - p[2] * p[2]
            disc = bq * bq - 4 * a * cq
            lam = (-bq - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            z = o[:, 2] + lam * d[:, 2]
            hit = (disc > 0) & (lam > 0.3) & (z < 6.0) & (z > 0.0)
            best = np.where(hit & (lam < best), lam, best)
    keep = np.isfinite(best) & (best < max_range)
    rngd = best[keep] + rng.normal(0, sigma, int(keep.sum()))
    pts = (ds[keep] * rngd[:, None]).astype(np.float32)
    if nan_every > 0:
        pts[::nan_every] = np.nan
    return pts


def drive(world, nsweeps, seed=0, speed=8.0, n_az=900, start=(0.0, 0.0), yaw0=0.0, wiggle=0.02, period=0.1):
    """Raw sweeps of a drive along the street: constant-ish speed with a slow sinusoidal heading
    wiggle.  Returns (list of [n,3] float32 sweeps, stamps[nsweeps], truth[nsweeps,3] = x, y, yaw at
    the START of each sweep)."""
    rng = np.random.default_rng(seed)
    x, y, yaw = float(start[0]), float(start[1]), float(yaw0)
    sweeps, truth = [], []
    for t in range(nsweeps):
        v = speed * (1.0 + 0.1 * math.sin(0.07 * t))
        yaw_rate = wiggle * math.cos(0.05 * t)
        vel = (v * math.cos(yaw), v * math.sin(yaw))
        truth.append((x, y, yaw))
        sweeps.append(raw_sweep(world, pos=(x, y), yaw=yaw, vel=vel, yaw_rate=yaw_rate, seed=int(rng.integers(1 << 30)),
                                n_az=n_az, period=period))
        x += vel[0] * period
        y += vel[1] * period
        yaw += yaw_rate * period
    stamps = 1494650700.0 + period * np.arange(nsweeps)
    return sweeps, stamps, np.array(truth)


def gprmc_for_path(stamps, xy, seed=0, sigma=1.5, origin=(3450164.0, 400633250.0)):
    """A 1 Hz GPRMC log for a path given in local metres (x along the street = north, y = -east ...
    any rigid placement does: the calibration solves for it).  The path is placed at `origin` in the
    reference's UTM-like plane (northing, easting + band offset) and converted with the inverse
    series the reference uses (gps_process.cc:1010-1058) through the CPU restatement's twin in numpy:
    here a plain inverse transverse Mercator, good to centimetres, is enough for test input."""
    rng = np.random.default_rng(seed)
    t0, t1 = math.floor(stamps[0]) - 1, math.ceil(stamps[-1]) + 1
    ts = np.arange(t0, t1 + 1, 1.0)
    x = np.interp(ts, stamps, xy[:, 0]) + rng.normal(0, sigma, len(ts))
    y = np.interp(ts, stamps, xy[:, 1]) + rng.normal(0, sigma, len(ts))
    north = origin[0] + x
    east = origin[1] + y
    band = math.floor(float(east[0]) / 1e7)
    e = east - band * 1e7 - 5e5
    lon0 = math.radians(band * 3.0)
    a, f, k0 = 6378137.0, 1 / 298.257223563, 0.9996
    e2 = f * (2 - f)
    lines = []
    for i in range(len(ts)):
        m = north[i] / k0
        mu = m / (a * (1 - e2 / 4 - 3 * e2 ** 2 / 64 - 5 * e2 ** 3 / 256))
        e1 = (1 - math.sqrt(1 - e2)) / (1 + math.sqrt(1 - e2))
        p1 = (mu + (3 * e1 / 2 - 27 * e1 ** 3 / 32) * math.sin(2 * mu) + (21 * e1 ** 2 / 16 - 55 * e1 ** 4 / 32) * math.sin(4 * mu)
              + (151 * e1 ** 3 / 96) * math.sin(6 * mu))
        n1 = a / math.sqrt(1 - e2 * math.sin(p1) ** 2)
        t1_ = math.tan(p1) ** 2
        c1 = e2 / (1 - e2) * math.cos(p1) ** 2
        r1 = a * (1 - e2) / (1 - e2 * math.sin(p1) ** 2) ** 1.5
        d = e[i] / (n1 * k0)
        lat = p1 - (n1 * math.tan(p1) / r1) * (d * d / 2 - (5 + 3 * t1_ + 10 * c1 - 4 * c1 * c1 - 9 * e2 / (1 - e2)) * d ** 4 / 24)
        lon = lon0 + (d - (1 + 2 * t1_ + c1) * d ** 3 / 6) / math.cos(p1)
        lines.append(gprmc_line(ts[i], math.degrees(lat), math.degrees(lon), True))
    return "\n\n".join(lines) + "\n\n"  # the shipped log separates fixes by an empty line


def write_sweep_file(path, bags, stamps):
    """gpscal_run --sweeps input: "GPSW1\\n", int32 nbag, per bag int32 nsweeps, per sweep
    { float64 stamp; int32 npoints; float32 xyz[3 * npoints] }."""
    import struct
    with open(path, "wb") as f:
        f.write(b"GPSW1\n")
        f.write(struct.pack("<i", len(bags)))
        for bag, st in zip(bags, stamps):
            f.write(struct.pack("<i", len(bag)))
            for sw, t in zip(bag, st):
                sw = np.ascontiguousarray(sw, dtype=np.float32)
                f.write(struct.pack("<di", float(t), len(sw)))
                f.write(sw.tobytes())


def write_rosbag(path, sweeps, stamps, topic="/velodyne_points", chunk_msgs=6, with_ring=True, compression="none",
                 publishers=1, foreign=(), chunk_order=None, corrupt=None):
    """Writes a rosbag V2.0 file holding one sensor_msgs/PointCloud2 message per sweep (Velodyne
    layout: x, y, z, intensity float32 + ring uint16, point_step 32), with connection, chunk, index
    and chunk-info records, as `rosbag record` lays them out.  compression: "none", "bz2" or "lz4".
    publishers > 1: the sweeps alternate between that many connections of the same topic (several publishers);
    foreign: (topic, type, stamp, payload bytes) messages of other topics, merged in by stamp on connections of
    their own; chunk_order: permutation of the chunks in the file (a bag recorded out of time order -- every chunk
    then carries the connection records it needs); corrupt: "field_offset" | "huge_width" | "chunk_size" writes
    a deliberately malformed file (reader robustness tests)."""
    import bz2
    import struct

    def field(name, value):
        b = name.encode() + b"=" + value
        return struct.pack("<I", len(b)) + b

    def record(header_fields, data):
        h = b"".join(field(k, v) for k, v in header_fields)
        return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data

    def ros_time(t):
        secs = int(math.floor(t))
        return struct.pack("<II", secs, int(round((t - secs) * 1e9)) % 1000000000)

    def ros_str(s):
        b = s.encode()
        return struct.pack("<I", len(b)) + b

    msg_def = "# synthetic\n"

    def make_conn(cid, tpc, typ):
        hdr = (field("topic", tpc.encode()) + field("type", typ.encode()) +
               field("md5sum", b"1158d486dd51d683ce2f1be655c3c181") + field("message_definition", msg_def.encode()))
        return record([("op", b"\x07"), ("conn", struct.pack("<I", cid)), ("topic", tpc.encode())], hdr)

    conns = {c: make_conn(c, topic, "sensor_msgs/PointCloud2") for c in range(publishers)}
    ftopics = []
    for ft, fty, _, _ in foreign:
        if (ft, fty) not in ftopics:
            ftopics.append((ft, fty))
            conns[publishers + len(ftopics) - 1] = make_conn(publishers + len(ftopics) - 1, ft, fty)

    def cloud_msg(seq, t, pts):
        n = len(pts)
        body = struct.pack("<I", seq) + ros_time(t) + ros_str("velodyne")
        body += struct.pack("<II", 1, n)
        fields = [("x", 0, 7), ("y", 4, 7), ("z", 8, 7), ("intensity", 16, 7)] + ([("ring", 20, 4)] if with_ring else [])
        if corrupt == "field_offset" and seq == 1:
            fields[2] = ("z", 30, 7)  # 30 + 4 > point_step 32
        if corrupt == "huge_width" and seq == 1:
            body = body[:-8] + struct.pack("<II", 1, 0x7fffffff)
        body += struct.pack("<I", len(fields))
        for name, off, dt in fields:
            body += ros_str(name) + struct.pack("<IBI", off, dt, 1)
        step = 32
        data = np.zeros((n, step), dtype=np.uint8)
        data[:, 0:12] = np.ascontiguousarray(pts, dtype="<f4").view(np.uint8).reshape(n, 12)
        body += struct.pack("<BII", 0, step, step * n) + struct.pack("<I", step * n) + data.tobytes() + struct.pack("<B", 1)
        return body

    # all messages in time order: (stamp, connection, payload)
    allmsgs = [(float(stamps[k]), k % publishers, cloud_msg(k, stamps[k], sweeps[k])) for k in range(len(sweeps))]
    for ft, fty, fst, fpay in foreign:
        allmsgs.append((float(fst), publishers + ftopics.index((ft, fty)), bytes(fpay)))
    allmsgs.sort(key=lambda m: m[0])
    chunks, infos = [], []
    seen = set()
    for c0 in range(0, len(allmsgs), chunk_msgs):
        part = allmsgs[c0:c0 + chunk_msgs]
        body, index = b"", []
        for cid in sorted(set(m[1] for m in part)):
            if chunk_order is not None or cid not in seen:  # a shuffled file repeats them in every chunk
                body += conns[cid]
                seen.add(cid)
        for st_, cid, pay in part:
            index.append((st_, len(body)))
            body += record([("op", b"\x02"), ("conn", struct.pack("<I", cid)), ("time", ros_time(st_))], pay)
        if compression == "bz2":
            payload = bz2.compress(body)
        elif compression == "lz4":  # roslz4 writes standard LZ4 frames
            import ctypes as C
            L = C.CDLL("liblz4.so.1")
            L.LZ4F_compressFrameBound.restype = C.c_size_t
            L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
            L.LZ4F_compressFrame.restype = C.c_size_t
            L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
            cap = L.LZ4F_compressFrameBound(len(body), None)
            dst = C.create_string_buffer(cap)
            n = L.LZ4F_compressFrame(dst, cap, body, len(body), None)
            payload = dst.raw[:n]
        else:
            payload = body
        declared = 0xfffffff0 if corrupt == "chunk_size" and c0 == 0 and compression != "none" else len(body)
        chunk = record([("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", declared))], payload)
        idx = record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 0)),
                      ("count", struct.pack("<I", len(index)))],
                     b"".join(ros_time(t) + struct.pack("<I", o) for t, o in index))
        chunks.append(chunk + idx)
        infos.append((index[0][0], index[-1][0], len(index)))
    if chunk_order is not None:
        chunks = [chunks[i] for i in chunk_order]
        infos = [infos[i] for i in chunk_order]
    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n")
        pos = 13 + 4096
        chunk_pos = []
        for c in chunks:
            chunk_pos.append(pos)
            pos += len(c)
        hdr = b"".join(field(k, v) for k, v in [("op", b"\x03"), ("index_pos", struct.pack("<Q", pos)),
                                                ("conn_count", struct.pack("<I", len(conns))),
                                                ("chunk_count", struct.pack("<I", len(chunks)))])
        pad = 4096 - 4 - len(hdr) - 4
        f.write(struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", pad) + b" " * pad)
        for c in chunks:
            f.write(c)
        for cid in sorted(conns):
            f.write(conns[cid])
        for cp, (t0, t1, cnt) in zip(chunk_pos, infos):
            f.write(record([("op", b"\x06"), ("ver", struct.pack("<I", 1)), ("chunk_pos", struct.pack("<Q", cp)),
                            ("start_time", ros_time(t0)), ("end_time", ros_time(t1)), ("count", struct.pack("<I", 1))],
                           struct.pack("<II", 0, cnt)))
