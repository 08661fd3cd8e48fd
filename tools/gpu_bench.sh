#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
rc=$?
tail -3 gpurun_out/bench_full.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value %.0f no_err %.0f ms %.3f frac %.3f search %.3f conv %.3f single %.0f build %.4f incl %.0f" % (d["value"], d["value_no_err"], d["ms_per_step"], r["frac"], r["frac_search"], r["frac_converged"], d["single_pair_iters_per_s"], d["index_build_s"], d["value_incl_build"]))
    print("pose", d["pose_check"])
    print("loam", d["loam_chain"]["gpu_sweeps_per_s"], d["loam_chain_48_segments"]["gpu_sweeps_per_s"])
    print("bag", {k: d["bag_to_kml"][k] for k in ("gpu_wall_s", "gpu_slam_s", "gpu_track_and_kml_s", "segments_total", "bags_total", "cpu_port_slam_s_est")})
except Exception as e:
    print("parse failed", e)
PY
exit $rc
