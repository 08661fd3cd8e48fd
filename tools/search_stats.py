"""Per-iteration statistics of the grid search inside icp_step_kernel (instrumented build:
tools/build_variant.sh stats -DGPSCAL_STATS; GPSCAL_LIB=variants/libgpscal_stats.so python tools/search_stats.py).
Lane-level against wave-level counts show what control-flow divergence costs."""
import ctypes
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gpscalibration_amd import Context, synth, _lib
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 16
NSTAT, STAT_ITERS = 24, 64
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
ctx = Context(0)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
L = _lib.load()
buf = (ctypes.c_ulonglong * (NSTAT * STAT_ITERS))()
L.gpscal_debug_stats.restype = ctypes.c_int
L.gpscal_debug_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.gpscal_debug_stats(buf, NSTAT * STAT_ITERS)  # clear
res = sb.icp(iters, want_err=False, profile=True)
ctx.sync()
L.gpscal_debug_stats(buf, NSTAT * STAT_ITERS)
names = {0: "queries", 1: "search_lanes", 2: "search_waves", 3: "level_passes_wave", 4: "level_passes_lane",
         5: "rows_wave", 6: "rows_lane", 7: "groups4_wave", 8: "candidates_lane", 16: "groups4_lane", 9: "coop_runs",
         10: "coop_candidates", 11: "lanes_lvl0", 12: "lanes_lvl1", 13: "lanes_lvl2", 14: "lanes_lvl3", 15: "lanes_lvl4+",
         17: "tier2_new_neighbour"}
out = []
for it in range(iters):
    row = {names[k]: int(buf[it * NSTAT + k]) for k in names}
    q, sl, sw = max(row["queries"], 1), max(row["search_lanes"], 1), max(row["search_waves"], 1)
    row["derived"] = {
        "search_lane_frac": row["search_lanes"] / q,
        "search_wave_frac": row["search_waves"] / (q / 64),
        "level_passes_per_search_wave": row["level_passes_wave"] / sw,
        "rows_per_search_wave": row["rows_wave"] / sw,
        "rows_per_search_lane": row["rows_lane"] / sl,
        "groups4_per_search_wave": row["groups4_wave"] / sw,
        "groups4_per_search_lane": row["groups4_lane"] / sl,
        "candidates_per_search_lane": row["candidates_lane"] / sl,
        "coop_candidates_per_search_wave": row["coop_candidates"] / sw,
    }
    out.append(row)
    d = row["derived"]
    print("it %2d search lanes %5.1f%% waves %5.1f%% | per searching wave: %.2f level passes, %.1f rows, %.1f groups of 4 "
          "(+%.0f coop cand) | per searching lane: %.1f rows, %.1f groups, %.1f candidates | lvl %s" % (
              it, 100 * d["search_lane_frac"], 100 * d["search_wave_frac"], d["level_passes_per_search_wave"],
              d["rows_per_search_wave"], d["groups4_per_search_wave"], d["coop_candidates_per_search_wave"],
              d["rows_per_search_lane"], d["groups4_per_search_lane"], d["candidates_per_search_lane"],
              [row["lanes_lvl%d" % k] if k < 4 else row["lanes_lvl4+"] for k in range(5)]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "search_stats_%d_%d.json" % (npairs, n)), "w"), indent=1)
