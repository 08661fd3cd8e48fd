"""Per-kernel table from a rocprofv3 --kernel-trace CSV: calls, total / average time, registers, LDS, scratch and the
occupancy those allow on gfx950 (512 VGPRs per SIMD lane in granules of 8, 8 waves per SIMD at most, 160 KiB of LDS
per CU).  python tools/trace_occupancy.py trace.csv out.json"""
import collections
import csv
import json
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("gpscal::", "")
    a = agg.setdefault(n, {"calls": 0, "total_ns": 0, "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                           "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                           "scratch_bytes": int(r["Scratch_Size"]), "workgroup": int(r["Workgroup_Size_X"]),
                           "max_grid": 0})
    a["calls"] += 1
    a["total_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a["max_grid"] = max(a["max_grid"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
tot = sum(a["total_ns"] for a in agg.values())
out = []
for n, a in sorted(agg.items(), key=lambda x: -x[1]["total_ns"]):
    regs = max(a["vgpr"] + a["agpr"], 1)
    alloc = (regs + 7) // 8 * 8
    waves_reg = min(8, 512 // alloc)
    wpg = (a["workgroup"] + 63) // 64
    wg_lds = (160 * 1024) // a["lds_bytes"] if a["lds_bytes"] else 10 ** 6
    waves_lds = min(8, wg_lds * wpg // 4) if a["lds_bytes"] else 8
    occ = max(1, min(waves_reg, waves_lds))
    max_wgs = a["max_grid"] // max(a["workgroup"], 1)
    a.update({"name": n, "avg_us": a["total_ns"] / a["calls"] / 1e3, "share": a["total_ns"] / tot,
              "waves_per_simd_allowed": occ, "limited_by": "vgpr" if waves_reg <= waves_lds else "lds",
              "workgroups_largest_launch": max_wgs,
              "waves_per_simd_largest_launch": round(min(occ, max_wgs * wpg / 1024.0), 2)})
    out.append(a)
    print("%-30s %5d calls %9.2f ms %8.1f us %5.1f%%  vgpr %3d lds %6d scratch %4d wg %3d -> %d waves/SIMD allowed (%s); largest launch %d WGs = %.2f waves/SIMD" % (
        n[:30], a["calls"], a["total_ns"] / 1e6, a["avg_us"], 100 * a["share"], a["vgpr"], a["lds_bytes"], a["scratch_bytes"],
        a["workgroup"], occ, a["limited_by"], max_wgs, a["waves_per_simd_largest_launch"]))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
