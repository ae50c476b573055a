#!/usr/bin/env python3
"""Developer tool: profiles/k_trace_traffic.json (what bench.py's `roofline.traffic` and `roofline.valu` quote) from the
outputs of tools/profile_configs.sh.
usage: tools/traffic_from_pmc.py [--append] TAG PROFILES_PREFIX "C1:0:k_trace<false, false, 512, 4, 0>" ...   (config : launch shape : kernel-name filter,
       the specs of tools/profile_configs.sh)
  reads  gpurun_out/TAG/<cfg>/pmc.json and gpurun_out/TAG/<cfg>/stats/**/run_kernel_trace.csv
  writes profiles/k_trace_traffic.json; the entries cite PROFILES_PREFIX_<cfg>_pmc.json as their source

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts KB with 128-byte requests tallied as
64 — doubled here; WRITE_SIZE in KB as is; both come from their own --pmc passes; Infinity-Cache hits are included
(what leaves the L2s, not what reaches HBM). Dispatches shorter than 0.2 x the median (the 1-spp probe launch of
yh_init_state) are left out of every average."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"C1": ("sphere-hairblock", 720, 64), "C2": ("straight-hair", 720, 64), "C3": ("curly-hair", 1280, 32), "C4": ("hair-curls", 1280, 32),
           "lobes": ("lobes", 720, 64), "volumes": ("volumes", 720, 64), "textured": ("textured", 720, 64)}
SCENE_KW = {"C2b": {"beta_m": 0.25}}  # scene overrides of a config name (C2b = C2 at beta_m 0.25: bench.py's config.other_configs)
CONFIGS["C2b"] = CONFIGS["C2"]
XCCS = 8  # GRBM_GUI_ACTIVE comes summed over the eight XCCs
SIMDS, CLOCK = 1024, 2.0e9  # 256 CUs x 4 SIMDs; a vector instruction of a 64-wide wave holds its SIMD for 2 cycles at full rate


def main():
    args = [a for a in sys.argv[1:] if a != "--append"]
    tag, prefix, specs = args[0], args[1], args[2:]
    entries = json.load(open(os.path.join(ROOT, "profiles", "k_trace_traffic.json"))) if "--append" in sys.argv else []
    for spec in specs:
        cfg, shape, kern = spec.split(":", 2)
        d = os.path.join(ROOT, "gpurun_out", tag, cfg)
        p = json.load(open(os.path.join(d, "pmc.json")))
        durs = []
        for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        durs.sort()
        full = [x for x in durs if x >= 0.2 * durs[len(durs) // 2]]
        avg_ms = sum(full) / len(full)
        scene, res, spp = CONFIGS[cfg]
        src = (f"{prefix}_{cfg}_pmc.json (rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE in KB, doubled per "
               "MI355X_MICROARCH.md HBM note: 128-B requests tallied at 64 B; Infinity-Cache hits are counted)")
        sys.path.insert(0, ROOT)
        import bench
        entries.append({
            "csrc_sha16": bench.csrc_sha16(),  # the device code the passes were taken on (bench.py quotes them only for the same)
            "kernel": kern, "launch_shape": int(shape), "config": f"{cfg} {res}x{res}, {spp} spp per launch, 1 GPU", "scene": scene, "resolution": res, "scale": 1.0,
            "scene_kw": SCENE_KW.get(cfg, {}), "spp_per_launch": spp, "source": src,
            "hbm_fetch_bytes_per_launch": p["FETCH_SIZE"] * 1024 * 2, "hbm_write_bytes_per_launch": p["WRITE_SIZE"] * 1024,
            "valu_issue_fraction": p["SQ_ACTIVE_INST_VALU"] / p["SQ_WAVE_CYCLES"], "wait_any_fraction": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"],
            "l2_hit_rate": p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"]),
            "valu_lane_utilisation": p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"]),
            "valu_instructions_per_launch": p["SQ_INSTS_VALU"], "salu_instructions_per_launch": p["SQ_INSTS_SALU"],
            "kernel_avg_ms": avg_ms, "kernel_launches_averaged": len(full),
            "valu_simd_busy_at_2GHz": 2 * p["SQ_INSTS_VALU"] / (SIMDS * CLOCK * avg_ms / 1e3),
            # round 5 (SURVEY.md 8d asks for them next to VALUBusy): MemUnitBusy = the texture-addresser's busy cycles over the GPU's active cycles (rocprof's
            # classic definition; GRBM_GUI_ACTIVE is per XCC, TA_BUSY_avr the mean over the TA instances); achieved occupancy = resident waves per SIMD =
            # OccupancyPercent's expression (400 * SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE / CU_NUM / 32, in per cent of 8 waves per SIMD) x 8 / 100; the share of the
            # L2's read requests that go to DRAM's side of the fabric (the Infinity Cache sits behind that interface: no counter of this rocprofv3 build separates its hits)
            "mem_unit_busy": (p["TA_BUSY_avr"] / (p["GRBM_GUI_ACTIVE"] / XCCS)) if p.get("GRBM_GUI_ACTIVE") else None,
            "mem_unit_busy_max": (p["TA_BUSY_max"] / (p["GRBM_GUI_ACTIVE"] / XCCS)) if p.get("GRBM_GUI_ACTIVE") else None,
            "achieved_waves_per_simd": (4 * p["SQ_WAVE_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / XCCS) / 256 / 4) if p.get("GRBM_GUI_ACTIVE") else None,
            "l2_read_requests_to_dram_share": (p["TCC_EA0_RDREQ_DRAM_sum"] / p["TCC_EA0_RDREQ_sum"]) if p.get("TCC_EA0_RDREQ_sum") else None,
            "mall_hit_rate": None})
        print(cfg, kern, f"{avg_ms:.3f} ms over {len(full)} launches, VALU {p['SQ_INSTS_VALU']:.3g}, lanes {entries[-1]['valu_lane_utilisation']:.3f}, "
              f"issue {entries[-1]['valu_issue_fraction']:.3f}, wait {entries[-1]['wait_any_fraction']:.3f}, L2 hit {entries[-1]['l2_hit_rate']:.3f}, "
              f"fetch {entries[-1]['hbm_fetch_bytes_per_launch'] / 1e9:.2f} GB")
    json.dump(entries, open(os.path.join(ROOT, "profiles", "k_trace_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
