#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

Metric: Msamples/sec (whole node) on tests/sphere-hairblock, 720x720, 1536 spp, eumelanin 1.3
(configs[1], "C1"), with the synthetic 1.6 M-segment hair block of tools/make_scenes.py (the
reference's hair-block.ply is not distributed); per-pixel L2 against the CPU reference.
samples = width * height * spp; the timed region is the sample loop only
(apps/yscenetrace/yscenetrace.cpp:256-268), scene already resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config C1|C2|C3|C4]

A "step" is ONE pass of the hot path: yh_trace_samples(spp) over the whole image = one launch of
the sample-loop kernel. The K steps of a run add up to EXACTLY the config's sample count (1536 spp
for C1 / C2, 4096 for C3 / C4): K does not divide it in general, so the first steps carry one
sample more than the last ones (K = 24: 24 x 64; K = 20: 16 x 77 + 4 x 76).

N > 1 (one process per GPU): under torch.distributed.run the ranks are the launcher's; as a bare
`python bench.py --gpus N` this process starts the N ranks itself (plain child processes, before it
touches the GPU, 127.0.0.1 rendezvous) and exits with the first failing rank's code. The image's 8x8
tiles are dealt round-robin to the ranks, every rank renders all samples of its own tiles with no
data-path collective, and ONE RCCL gather of the packed float4 tiles follows the timed loop
(gather_ms). With fewer devices than ranks (a one-GPU box) the ranks share devices and the gather
goes through gloo (RCCL refuses two ranks on one device); config.collective says which ran.
The run is on the CONFIG'S OWN IMAGE (strong scaling, "scaling": "strong") — that is what
BASELINE.json asks to be reported at 1 / 2 / 4 / 8 GPUs; a second, shorter run on an image with N
times the pixels (side x sqrt(N): pixels per GPU as at N = 1) is reported next to it as
config.weak_scaling. `--weak` makes the weak run the reported one.

The JSON line also carries
  roofline      the sample-loop kernel against the HBM roofline (the contract's bound) — achieved =
                algorithmic bytes per launch / average launch duration (HIP events on the kernel's
                own stream) — and, in `valu`, the bound that actually binds: vector-instruction
                issue x active lanes from the committed counter passes (profiles/).
  parity        the metric's second half: per-pixel L2 and relRMSE of the GPU image against the CPU
                image at equal spp and seed, the seed-to-seed floor of the CPU path, and the share
                of pixels within 4 sigma (variance pooled over both seeds of both sides).
  cpu_baseline  the reference itself (oracle/_ref/libyh_ref.so, kind "reference") where it has been
                built (the container that holds /root/reference builds it and the .so travels with
                the repository), else the CPU oracle (kind "port", bit-identical to the reference);
                timed on this host's cores on a bounded number of spp of the same scene.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
K_SIGMA = 4.0

# BASELINE.json configs[1..4]
CONFIGS = {
    "C1": dict(scene="sphere-hairblock", resolution=720, spp=1536, kw={}),
    "C2": dict(scene="straight-hair", resolution=720, spp=1536, kw={}),  # --beta-m 0.1 | 0.25 | 0.6 for the sweep
    "C3": dict(scene="curly-hair", resolution=1280, spp=4096, kw={}),
    "C4": dict(scene="hair-curls", resolution=1280, spp=4096, kw={}),
}
# What the headline line reports next to C1 (config.other_configs): BASELINE.json configs[2..4], C2 at the middle of its beta_m sweep
OTHER_CONFIGS = [("C2", {"beta_m": 0.25}, 8, False), ("C3", {}, 8, True), ("C4", {}, 8, True)]  # (config, scene overrides, launches the spp are cut into, project shard 0 of 8)
KERNELS = {0: "k_trace<512 x 4>", 1: "k_trace<256 x 5>", 2: "k_trace<512 x 4, 8-wide nodes>", 3: "k_stream", 4: "k_trace<256 x 4, octets>",
           5: "k_trace_sbs<512 x 4>: quads + the top items as octets, one launch", 6: "k_trace<256 x 4, 16 lanes per path>", 7: "k_trace<256 x 4, octets, leaf pairs>",
           8: "k_trace<256 x 4, 16 lanes per path, leaf groups>"}


def cpu_leg(scene_json, resolution, budget_s, force_spp=0, full_spp=0, full_budget_s=0.0, want_counts=True):
    """The CPU path on this host (test infrastructure, used here only as the reported baseline and as
    the checker of the `parity` field): two renders of the same scene at equal spp, seeds A and B.
    spp = force_spp if given; else full_spp (the config's own sample count) when one seed of it fits full_budget_s on this
    host (judged from a 1-spp render); else what budget_s allows, at most 256.
    Returns (cpu_baseline dict, image A, image B, spp, seed B, work counts of the reference algorithm or None)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_capi as oc
    import yhair_capi as yh
    threads = os.cpu_count() or 1
    o = oc.Oracle()
    sf = yh.SceneFile(scene_json)
    t0 = time.time()
    osc = o.scene(sf.desc)
    build_s = time.time() - t0
    pa = yh.TraceParams.default(resolution=resolution)
    seed_b = 12345
    pb = yh.TraceParams.default(resolution=resolution, seed=seed_b)
    kind, what = "port", f"oracle/libyh_oracle.so with {threads} threads"
    rsc = None
    if oc.have_ref():
        try:
            rsc = oc.Ref().scene(scene_json)
            kind, what = "reference", f"the reference itself (oracle/_ref/libyh_ref.so, its own std::async row stealing over {threads} hardware threads)"
        except Exception:  # a reference build from another image: fall back to the port
            rsc = None
    render = (lambda p, n: rsc.render(p, n)) if rsc is not None else (lambda p, n: osc.render(p, n, nthreads=threads))
    t0 = time.time()
    render(pa, 1)
    t1 = time.time() - t0
    if force_spp > 0:
        spp = force_spp
    elif full_spp > 0 and full_spp * t1 <= full_budget_s:
        spp = full_spp
    else:
        spp = int(max(1, min(256, budget_s / max(t1, 1e-3))))
    t0 = time.time()
    img_a = render(pa, spp)
    dt = time.time() - t0
    img_b = render(pb, spp)  # the seed-to-seed floor of the CPU path
    n = img_a.shape[0] * img_a.shape[1] * spp
    # work counts of the REFERENCE algorithm (binary BVH, <= 4 primitives per leaf) on this scene:
    # the N_* of SURVEY.md 8(d)'s algorithmic-bytes formula
    wc = osc.render(pa, 2, nthreads=threads, want_counts=True)[1] if want_counts else None
    osc.close()
    if rsc is not None:
        rsc.close()
    sf.close()
    base = {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": f"{img_a.shape[1]}x{img_a.shape[0]} x {spp} spp of the same scene, {dt:.1f} s, {what} "
                      f"(scene load and BVH build {build_s:.1f} s not counted)"}
    return base, img_a, img_b, spp, seed_b, wc


def end_to_end(scene_json, resolution, spp, cpu_rate_msamples):
    """BASELINE's metric times the sample loop only; this is the rest of the job next to it (SURVEY.md 8 f3's reason to exist): the wall-clock
    of the command line a user runs — yocto-hair_amd/yscenetrace scene -r R -s SPP -o out.pfm — on a COLD kernel-trial record (the first run
    on a machine: the trials are cut off the front of the render) and on a WARM one, split by phase (--timing), next to the reference's own
    command line (oracle/_ref/yscenetrace_ref, built from /root/reference where that exists) on the same scene file at a stated reduced spp."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "yocto-hair_amd", "yscenetrace")
    work = tempfile.mkdtemp(prefix="yhair_e2e_")
    out = {"command": f"yocto-hair_amd/yscenetrace {os.path.basename(os.path.dirname(scene_json))}/scene.json -r {resolution} -s {spp} -o out.pfm --timing",
           "what": "wall-clock seconds of the whole command by phase: load_scene = JSON + PLY + HDR parse; bvh_and_upload = flatten + yh_upload_scene (reference-order BVH on "
                   "the device, leaf-ordered records, H2D copies); init_state_and_probe = pixel RNG streams + the 1-spp probe launch; sample_loop = every launch incl. the kernel "
                   "trials of a cold record; download_or_gather; save = .pfm write; process start to exit in total_wall_s"}
    try:
        env = dict(os.environ, YHAIR_CACHE_DIR=os.path.join(work, "cache"))
        env.pop("YHAIR_NO_DISK_CACHE", None), env.pop("YHAIR_SHAPE", None)
        for key in ("cold_trial_record", "warm_trial_record"):
            time.sleep(1.0)  # (a process that starts within milliseconds of the end of another GPU process waits ~ 0.15 s in its first HIP call for the driver to finish with the old one: profiles/r06/e2e_laps_after.txt)
            t0 = time.time()
            r = subprocess.run([exe, scene_json, "-r", str(resolution), "-s", str(spp), "-o", os.path.join(work, "out.pfm"), "--timing"], env=env, capture_output=True, text=True, timeout=300)
            wall = time.time() - t0
            if r.returncode != 0:
                raise RuntimeError(f"yscenetrace exited {r.returncode}: {(r.stdout + r.stderr)[-300:]}")
            t = json.loads(re.search(r"^timing: (\{.*\})$", r.stdout, re.M).group(1))
            t["total_wall_s"] = round(wall, 3)
            t["msamples_per_s_whole_command"] = round(t["width"] * t["height"] * t["samples"] / wall / 1e6, 1)
            out[key] = t
        warm = out["warm_trial_record"]
        shares = {k: warm[k] for k in ("load_scene_s", "convert_s", "bvh_and_upload_s", "lights_s", "init_state_and_probe_s", "download_or_gather_s", "save_s")}
        big = max(shares, key=shares.get)
        out["largest_non_render_phase"] = {"phase": big, "seconds": shares[big], "share_of_total": round(shares[big] / warm["total_s"], 3),
                                           "sample_loop_share_of_total": round(warm["sample_loop_s"] / warm["total_s"], 3)}
        ref = os.path.join(ROOT, "oracle", "_ref", "yscenetrace_ref")
        if os.path.exists(ref):
            rspp = int(max(2, min(256, 8.0 * cpu_rate_msamples * 1e6 / (resolution * resolution)))) if cpu_rate_msamples else 16
            walls = {}
            for n in (1, rspp):
                t0 = time.time()
                r = subprocess.run([ref, scene_json, "-r", str(resolution), "-s", str(n), "-o", os.path.join(work, "ref.pfm")], capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    raise RuntimeError(f"yscenetrace_ref exited {r.returncode}: {(r.stdout + r.stderr)[-300:]}")
                walls[n] = time.time() - t0
            per_spp = (walls[rspp] - walls[1]) / (rspp - 1)
            out["reference_cli"] = {"command": f"oracle/_ref/yscenetrace_ref scene.json -r {resolution} -s N -o ref.pfm", "spp": rspp, "total_wall_s": round(walls[rspp], 3),
                                    "total_wall_s_at_1_spp": round(walls[1], 3), "seconds_per_spp": round(per_spp, 4), "threads": os.cpu_count(),
                                    "fixed_part_s": round(walls[1] - per_spp, 3),
                                    "projected_total_wall_s_at_full_spp": round(walls[1] - per_spp + per_spp * spp, 1),
                                    "note": "the reference's own command line (load, init_bvh, init_lights, sample loop over this host's threads, save); "
                                            "its full-spp time is projected from two runs (time is linear in spp)"}
            out["speedup_whole_command_vs_reference_projected"] = round(out["reference_cli"]["projected_total_wall_s_at_full_spp"] / warm["total_wall_s"], 1)
    except Exception as e:  # never at the expense of the reported line
        out["error"] = str(e)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def parity_field(np, gpu_a, gpu_b, cpu_a, cpu_b, spp):
    """Per-pixel agreement of the GPU image with the CPU image at equal spp and seed (BASELINE.json's
    "per-pixel L2 vs CPU ref"), next to what two seeds of the CPU path differ by."""
    ga, gb, ca, cb = (x[..., :3].astype(np.float64) for x in (gpu_a, gpu_b, cpu_a, cpu_b))
    mean = float(ca.mean())
    l2 = np.sqrt(((ga - ca) ** 2).sum(axis=2))
    l2_floor = np.sqrt(((cb - ca) ** 2).sum(axis=2))
    rel = float(np.sqrt(np.mean((ga - ca) ** 2)) / mean)
    rel_floor = float(np.sqrt(np.mean((cb - ca) ** 2)) / mean)
    # variance of an spp-sample pixel mean, pooled over the four renders (device paths leave the CPU's
    # after a hair bounce or two, so these are four draws of one estimator up to the shared first bounce)
    var = np.stack([ga, gb, ca, cb]).var(axis=0, ddof=1)
    ok = np.abs(ga - ca) <= K_SIGMA * np.sqrt(2.0 * var) + 1e-3 * np.abs(ca) + 1e-6
    return {"spp": spp, "per_pixel_l2_mean": round(float(l2.mean()), 6), "per_pixel_l2_max": round(float(l2.max()), 5),
            "rel_rmse_gpu_vs_cpu": round(rel, 5), "rel_rmse_cpu_seed_floor": round(rel_floor, 5),
            "per_pixel_l2_mean_cpu_seed_floor": round(float(l2_floor.mean()), 6),
            "ratio_to_floor": round(rel / rel_floor, 4) if rel_floor > 0 else None,
            "share_within_4_sigma": round(float(ok.all(axis=2).mean()), 5),
            "alpha_identical": bool(np.array_equal(gpu_a[..., 3] > 0, cpu_a[..., 3] > 0)),
            "note": "same seed, equal spp; tolerance: rel_rmse <= 0.5 x floor and >= 99 % of pixels within 4 sigma (tests/test_gpu_parity.py)"}


def csrc_sha16():
    """Fingerprint of the device code (csrc/*.hip, *.h): a committed counter pass describes this run only if it was
    taken on the same kernels (tools/traffic_from_pmc.py stores it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "yocto-hair_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


ALGO = ("samples", "rays", "nodes", "seg_tests", "tri_tests", "hair_shades", "surf_shades", "env_lookups", "env_samples")


def env_is_textured(desc):
    """SURVEY.md 8(d) counts the environment's bilinear texels for a TEXTURED environment only (a constant one reads nothing:
    eval_environment, pt.cpp:536-547, returns its emission)."""
    d = desc.contents
    return any(d.environments[i].tex_width > 0 for i in range(d.num_environments))


def algorithmic_bytes_per_sample(per_sample, spp_launch, env_textured, node_bytes=32):
    """bytes per camera sample of the reference algorithm from its per-sample work counts."""
    return (node_bytes * per_sample["nodes"] + 44 * per_sample["seg_tests"] + 52 * per_sample["tri_tests"] + 104 * per_sample["hair_shades"] +
            (48 * per_sample["env_lookups"] if env_textured else 0.0) + 88 * per_sample["env_samples"]) + 32.0 / spp_launch


def committed_workcounts(scene_name, scene_kw, resolution, scale):
    """The committed counts of the reference algorithm for this scene (tests/golden/workcounts.json, oracle/make_workcounts.py;
    the per-sample averages do not depend on the image size), or None."""
    try:
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "workcounts.json")))
        kw = {k: float(v) for k, v in scene_kw.items()}
        cands = [c for c in fx.values() if c["scene"] == scene_name and {k: float(v) for k, v in c["overrides"].items()} == kw and scale == 1.0]
        return min(cands, key=lambda c: abs(c["resolution"] - resolution)) if cands else None
    except Exception:
        return None


def committed_counters(scene_name, scene_kw, resolution, scale, world, shape_used):
    """Counters of the sample-loop kernel from the committed PMC passes (profiles/k_trace_traffic.json), valid for the scene,
    overrides, image size, launch shape AND device code they were taken on. Returns (entry or None, why not)."""
    why = None
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "k_trace_traffic.json")))
        allp = allp if isinstance(allp, list) else [allp]
        kw = {k: float(v) for k, v in scene_kw.items()}
        for cand in allp:  # the passes were taken on one kernel: they describe this run only if it chose the same one
            ckw = {k: float(v) for k, v in cand.get("scene_kw", {}).items()}
            if (cand["scene"], cand["resolution"], cand["scale"]) == (scene_name, resolution, scale) and world == 1 and ckw == kw \
                    and cand.get("launch_shape") == shape_used:
                if cand.get("csrc_sha16") != csrc_sha16():  # taken on other device code: not a statement about this run
                    why = f"stale: {cand['source'].split(' ')[0]} was taken on other kernels (csrc fingerprint differs); retake with tools/profile_configs.sh"
                    continue
                return cand, None
    except Exception:
        pass
    return None, why


def valu_block(pmc):
    if not pmc:
        return None
    issue, lanes = pmc.get("valu_issue_fraction"), pmc.get("valu_lane_utilisation")
    return {"bound": "valu", "issue_fraction": issue, "lane_utilisation": lanes,
            "achieved": round(issue * lanes, 4) if issue and lanes else None, "peak": 1.0, "unit": "share of vector lane-cycles doing work",
            "simd_busy": pmc.get("valu_simd_busy_at_2GHz"),
            "wait_fraction": pmc.get("wait_any_fraction"), "l2_hit_rate": pmc.get("l2_hit_rate"),
            "mem_unit_busy": pmc.get("mem_unit_busy"), "achieved_waves_per_simd": pmc.get("achieved_waves_per_simd"),
            "l2_read_requests_to_dram_share": pmc.get("l2_read_requests_to_dram_share"),
            "mall_hit_rate": pmc.get("mall_hit_rate"), "mall_note": "no counter of this rocprofv3 build separates Infinity-Cache hits (FETCH_SIZE counts what leaves the L2s)",
            "source": pmc.get("source")}


def summary_scalars(others, projected, parity, e2e):
    """Every config's value, roofline fraction, parity ratio, CPU rate and 8-GPU projection, and the whole command's wall clock, as ONE flat dict of
    scalars: what bench.py adds to `config` and repeats as the line's last key `summary` (the driver's record keeps the scalars of
    the nested objects and the tail of the line — in round 5 two of the four configs' values were in no driver-written record).
    others: config.other_configs.runs or None; projected: projected_strong_scaling's runs or None; parity: the headline's parity field or None;
    e2e: config.end_to_end or None. A leg that failed contributes `<config>_error`, a leg that did not run contributes nothing."""
    flat = {}
    for oname, r in (others or {}).items():
        k = oname.lower()
        if "error" in r:
            flat[k + "_error"] = str(r["error"])[:100]
            continue
        flat[k + "_msamples"], flat[k + "_frac"], flat[k + "_launch_shape"] = r["value"], r["roofline"]["frac"], r["launch_shape"]
        flat[k + "_launches_in_timed_steps"] = r["launches_in_timed_steps"]
        if isinstance(r.get("parity"), dict) and "ratio_to_floor" in r["parity"]:
            flat[k + "_parity_ratio"], flat[k + "_parity_spp"] = r["parity"]["ratio_to_floor"], r["parity"]["spp"]
            flat[k + "_share_within_4_sigma"] = r["parity"]["share_within_4_sigma"]
        if isinstance(r.get("cpu_baseline"), dict):
            flat[k + "_cpu_msamples"] = r["cpu_baseline"]["value"]
        if isinstance(r.get("projected_8_gpus"), dict) and "seconds" in r["projected_8_gpus"]:
            flat[k + "_proj8_msamples"], flat[k + "_proj8_shard_s"] = r["projected_8_gpus"]["value_if_every_gpu_takes_this_long"], r["projected_8_gpus"]["seconds"]
    for r in (projected or []):
        if "seconds" in r:
            flat[f"c1_proj{r['n_gpus']}_msamples"], flat[f"c1_proj{r['n_gpus']}_shard_s"] = r["value_if_every_gpu_takes_this_long"], r["seconds"]
    if parity is not None:
        flat["c1_parity_ratio"], flat["c1_parity_spp"], flat["c1_share_within_4_sigma"] = parity.get("ratio_to_floor"), parity.get("spp"), parity.get("share_within_4_sigma")
    if e2e and "warm_trial_record" in e2e:
        wm = e2e["warm_trial_record"]
        flat["e2e_warm_total_wall_s"], flat["e2e_warm_msamples_whole_command"] = wm["total_wall_s"], wm["msamples_per_s_whole_command"]
        flat["e2e_warm_bvh_and_upload_s"], flat["e2e_warm_sample_loop_s"] = wm.get("bvh_and_upload_s"), wm.get("sample_loop_s")
        flat["e2e_warm_sample_loop_kernel_s"] = wm.get("sample_loop_kernel_s")
        if "cold_trial_record" in e2e:
            flat["e2e_cold_total_wall_s"] = e2e["cold_trial_record"]["total_wall_s"]
    return flat


def spawn_ranks(n):
    """Bare `bench.py --gpus N`: N rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as the
    launcher would set them). The parent never touches the GPU; it waits, ends the others when one fails, and
    returns the first non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # a failed rank leaves the others waiting in a collective: end them (exact PIDs)
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C1", choices=sorted(CONFIGS), help="BASELINE.json config (C1 = the metric's)")
    ap.add_argument("--spp-per-step", type=int, default=0, help="override: fixed samples per step (total = steps x this)")
    ap.add_argument("--resolution", type=int, default=0, help="override the config's resolution")
    ap.add_argument("--scene", default="", help="override the config's scene")
    ap.add_argument("--beta-m", type=float, default=None, help="C2: straight-hair beta_m sweep value")
    ap.add_argument("--scale", type=float, default=1.0, help="hair strand-count multiplier (1.0 = the config's scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU rendering per seed (two seeds)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples of the CPU leg and of the parity field (default: the config's OWN sample count when one seed of it "
                                                           "fits --cpu-full-budget on this host — C1's 1536 spp take about 40 s per seed on 256 threads — else what --cpu-budget allows, at most 256)")
    ap.add_argument("--cpu-full-budget", type=float, default=75.0, help="seconds one seed of the CPU leg may take at the config's own sample count (two seeds are rendered)")
    ap.add_argument("--other-cpu-budget", type=float, default=6.0, help="seconds of CPU rendering per seed for the parity / cpu_baseline of each of config.other_configs")
    ap.add_argument("--no-end-to-end", dest="end_to_end", action="store_false",
                    help="headline run at N = 1: skip config.end_to_end (the yscenetrace command line on the config's scene file, cold and warm trial record, next to the reference's)")
    ap.add_argument("--no-trial-cache", action="store_true", help="do not keep the kernel-trial record on disk (yh_set_trial_cache_dir; the environment's YHAIR_NO_DISK_CACHE does the same)")
    ap.add_argument("--save", default="", help="write the final image (.pfm/.hdr) on rank 0")
    ap.add_argument("--weak", action="store_true", help="N > 1: report the weak-scaling run (image side x sqrt(N))")
    ap.add_argument("--strong", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (CPU staging) lets the N > 1 flow be exercised on a one-GPU box")
    ap.add_argument("--no-project-scaling", dest="project_scaling", action="store_false",
                    help="N = 1: skip the three extra renders of shard 0 of 2 / 4 / 8 (config.projected_strong_scaling)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="headline run at N = 1: skip the renders of C2 (beta_m 0.25), C3 and C4 at their full sample counts (config.other_configs)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: still create the process group and run the framebuffer gather through the collective "
                         "(a one-rank RCCL communicator: how a one-GPU box executes the nccl branch)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import make_scenes
    import yhair_capi as yh
    import yhair_dist

    # the kernel-trial record on disk is opt-in for library callers (include/yhair.h: yh_set_trial_cache_dir); the bench opts in like the
    # command lines do, so that the ranks of a run and the runs on a node settle on one kernel per image
    trial_cache = None if a.no_trial_cache else yh.set_trial_cache_dir(default=True)

    cfg = dict(CONFIGS[a.config])
    scene_name = a.scene or cfg["scene"]
    base_res = a.resolution or cfg["resolution"]
    scene_kw = dict(cfg["kw"])
    if a.beta_m is not None:
        scene_kw["beta_m"] = a.beta_m
    headline = (a.config == "C1" and not a.scene and not a.resolution and a.scale == 1.0 and not a.spp_per_step and a.beta_m is None)
    # samples of each step: exactly the config's total over the K steps
    if a.spp_per_step:
        step_spp = [a.spp_per_step] * a.steps
    else:
        total = max(cfg["spp"], a.steps)
        step_spp = [total // a.steps + (1 if k < total % a.steps else 0) for k in range(a.steps)]
    spp_total = sum(step_spp)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE=1")
    ndev = max(1, torch.cuda.device_count())  # (counting devices does not initialise the GPU)
    # stdout carries the ONE JSON line and nothing else: whatever native libraries print there (gloo's "[Gloo] Rank ..."
    # banner, RCCL's NCCL_DEBUG lines) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    backend = a.backend
    if backend == "nccl" and world > ndev:  # ranks share a device: RCCL refuses duplicate GPUs, stage through gloo
        backend = "gloo"
    if backend == "gloo":
        local_rank %= ndev
    if world > ndev:  # several ranks render on one device at once: each plans for its share of the resident waves
        os.environ["YHAIR_DEVICE_SHARE"] = str(-(-world // ndev))
    collective = (f"{backend}" + (f" ({world} ranks on {ndev} device(s): CPU staging)" if backend == "gloo" and world > 1 else
                                  " = RCCL" if backend == "nccl" else ""))
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where collective payloads live
    grouped = world > 1 or a.force_collective
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- scene (rank 0 writes the files, everyone loads them) ------------------------------------
    scenes_dir = os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes")
    if rank == 0:
        scene_json = make_scenes.ensure_scene(scene_name, scenes_dir, scale=a.scale, **scene_kw)
    if world > 1:
        dist.barrier()
    scene_json = make_scenes.ensure_scene(scene_name, scenes_dir, scale=a.scale, **scene_kw)
    ctx = yh.Context(local_rank)
    sf = yh.SceneFile(scene_json)
    t0 = time.time()
    ctx.upload_scene(sf.desc)
    upload_s = time.time() - t0
    segments = sum(sf.desc.contents.shapes[i].num_lines for i in range(sf.desc.contents.num_shapes))
    ctx.set_shard(rank, world)
    weak_res = int(round(base_res * world ** 0.5 / 8.0)) * 8

    ctx_launches = []  # kernel launches of each timed step of the last timed_run (1 each once the kernel trials are over)

    def timed_run(resolution, spps, warmup, warm_spps=None):
        """W untimed steps, then the steps of `spps` between barriers; returns the max over ranks."""
        p = yh.TraceParams.default(resolution=resolution)
        width, height = ctx.init_state(p)
        for k in range(warmup):
            ctx.trace_samples(spps[k % len(spps)])
        for n in (warm_spps or []):
            ctx.trace_samples(n)
        # every kernel trial of this image (and shard) before the timed steps, whatever --warmup says: count trial launches,
        # not steps (a trial is a 32-sample launch cut off the front of a request of 64 or more; yh_trials_pending)
        extra = 0
        while ctx.trials_pending() and extra < 16:
            ctx.trace_samples(64)
            extra += 1
        ctx.init_state(p)
        kernel_ms = 0.0
        del ctx_launches[:]
        barrier()
        t0 = time.perf_counter()
        for n in spps:
            ctx.trace_samples(n)  # one launch of the sample-loop kernel, blocking
            ms, launches = ctx.last_trace_ms()
            kernel_ms += ms
            ctx_launches.append(int(launches))
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = t.tolist()
        return width, height, elapsed, kernel_ms, p

    def other_config(name, kw, steps, project8=False):
        """One more BASELINE config on this GPU at its FULL sample count, with its own warm-up (its kernel trials stay out of
        the timed steps): value, ms per step, the kernel chosen and its roofline against the committed work counts and
        counter passes. The scene replaces the context's; the caller is done with the headline scene."""
        c = CONFIGS[name]
        path = make_scenes.ensure_scene(c["scene"], scenes_dir, scale=1.0, **kw)
        osf = yh.SceneFile(path)
        try:
            ctx.upload_scene(osf.desc)
            ctx.set_shard(0, 1)
            spps = [c["spp"] // steps + (1 if k < c["spp"] % steps else 0) for k in range(steps)]
            w, h, el, kms, _ = timed_run(c["resolution"], spps, 0, warm_spps=[96, 96, 96])
            shape = ctx.launch_shape()
            launches = sum(ctx_launches)
            spp_launch, launch_s = c["spp"] / steps, kms / 1e3 / steps
            fx = committed_workcounts(c["scene"], kw, c["resolution"], 1.0)
            env_tex = env_is_textured(osf.desc)
            if fx is not None:
                bps, src = algorithmic_bytes_per_sample(fx["per_sample"], spp_launch, env_tex), "committed fixture tests/golden/workcounts.json"
            else:
                osc_counts = ctx.trace_samples_counted(1).as_dict()
                g = {k: osc_counts[k] / max(1, osc_counts["samples"]) for k in ALGO[1:]}
                bps, src = algorithmic_bytes_per_sample(g, spp_launch, env_tex, node_bytes=128), "instrumented kernel (no fixture)"
            achieved = bps * w * h * spp_launch / launch_s / 1e9
            pmc, why = committed_counters(c["scene"], kw, c["resolution"], 1.0, 1, shape)
            traffic = round((pmc["hbm_fetch_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]) * spp_launch / pmc["spp_per_launch"] / 1e9, 3) if pmc else None
            # the config's own parity statement and CPU baseline: the reference on this host's threads at the config's resolution, at the
            # spp a few seconds per seed allow (stated), and the device at that spp and both seeds
            o_parity = o_cpu = None
            if not a.no_cpu_baseline:
                try:
                    o_cpu, ca, cb, ospp, oseed_b, _ = cpu_leg(path, c["resolution"], a.other_cpu_budget, want_counts=False)
                    oimgs = []
                    for sd in (None, oseed_b):
                        ctx.init_state(yh.TraceParams.default(resolution=c["resolution"]) if sd is None else yh.TraceParams.default(resolution=c["resolution"], seed=sd))
                        ctx.trace_samples(ospp)
                        oimgs.append(ctx.download())
                    o_parity = parity_field(np, oimgs[0], oimgs[1], ca, cb, ospp)
                except Exception as e:  # never at the expense of the throughput numbers
                    o_parity = {"error": str(e)}
            # what ONE GPU of eight would take on this config (its shard rendered alone here, as config.projected_strong_scaling does for the headline):
            # C3 / C4 are the configs BASELINE.json pairs with 8 GPUs
            proj8 = None
            if project8 and a.project_scaling:
                try:
                    ctx.set_shard(0, 8)
                    pw, ph, pel, _, _ = timed_run(c["resolution"], spps, 0, warm_spps=[96, 96, 96])
                    proj8 = {"n_gpus": 8, "shard": "0 of 8", "seconds": round(pel, 4), "value_if_every_gpu_takes_this_long": round(pw * ph * c["spp"] / pel / 1e6, 1),
                             "kernel": KERNELS.get(ctx.launch_shape(), "?"), "launches_in_timed_steps": sum(ctx_launches)}
                except Exception as e:  # never at the expense of the throughput numbers
                    proj8 = {"error": str(e)}
                ctx.set_shard(0, 1)
            return {"workload": f"{name}: {c['scene']} {w}x{h} x {c['spp']} spp" + "".join(f" {k} {v:g}" for k, v in kw.items()),
                    "value": round(w * h * c["spp"] / el / 1e6, 2), "unit": "Msamples/s", "steps": steps, "ms_per_step": round(el * 1e3 / steps, 3),
                    "kernel": KERNELS.get(shape, "?"), "launch_shape": shape, "launches_in_timed_steps": launches, "projected_8_gpus": proj8,
                    "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                                 "traffic": traffic, "traffic_unit": "GB per launch", "traffic_source": pmc["source"] if pmc else why,
                                 "avg_launch_ms": round(launch_s * 1e3, 3), "algorithmic_bytes_per_sample": round(bps, 1), "work_counts_from": src,
                                 "valu": valu_block(pmc)},
                    "parity": o_parity, "cpu_baseline": o_cpu}
        finally:
            osf.close()

    # ---- work counts and the CPU leg (outside the timed region, rank 0 at N = 1 only) --------------
    ctx.init_state(yh.TraceParams.default(resolution=base_res))
    gpu_counts = ctx.trace_samples_counted(2).as_dict()
    gpu_general = any(sf.desc.contents.materials[i].specular or sf.desc.contents.materials[i].metallic or sf.desc.contents.materials[i].transmission
                      or sf.desc.contents.materials[i].opacity < 1 or sf.desc.contents.materials[i].color_tex or sf.desc.contents.materials[i].emission_tex
                      for i in range(sf.desc.contents.num_materials))  # (label only: the GENERAL dense shape runs at 256 x 4)
    cpu = ref_wc = parity = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu, cpu_a, cpu_b, cpu_spp, seed_b, ref_wc = cpu_leg(scene_json, base_res, a.cpu_budget, a.cpu_spp, full_spp=spp_total if headline else 0,
                                                             full_budget_s=a.cpu_full_budget)
        imgs = []
        for sd in (None, seed_b):  # the device at the CPU leg's spp, both seeds (untimed)
            p = yh.TraceParams.default(resolution=base_res) if sd is None else yh.TraceParams.default(resolution=base_res, seed=sd)
            ctx.init_state(p)
            ctx.trace_samples(cpu_spp)
            imgs.append(ctx.download())
        parity = parity_field(np, imgs[0], imgs[1], cpu_a, cpu_b, cpu_spp)

    # ---- the reported run (strong: the config's own image; --weak: N times the pixels) ---------------
    res_main = weak_res if (a.weak and world > 1) else base_res
    width, height, elapsed, kernel_ms, p = timed_run(res_main, step_spp, a.warmup)
    ctx_launches_main = list(ctx_launches)
    shape_used = ctx.launch_shape() if hasattr(ctx, "launch_shape") else None
    trials = {KERNELS.get(k, str(k)): {"ms_per_spp": v[0], "trials": v[1]} for k, v in ctx.kernel_trials().items()}

    # ---- one rank: what ONE GPU of N would take on this image (its shard rendered alone here) ----------
    projected = None
    if world == 1 and a.project_scaling:
        projected = []
        try:
            for n_proj in (2, 4, 8):
                ctx.set_shard(0, n_proj)
                pw, ph, pel, _, _ = timed_run(res_main, step_spp, 3)  # (three warm-up steps: the shard's kernel trials — and the wide BVH nodes the octet kernels build at their first launch — stay out of the timed steps, as the run's own warm-up keeps them out of a rank's)
                projected.append({"n_gpus": n_proj, "shard": f"0 of {n_proj}", "seconds": round(pel, 4),
                                  "value_if_every_gpu_takes_this_long": round(pw * ph * spp_total / pel / 1e6, 1),
                                  "kernel": KERNELS.get(ctx.launch_shape(), "?"),
                                  "launches_in_timed_steps": sum(ctx_launches),  # = the steps: every trial of the shard's (up to five) candidates ran before them
                                  "kernel_trials": {KERNELS.get(k, str(k)): v[1] for k, v in ctx.kernel_trials().items()}})
        except Exception as e:  # never at the expense of the reported line
            projected.append({"error": str(e)})
        ctx.set_shard(0, 1)
        ctx.init_state(p)
        for n in step_spp:  # the image the gather below packs is the full render again
            ctx.trace_samples(n)

    # ---- the one collective: gather the float4 framebuffer on rank 0 -----------------------------
    t0 = time.perf_counter()
    n = ctx.shard_pixels(rank, world)
    packed = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # torch's fill is on torch's stream, the pack kernel on the context's
    ctx.pack_tiles_device(packed.data_ptr(), n)
    if cdev == "cpu":
        packed = packed.cpu()
    image = yhair_dist.gather_framebuffer(packed, width, height, rank, world, ctx=ctx, force_collective=a.force_collective)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3

    # ---- N > 1: "pixels do not depend on the shard", checked on the real collective: the ranks render the config's image
    # once more at a few spp and gather it; rank 0 then renders the WHOLE image alone at the same spp and seed — bitwise equal
    shard_check = None
    if world > 1:
        check_spp = 4
        pc = yh.TraceParams.default(resolution=res_main)
        ctx.set_shard(rank, world)
        ctx.init_state(pc)
        ctx.trace_samples(check_spp)
        npix = ctx.shard_pixels(rank, world)
        pk = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.pack_tiles_device(pk.data_ptr(), npix)
        if cdev == "cpu":
            pk = pk.cpu()
        gathered = yhair_dist.gather_framebuffer(pk, width, height, rank, world, ctx=ctx)
        torch.cuda.synchronize()
        if rank == 0:
            ctx.set_shard(0, 1)
            ctx.init_state(pc)
            ctx.trace_samples(check_spp)
            alone = ctx.download()
            got = gathered.cpu().numpy()
            diff = int(np.any(got != alone, axis=2).sum())
            shard_check = {"what": f"the image gathered from {world} ranks against the same image rendered by rank 0 alone, same seed, {check_spp} spp",
                           "spp": check_spp, "bitwise_equal": diff == 0, "differing_pixels": diff,
                           "collective_ranks": dist.get_world_size(), "backend": dist.get_backend()}
        ctx.set_shard(rank, world)
        dist.barrier()

    # ---- N > 1: the other scaling mode, a shorter run, reported next to the main one -------------------
    other = None
    if world > 1:
        res_other = base_res if a.weak else weak_res
        k_other = max(1, min(4, a.steps))
        ow, oh, oel, _, _ = timed_run(res_other, step_spp[:k_other], 1)
        other = {"mode": "strong" if a.weak else "weak", "image": f"{ow}x{oh}", "steps": k_other,
                 "value": round(ow * oh * sum(step_spp[:k_other]) / oel / 1e6, 2), "unit": "Msamples/s"}

    img = image.cpu().numpy() if rank == 0 else None  # (a copy: the context's buffers are about to be reused)
    # ---- how closely device paths FOLLOW the reference's on the scene where the default BSDF arithmetic shows most: the
    # reference's own sphere-hairblock.json (light hair, colour 0.8: eight-bounce paths) against the reference's images of it
    # (tests/golden/refscenes.npz), with the default arithmetic and with yh_trace_params::hair_exact -------------------
    follow = None
    if rank == 0 and world == 1 and headline:
        try:
            g = np.load(os.path.join(ROOT, "tests", "golden", "refscenes.npz"))
            ref8, ref8_other_seed = g["sphere-hairblock|8"], g["sphere-hairblock|8_seed777"]
            rsf = yh.SceneFile(make_scenes.ensure_scene("ref-sphere-hairblock", scenes_dir, scale=0.05))
            ctx.upload_scene(rsf.desc)
            ctx.set_shard(0, 1)
            relrmse = lambda a, b: float(np.sqrt(np.mean((a[..., :3] - b[..., :3]) ** 2)) / max(1e-12, np.mean(b[..., :3])))
            floor = relrmse(ref8_other_seed, ref8)
            follow = {"scene": "the reference's tests/sphere-hairblock/sphere-hairblock.json (hair colour 0.8) with stand-in geometry x 0.05, "
                               f"{ref8.shape[1]}x{ref8.shape[0]}, 8 spp, against the reference's own image of it (tests/golden/refscenes.npz)",
                      "rel_rmse_cpu_seed_floor": round(floor, 5), "stated_bar": {"fast_bsdf": 0.5, "exact_bsdf": 0.5}}
            for key, exact in (("fast_bsdf", False), ("exact_bsdf", True)):
                ctx.init_state(yh.TraceParams.default(resolution=max(ref8.shape[0], ref8.shape[1]), hair_exact=exact))
                ctx.trace_samples(8)
                follow["ratio_to_floor_" + key] = round(relrmse(ctx.download(), ref8) / floor, 4)
            rsf.close()
        except Exception as e:  # never at the expense of the reported line
            follow = {"error": str(e)}

    # ---- the other BASELINE configs, after everything the headline line needs has been measured -----------
    main_launches = sum(ctx_launches_main)
    others = None
    if rank == 0 and world == 1 and headline and a.other_configs:
        others = {}
        for oname, okw, osteps, oproj in OTHER_CONFIGS:
            try:
                others[oname] = other_config(oname, okw, osteps, oproj)
            except Exception as e:  # never at the expense of the reported line
                others[oname] = {"error": str(e)}

    e2e = None
    if rank == 0 and world == 1 and headline and a.end_to_end:
        e2e = end_to_end(scene_json, base_res, spp_total, cpu["value"] if cpu else None)

    if rank == 0:
        if a.save:
            err = C.create_string_buffer(256)
            yh.load().yh_save_image(a.save.encode(), width, height, yh.fptr(img), err, 256)
        samples = width * height * spp_total
        value = samples / elapsed / 1e6
        launch_s = kernel_ms / 1e3 / max(1, a.steps)               # average launch duration (per rank)
        spp_launch = spp_total / max(1, a.steps)                   # average samples per launch
        env_tex = env_is_textured(sf.desc)
        if ref_wc is not None:
            counts, counts_from = ref_wc.as_dict(), "reference algorithm (CPU oracle, 2 spp)"
            bytes_per_sample = ref_wc.bytes_per_sample(spp_launch, env_tex)
        else:
            # no oracle run here (N > 1 or --no-cpu-baseline): the committed counts of the reference algorithm for this
            # scene, else the kernel's own counters with its 4-wide nodes counted as 128 B
            fixture = committed_workcounts(scene_name, scene_kw, base_res, a.scale)
            if fixture is not None:
                p_ = fixture["per_sample"]
                counts = {k: p_.get(k, 0.0) for k in ALGO[1:]}
                counts["samples"] = 1
                counts_from = "reference algorithm (committed fixture tests/golden/workcounts.json)"
                bytes_per_sample = algorithmic_bytes_per_sample(p_, spp_launch, env_tex)
            else:
                counts, counts_from = gpu_counts, "instrumented kernel (4-wide BVH; no oracle run, no fixture for this scene)"
                g = {k: gpu_counts[k] / max(1, gpu_counts["samples"]) for k in ALGO[1:]}
                bytes_per_sample = algorithmic_bytes_per_sample(g, spp_launch, env_tex, node_bytes=128)
        bytes_per_launch = bytes_per_sample * width * height * spp_launch / world
        achieved = bytes_per_launch / launch_s / 1e9
        traffic = None
        pmc, traffic_src = committed_counters(scene_name, scene_kw, res_main, a.scale, world, shape_used)
        if pmc:
            scale_spp = spp_launch / pmc["spp_per_launch"]  # traffic is proportional to the samples of a launch
            traffic = round((pmc["hbm_fetch_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]) * scale_spp / 1e9, 3)
            traffic_src = pmc["source"]
        valu = valu_block(pmc)
        scaling = "weak" if (a.weak or world == 1) else "strong"
        if headline:
            metric = "Msamples/sec (whole node), 720x720x1536spp sphere-hairblock; per-pixel L2 vs CPU ref"
        else:
            metric = f"Msamples/sec (whole node), {width}x{height}x{spp_total}spp {scene_name}; per-pixel L2 vs CPU ref"
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(1, a.steps), 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "is_headline_config": bool(headline and res_main == base_res),
            "config": {"workload": f"{a.config}: {scene_name} {width}x{height} x {spp_total} spp"
                                   + (" (eumelanin 1.3, aspect 1.0)" if scene_name == "sphere-hairblock" else "")
                                   + (f" beta_m {a.beta_m:g}" if a.beta_m is not None else "")
                                   + f", synthetic hair {segments} segments x scale {a.scale:g}",
                       "spp_per_step": sorted(set(step_spp), reverse=True), "spp_total": spp_total, "bounces": 8, "seed": 961748941,
                       "sharding": f"8x8 tiles round-robin over {world} GPU(s), one RCCL gather after the loop"
                                   + ("" if world == 1 else (f" (weak: image side {base_res} * sqrt({world}) -> {width}, {width * height // world} pixels per GPU)"
                                                             if a.weak else " (strong: the config's own image)")),
                       "collective": collective if grouped else "none (one rank: the packed tiles are un-interleaved in place)",
                       "collective_ranks": dist.get_world_size() if grouped else 1,
                       "upload_s": round(upload_s, 2), "gather_ms": round(gather_ms, 2),
                       "image_mean_rgb": [round(float(x), 5) for x in img[..., :3].mean(axis=(0, 1))]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_unit": "GB per launch",
                         "traffic_source": traffic_src, "algorithmic_gb_per_launch": round(bytes_per_launch / 1e9, 3),
                         "binds": "valu" if valu else None,
                         "note": "hbm is the contract's designated roofline; the scene lives in L2 / Infinity Cache and the kernel is bound by "
                                 "vector-instruction issue (see valu)",
                         "valu": valu,
                         "kernel": KERNELS.get(shape_used, "k_trace") if not (shape_used == 1 and gpu_general) else "k_trace<256 x 4>", "launch_shape": shape_used,
                         "kernel_trials": trials,  # what the choice rests on: the fastest 32-sample trial of every candidate on this image
                         "avg_launch_ms": round(launch_s * 1e3, 3), "avg_spp_per_launch": round(spp_launch, 2),
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                         "work_counts_from": counts_from,
                         "work_counts_per_sample": {k: round(counts[k] / max(1, counts["samples"]), 3) for k in ALGO[1:]},
                         "kernel_counts_per_sample": {k: round(gpu_counts[k] / max(1, gpu_counts["samples"]), 3) for k in ALGO[1:]}},
        }
        if other is not None:
            out["config"]["weak_scaling" if other["mode"] == "weak" else "strong_scaling"] = other
        if projected:
            out["config"]["projected_strong_scaling"] = {
                "what": "shard 0 of N (tile_id % N == 0) of this image rendered ALONE on this GPU, same steps: the time one GPU of N "
                        "would take (the shards are statistically alike: 8x8 tiles dealt round-robin); the gather is not in it. "
                        "A projection from one GPU, not a measurement on N.", "runs": projected}
        out["roofline"]["launches_in_timed_steps"] = main_launches  # K when no kernel trial leaked into the timed region
        if others:
            out["config"]["other_configs"] = {
                "what": "BASELINE.json configs[2..4] on this GPU after the headline run, each at its FULL sample count in 8 launches with its own "
                        "warm-up (three 96-sample launches: the kernel trials); roofline as for the headline (committed work counts / counter passes); "
                        "projected_8_gpus (C3, C4): shard 0 of 8 of the config rendered alone here, as projected_strong_scaling does for the headline",
                "runs": others}
        if e2e:
            out["config"]["end_to_end"] = e2e
        out["config"]["trial_cache_dir"] = trial_cache  # where the kernel-trial record is kept (opt-in; None: nowhere)
        if parity is not None:
            out["parity"] = parity
            if follow is not None:  # the path-following ratio of the DEFAULT arithmetic next to the exact one's (one bar, 0.5, for both since round 4)
                out["parity"]["path_following_light_hair"] = follow
        elif shard_check is not None:  # N > 1: no CPU leg; the parity statement of this line is the shard invariance on the real collective
            out["parity"] = shard_check
        if cpu is not None and world == 1:
            out["cpu_baseline"] = cpu
        # ---- every config's number as SCALARS of `config` (a record that keeps only scalars of the nested objects still shows all four configs and the
        # 8-GPU projections), and once more as `summary`, the LAST key of the line (a record that keeps only the tail of the line shows them too) -------
        flat = summary_scalars(others, projected, parity, e2e)
        out["config"].update(flat)
        out["summary"] = dict({"c1_msamples": out["value"], "c1_frac": out["roofline"]["frac"], "c1_launch_shape": shape_used, "n_gpus": world,
                               "c1_cpu_msamples": cpu["value"] if cpu else None}, **flat)
        print(json.dumps(out), file=json_out, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
