#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

Metric: Msamples/sec (whole node) on tests/sphere-hairblock, 720x720, 1536 spp, eumelanin 1.3
(configs[1], "C1"), with the synthetic 1.6 M-segment hair block of tools/make_scenes.py (the
reference's hair-block.ply is not distributed). samples = width * height * spp; the timed region
is the sample loop only (apps/yscenetrace/yscenetrace.cpp:256-268), scene already resident in
HBM.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path: yh_trace_samples(spp_per_step) over the whole image, i.e.
one k_trace launch adding `spp_per_step` samples to every pixel. The defaults (24 steps of 64
spp) are exactly the 1536 spp the metric is quoted on. For N > 1 the image's 8x8 tiles are dealt
round-robin to the ranks (one process per GPU, launched by torch.distributed.run), every rank
renders all samples of its own tiles with no data-path collective, and ONE RCCL gather of the
packed float4 tiles follows the timed loop (reported as gather_ms).

Scaling is WEAK by default: the units that shard are pixel tiles, and N GPUs render the same scene
with N times the tiles (image side 720 * sqrt(N), rounded to a multiple of 8; 518 400 pixels per
GPU as at N = 1), the way BASELINE.json pairs its 8-GPU configs with larger images. `--strong`
keeps the 720x720 image instead; because a pixel's samples are sequential (one PCG32 stream per
pixel, pt.cpp:1942-1945) the 1536-sample chain of the most expensive pixel bounds that run at any
N.

The JSON line also carries
  roofline:     the dominant kernel (k_trace) against the HBM roofline. achieved = algorithmic
                bytes per launch / average launch duration (HIP events on the kernel's own
                stream); algorithmic bytes per sample = SURVEY.md 8(d)'s formula with the work
                counts N_* measured by the instrumented kernel variant on this scene.
  cpu_baseline: the CPU oracle (oracle/libyh_oracle.so, kind "port", bit-identical to the
                reference in the build container) timed on this host's cores on a bounded
                number of spp of the same scene.
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


def cpu_baseline(scene_json, resolution, budget_s=15.0):
    """Times the CPU oracle (test infrastructure, used here only as the reported baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_capi as oc
    import yhair_capi as yh
    o = oc.Oracle()
    sf = yh.SceneFile(scene_json)
    t0 = time.time()
    sc = o.scene(sf.desc)
    build_s = time.time() - t0
    p = yh.TraceParams.default(resolution=resolution)
    threads = os.cpu_count() or 1
    t0 = time.time()
    img = sc.render(p, 1, nthreads=threads)
    t1 = time.time() - t0
    spp = int(max(1, min(1024, budget_s / max(t1, 1e-3))))
    t0 = time.time()
    img = sc.render(p, spp, nthreads=threads)
    dt = time.time() - t0
    n = img.shape[0] * img.shape[1] * spp
    # work counts of the REFERENCE algorithm (binary BVH, <= 4 primitives per leaf) on this scene:
    # the N_* of SURVEY.md 8(d)'s algorithmic-bytes formula
    _, wc = sc.render(p, 2, nthreads=threads, want_counts=True)
    sc.close()
    sf.close()
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{img.shape[1]}x{img.shape[0]} x {spp} spp of the same scene, {dt:.1f} s, "
                      f"oracle/libyh_oracle.so with {threads} threads (BVH build {build_s:.1f} s not counted)"}, wc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--resolution", type=int, default=720)
    ap.add_argument("--scene", default="sphere-hairblock")
    ap.add_argument("--scale", type=float, default=1.0, help="hair strand-count multiplier (1.0 = the metric's scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save", default="", help="write the final image (.pfm/.hdr) on rank 0")
    ap.add_argument("--strong", action="store_true", help="N > 1: keep the --resolution image (strong scaling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (CPU staging) lets the N > 1 flow be exercised on a one-GPU box")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import make_scenes
    import yhair_capi as yh
    import yhair_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if a.backend == "gloo":  # functional test of the N > 1 flow: ranks may share a GPU
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if a.backend == "nccl" else "cpu"  # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- scene (rank 0 writes the files, everyone loads them) ------------------------------------
    scenes_dir = os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes")
    if rank == 0:
        scene_json = make_scenes.ensure_scene(a.scene, scenes_dir, scale=a.scale)
    if world > 1:
        dist.barrier()
    scene_json = make_scenes.ensure_scene(a.scene, scenes_dir, scale=a.scale)
    ctx = yh.Context(local_rank)
    sf = yh.SceneFile(scene_json)
    t0 = time.time()
    ctx.upload_scene(sf.desc)
    upload_s = time.time() - t0
    segments = sum(sf.desc.contents.shapes[i].num_lines for i in range(sf.desc.contents.num_shapes))
    resolution = a.resolution
    if world > 1 and not a.strong:  # weak scaling: pixels per GPU fixed
        resolution = int(round(a.resolution * world ** 0.5 / 8.0)) * 8
    p = yh.TraceParams.default(resolution=resolution)
    ctx.set_shard(rank, world)
    width, height = ctx.init_state(p)

    # ---- work counts (outside the timed region) ---------------------------------------------------
    # The roofline's algorithmic bytes use the work counts of the REFERENCE algorithm, measured by
    # the CPU oracle on this scene (rank 0; 2 spp). The instrumented kernel's own counts are
    # reported next to them: its 4-wide BVH visits fewer, fatter nodes.
    ALGO = ("samples", "rays", "nodes", "seg_tests", "tri_tests", "hair_shades", "surf_shades", "env_lookups", "env_samples")
    gpu_counts = ctx.trace_samples_counted(2).as_dict()
    ctx.init_state(p)  # start the measured render from a fresh state
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:  # the CPU leg runs at N = 1 only
        cpu, ref_wc = cpu_baseline(scene_json, resolution)
    elif rank == 0:
        ref_wc = None

    # ---- warmup + timed steps ------------------------------------------------------------------
    for _ in range(a.warmup):
        ctx.trace_samples(a.spp_per_step)
    ctx.init_state(p)
    kernel_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.trace_samples(a.spp_per_step)  # one k_trace launch, blocking
        kernel_ms += ctx.last_trace_ms()[0]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = t.tolist()

    # ---- the one collective: gather the float4 framebuffer on rank 0 -----------------------------
    t0 = time.perf_counter()
    n = ctx.shard_pixels(rank, world)
    packed = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # torch's fill is on torch's stream, the pack kernel on the context's
    ctx.pack_tiles_device(packed.data_ptr(), n)
    if cdev == "cpu":
        packed = packed.cpu()
    image = yhair_dist.gather_framebuffer(packed, width, height, rank, world, ctx=ctx)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3

    if rank == 0:
        img = image.cpu().numpy()
        if a.save:
            err = C.create_string_buffer(256)
            yh.load().yh_save_image(a.save.encode(), width, height, yh.fptr(img), err, 256)
        spp_total = a.steps * a.spp_per_step
        samples = width * height * spp_total
        value = samples / elapsed / 1e6
        launch_s = kernel_ms / 1e3 / max(1, a.steps)               # average k_trace duration (per rank)
        if ref_wc is not None:
            counts, counts_from = ref_wc.as_dict(), "reference algorithm (CPU oracle, 2 spp)"
            bytes_per_sample = ref_wc.bytes_per_sample(a.spp_per_step)
        else:
            # no oracle run here (N > 1 or --no-cpu-baseline): the committed counts of the reference
            # algorithm for this scene (tests/golden/workcounts.json, oracle/make_workcounts.py; the
            # per-sample averages do not depend on the image size), else the kernel's own counters
            # with its 4-wide nodes counted as 128 B
            fixture = None
            try:
                fx = json.load(open(os.path.join(ROOT, "tests", "golden", "workcounts.json")))
                cands = [c for c in fx.values() if c["scene"] == a.scene and not c["overrides"] and a.scale == 1.0]
                fixture = min(cands, key=lambda c: abs(c["resolution"] - a.resolution)) if cands else None
            except Exception:
                pass
            if fixture is not None:
                p_ = fixture["per_sample"]
                counts = {k: p_.get(k, 0.0) for k in ALGO[1:]}
                counts["samples"] = 1
                counts_from = "reference algorithm (committed fixture tests/golden/workcounts.json)"
                bytes_per_sample = (32 * p_["nodes"] + 44 * p_["seg_tests"] + 52 * p_["tri_tests"] + 104 * p_["hair_shades"] +
                                    48 * p_["env_lookups"] + 88 * p_["env_samples"]) + 32.0 / a.spp_per_step
            else:
                counts, counts_from = gpu_counts, "instrumented kernel (4-wide BVH; no oracle run, no fixture for this scene)"
                g = gpu_counts
                bytes_per_sample = (128 * g["nodes"] + 44 * g["seg_tests"] + 52 * g["tri_tests"] + 104 * g["hair_shades"] +
                                    48 * g["env_lookups"] + 88 * g["env_samples"]) / max(1, g["samples"]) + 32.0 / a.spp_per_step
        bytes_per_launch = bytes_per_sample * width * height * a.spp_per_step / world
        achieved = bytes_per_launch / launch_s / 1e9
        # HBM traffic of k_trace from the committed PMC passes (profiles/), valid for the default config only
        traffic, traffic_src, pmc = None, None, {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "k_trace_traffic.json")))
            if (pmc["scene"], pmc["resolution"], pmc["spp_per_launch"], pmc["scale"]) == (a.scene, resolution, a.spp_per_step, a.scale) and world == 1:
                traffic = round((pmc["hbm_fetch_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]) / 1e9, 3)
                traffic_src = pmc["source"]
        except Exception:
            pass
        out = {
            "metric": "Msamples/sec (whole node), 720x720x1536spp sphere-hairblock; per-pixel L2 vs CPU ref",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(1, a.steps), 3), "higher_is_better": True,
            "scaling": "strong" if (a.strong and world > 1) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.scene} {width}x{height} x {spp_total} spp (C1: eumelanin 1.3, aspect 1.0), "
                                   f"synthetic hair block {segments} segments x scale {a.scale:g}",
                       "spp_per_step": a.spp_per_step, "bounces": 8, "seed": 961748941,
                       "sharding": f"8x8 tiles round-robin over {world} GPU(s), one RCCL gather after the loop"
                                   + ("" if world == 1 else (" (strong: fixed image)" if a.strong else
                                      f" (weak: image side {a.resolution} * sqrt({world}) -> {width}, {width * height // world} pixels per GPU)")),
                       "upload_s": round(upload_s, 2), "gather_ms": round(gather_ms, 2),
                       "image_mean_rgb": [round(float(x), 5) for x in img[..., :3].mean(axis=(0, 1))]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_unit": "GB per launch",
                         "traffic_source": traffic_src, "algorithmic_gb_per_launch": round(bytes_per_launch / 1e9, 3),
                         "valu_issue_fraction": pmc.get("valu_issue_fraction") if traffic is not None else None,
                         "wait_fraction": pmc.get("wait_any_fraction") if traffic is not None else None,
                         "valu_lane_utilisation": pmc.get("valu_lane_utilisation") if traffic is not None else None,
                         "kernel": "k_trace",
                         "avg_launch_ms": round(launch_s * 1e3, 3),
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                         "work_counts_from": counts_from,
                         "work_counts_per_sample": {k: round(counts[k] / max(1, counts["samples"]), 3) for k in ALGO[1:]},
                         "kernel_counts_per_sample": {k: round(gpu_counts[k] * world / max(1, gpu_counts["samples"] * world), 3)
                                                      for k in ALGO[1:]}},
        }
        if cpu is not None and world == 1:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
