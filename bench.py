#!/usr/bin/env python3
"""Headline benchmark: the PathIntegrator at 1920x1080 on the MI355X.

The line's `value` is BASELINE.json configs[1]: the Cornell box with Lambertian + microfacet BxDFs.  The same JSON
object carries, under "bigmesh", configs[2]: the 1.18 M-triangle scene with conductor / dielectric microfacet
materials (the scene the north star's target is stated on), measured the same way with fewer steps.

A "step" renders 16 samples per pixel of the whole 1080p film on every rank (33.2 M camera samples per rank and
step); the 1024-spp job of the configs is 64 such steps.  Ranks take disjoint sample numbers (spp sharding) and the
per-rank films are summed onto rank 0 by ONE ncclReduce inside the library (enigma_film_reduce), inside the timed
region.  value = rays traced by all ranks / wall time (max over ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (one child process per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SPP_PER_STEP = 16
XRES, YRES = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, datasheet
# 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz: lane operations per second, in 1e12 (a SIMD finishes a wave64 instruction every two
# cycles; the 157 TFLOP/s of fp32 vector peak are two flops per such lane operation)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
JOB_SPP = {"bigmesh_fog": 512}  # samples per pixel of the configuration's whole job (1024 unless named here)
# bytes the closest-hit kernel moves per ray besides the tree: the 64-byte path record read, the 16-byte hit record
# written, the queue word read and the material-queue word written
RAY_IO_BYTES = 64 + 16 + 8
TRIANGLE_BYTES = 48


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--variant", default="default",
                    help="default = the Cornell microfacet line with the bigmesh object inside it; one workload only: microfacet | diffuse | specular | "
                         "bigmesh (configs[2], 1.18 M triangles) | manylights (configs[3]) | fog (Cornell with a fog slab, for --integrator volpath) | "
                         "bigmesh_fog (the defined part of configs[4]: the textured 1.18 M-triangle mesh in a homogeneous medium, for --integrator volpath)")
    ap.add_argument("--integrator", default="path", choices=["path", "direct-all", "direct-one", "volpath"],
                    help="path = PathIntegrator (the headline); direct-* = DirectLightingIntegrator with strategy all / one; volpath = VolPathIntegrator")
    ap.add_argument("--light-strategy", default="power", choices=["uniform", "power", "spatial"],
                    help="lightsamplestrategy of the path / volpath integrators (the headline configuration names power)")
    ap.add_argument("--sampler", default="sobol", choices=["sobol", "halton"],
                    help="the metric is quoted on the Sobol' sampler; halton is the reference's default sampler, for comparison")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-spp", type=int, default=0, help="sample numbers per pixel the CPU baseline renders (0 = two per thread for Cornell scenes, one per eight threads for bigmesh: about 10 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every CPU this process may run on)")
    return ap.parse_args()


# ---------------------------------------------------------------- ranks
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """One child process per GPU, started before this process has touched torch or the GPU; rank 0's line is relayed.
    The communicator set-up and the film reduce are collectives without a timeout, so a rank that dies, or one that never
    arrives, would leave the others waiting for good: the first bad exit, or ENIGMA_BENCH_DEADLINE_S seconds (default 1500),
    ends every rank and this process exits non-zero."""
    port = free_port()
    deadline = time.monotonic() + float(os.environ.get("ENIGMA_BENCH_DEADLINE_S", "1500"))
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    timed_out = False
    while True:  # rank 0 prints one short line, so its pipe cannot fill up while nobody reads it
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes) or all(c is not None for c in codes):
            break
        if time.monotonic() > deadline:
            timed_out = True
            break
        time.sleep(0.2)
    if timed_out or any(c not in (None, 0) for c in codes):
        for p in procs:
            if p.poll() is None:
                p.kill()
    out = procs[0].stdout.read().decode()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    if timed_out:
        raise SystemExit("ranks did not finish before the deadline (ENIGMA_BENCH_DEADLINE_S); all were ended: %s" % codes)
    if any(codes):
        raise SystemExit("ranks exited with %s" % codes)


def arm_watchdog():
    """Under a launcher that is not ours (torch.distributed.run) nobody ends a rank whose peers have died inside a collective:
    every rank ends itself, loudly, when the whole run has taken longer than ENIGMA_BENCH_DEADLINE_S seconds."""
    import threading
    limit = float(os.environ.get("ENIGMA_BENCH_DEADLINE_S", "1500"))

    def fire():
        sys.stderr.write("bench.py: rank %s still running after %.0f s (a peer lost inside a collective?); giving up\n" % (os.environ.get("RANK", "0"), limit))
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


# ---------------------------------------------------------------- host description
def host_cpus():
    """CPUs this process may use: the affinity mask, cut down to the cgroup's CPU quota if it has one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        pass
    return n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pow2_at_least(v):
    p = 1
    while p < v:
        p *= 2
    return p


# ---------------------------------------------------------------- one workload
def run_workload(args, variant, steps, ctx):
    """Renders `steps` timed steps of one scene on every rank; returns the rank-0 dictionary (None on other ranks)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import raytrace_amd as ra
    from raytrace_amd import abi, scenes

    rank, world, dev = ctx["rank"], ctx["world"], ctx["dev"]
    K, W, N = steps, args.warmup, world
    if variant == "bigmesh":
        scene, camera, label = scenes.big_mesh(), scenes.big_mesh_camera(XRES, YRES), "BigMesh (terrain + spheres, 1.18 M triangles, conductor/dielectric microfacet)"
    elif variant == "bigmesh_fog":
        scene, camera, label = scenes.big_mesh_fog(), scenes.big_mesh_camera(XRES, YRES), "BigMesh in fog (1.18 M triangles with uvs, trilinear image textures, a homogeneous medium around them)"
    elif variant == "manylights":
        scene, camera, label = scenes.many_lights(), scenes.cornell_camera(XRES, YRES), "ManyLights (512 emissive triangles + 64 point lights)"
    elif variant == "fog":
        scene, camera, label = scenes.cornell_fog("microfacet"), scenes.cornell_camera(XRES, YRES), "Cornell box (microfacet) with a fog slab and a medium-filled glass block"
    else:
        scene, camera, label = scenes.cornell_box(variant), scenes.cornell_camera(XRES, YRES), "Cornell box (%s)" % variant
    film = scenes.film_desc(XRES, YRES, "box")
    # the sampler's samples per pixel: the configuration's own (1024; 512 for the fog configuration) unless more are asked for;
    # the warm-up renders the last 16 sample numbers of that range into a film that is cleared before the timed steps
    total_spp = pow2_at_least(max(JOB_SPP.get(variant, 1024), K * N * SPP_PER_STEP))
    strategy = {"uniform": abi.LIGHTS_UNIFORM, "power": abi.LIGHTS_POWER, "spatial": abi.LIGHTS_SPATIAL}[args.light_strategy]
    if args.integrator == "path":
        integ = scenes.integrator_desc(film, total_spp, max_depth=5, rr_threshold=1.0, light_strategy=strategy)
        integ_label = "PathIntegrator maxdepth 5, Sobol, box filter, %s light sampling" % args.light_strategy
    elif args.integrator == "volpath":
        integ = scenes.volpath_desc(film, total_spp, max_depth=5, rr_threshold=1.0, light_strategy=strategy)
        integ_label = "VolPathIntegrator maxdepth 5, homogeneous media, Sobol, box filter, %s light sampling" % args.light_strategy
    else:
        strat = abi.DL_SAMPLE_ALL if args.integrator == "direct-all" else abi.DL_SAMPLE_ONE
        integ = scenes.direct_lighting_desc(film, total_spp, max_depth=5, strategy=strat)
        integ_label = "DirectLightingIntegrator maxdepth 5 strategy %s, Sobol, box filter" % ("all" if strat == abi.DL_SAMPLE_ALL else "one")
    if args.sampler == "halton":
        integ_label = integ_label.replace("Sobol", "Halton")
    r = ra.Renderer(ctx["local_rank"]).configure(scene, camera, film, integ, sampler=args.sampler)
    torch_reduce = None
    if world > 1:  # the library's own communicator: rank 0 makes the id, torch's process group only carries the 128 bytes
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        ident = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        try:
            # EVERY rank asks for an id: the call loads RCCL into the library, so a rank that cannot load it says so here,
            # before anybody enters the collective ncclCommInitRank (which has no timeout); rank 0's id is the one used
            mine = ra.Renderer.comm_unique_id()
            if rank == 0:
                ident.copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
        except ra.EnigmaError as e:
            sys.stderr.write("rank %d: %s\n" % (rank, e))
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()):
            dist.broadcast(ident, 0)
            try:
                r.comm_init(rank, world, bytes(ident.cpu().numpy().tobytes()))
                if r.comm_info() != (rank, world):
                    raise ra.EnigmaError(-3, "the communicator reports rank %d of %d, expected %d of %d" % (r.comm_info() + (rank, world)))
            except ra.EnigmaError as e:
                sys.stderr.write("rank %d: %s\n" % (rank, e))
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                r.comm_destroy()
        if not int(ok.item()):
            # The library's communicator did not come up on some rank.  A scaling run must not credit torch's reduce to the
            # library: every rank stops here unless the caller asked for the fallback by name (rehearsals on one GPU do).
            if os.environ.get("ENIGMA_BENCH_ALLOW_FALLBACK") != "1":
                if rank == 0:
                    print(json.dumps({"metric": "Mrays/s", "value": None, "n_gpus": world, "reduce": "torch-fallback",
                                      "error": "the library's RCCL communicator did not come up (reason on stderr); set ENIGMA_BENCH_ALLOW_FALLBACK=1 to "
                                               "measure with torch.distributed's reduce instead"}))
                    sys.stdout.flush()
                r.close()
                dist.barrier()
                raise SystemExit(4)
            from raytrace_amd import distributed
            film_view = distributed.film_tensor(r, dev)
            torch_reduce = lambda: dist.reduce(film_view, dst=0, op=dist.ReduceOp.SUM)
    rccl_rank, rccl_world = r.comm_info()
    reduce_film = torch_reduce if torch_reduce is not None else (lambda: r.film_reduce(0))
    reduce_kind = "none (one rank)" if world == 1 else ("torch-fallback" if torch_reduce is not None else "rccl-native")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: the last sample numbers of the range (and one reduce, which sets the communicator's channels up), then a clean film
    for _ in range(W):
        r.render(total_spp - SPP_PER_STEP, total_spp)
    reduce_film()
    r.clear()
    s_before = r.stats()

    barrier()
    t0 = time.perf_counter()
    ms_closest = ms_shadow = ms_shade = 0.0
    n_closest = n_shadow = n_shade = 0
    for i in range(K):
        s0 = (i * N + rank) * SPP_PER_STEP
        r.render(s0, s0 + SPP_PER_STEP)
        st = r.stats()
        ms_closest += st["ms_intersect"]; n_closest += st["intersect_launches"]
        ms_shadow += st["ms_shadow"]; n_shadow += st["shadow_launches"]
        ms_shade += st["ms_shade"]; n_shade += st["shade_launches"]
    reduce_film()
    barrier()
    elapsed = time.perf_counter() - t0

    s_after = r.stats()
    keys = ("camera_samples", "rays_closest", "rays_shadow", "rays_mis", "nodes_closest", "tris_closest", "nodes_shadow", "tris_shadow",
            "node_steps_closest", "tri_steps_closest", "node_steps_shadow", "tri_steps_shadow", "shaded_vertices")
    counts = torch.tensor([s_after[k] - s_before[k] for k in keys], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    samples, rc, rs, rm, nodes_c, tris_c, nodes_s, tris_s, nsteps_c, tsteps_c, nsteps_s, tsteps_s, shaded = [float(v) for v in counts.tolist()]
    rays = rc + rs + rm
    if rank != 0:
        r.close()
        return None

    job_spp = JOB_SPP.get(variant, 1024)
    out = {
        "value": rays / elapsed / 1e6,
        "unit": "Mrays/s",
        "msamples_per_s": samples / elapsed / 1e6,
        "steps": K,
        "ms_per_step": elapsed / K * 1e3,
        # the whole job of the config on these N GPUs (strong scaling): job_spp / 16 / N steps per rank and one film reduce.
        # Measured when exactly that was rendered (job_measured), else the step time scaled to it.
        "job_spp": job_spp,
        "job_seconds": elapsed if K * N * SPP_PER_STEP == job_spp else elapsed / K * (float(job_spp) / SPP_PER_STEP) / N,
        "job_measured": K * N * SPP_PER_STEP == job_spp,
        "config": {"workload": "%s 1920x1080, %s; %d spp per step and GPU (the %d-spp job = %d steps)" % (label, integ_label, SPP_PER_STEP, job_spp, job_spp // SPP_PER_STEP),
                   "triangles": int(scene.indices.shape[0]), "spp_rendered": K * N * SPP_PER_STEP,
                   "parallelism": "spp-shard x%d + one %s of the film" % (N, "ncclReduce inside the library" if torch_reduce is None else "torch.distributed reduce (the library's own RCCL communicator did not come up; the reason is on stderr)")},
        "rccl_ranks": rccl_world,
        "reduce": reduce_kind,
        "ms_bvh_build": s_after["ms_bvh_build"],
        "rays": {"closest": rc, "shadow": rs, "mis": rm, "per_sample": rays / samples},
    }
    out["job_1024spp_seconds"] = out["job_seconds"] * 1024.0 / job_spp

    # ---- rooflines.  Live in this run: launch durations (HIP events on the library's stream), rays, and the bytes THIS
    # build's traversal needs for them (visit counts tallied on the device).  Recorded: what only a counter pass can tell --
    # HBM bytes, L2 hit rate, VALU issue share and lane use per kernel -- read from profiles/traffic.json, which names the
    # commit the passes were taken at; everything under "recorded" is of that commit, not of this run.
    node_bytes = float(s_after["bvh_node_bytes"])
    in_lds = bool(s_after["scene_in_lds"])
    wl_key = "%s_%s_%s" % (variant, args.integrator, args.light_strategy)
    recorded_all = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            recorded_all = json.load(open(tpath)).get(wl_key) or {}
        except Exception:
            recorded_all = {}

    def roofline(kernel_family, kernel_label, ms_total, launches, units_per_launch, bytes_per_unit, extra):
        """One kernel's entry: the roof that binds it (`bound`, `frac`) and the HBM view beside it."""
        avg_ms = ms_total / max(1, launches)
        algorithmic = bytes_per_unit * units_per_launch                     # bytes per launch
        hbm_alg = algorithmic / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        rec = (recorded_all.get("kernels") or {}).get(kernel_family)
        e = {"kernel": kernel_label, "avg_launch_ms": avg_ms, "launches": launches, "units_per_launch": units_per_launch, "bytes_per_unit": bytes_per_unit,
             "hbm_algorithmic": {"achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_alg / HBM_PEAK_GBS}, "traffic": None}
        valu = None
        if rec:
            e["recorded"] = dict(rec, commit=recorded_all.get("commit"), source=recorded_all.get("source"))
            if rec.get("bytes_per_launch") is not None:
                e["traffic"] = rec["bytes_per_launch"]
                # HBM bytes of the counter pass over THIS run's launch time; what of the algorithmic bytes the caches served
                e["hbm_frac_counter"] = rec["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else None
                e["l2_served"] = 1.0 - rec["bytes_per_launch"] / algorithmic if algorithmic > 0 else None
            if rec.get("valu_issue_share") is not None and rec.get("valu_lane_use") is not None:
                valu = rec["valu_issue_share"] * rec["valu_lane_use"]
        hbm_frac = e.get("hbm_frac_counter") or e["hbm_algorithmic"]["frac"]
        if valu is not None and rec["valu_issue_share"] >= hbm_frac:
            # VALU issue is the nearer roof: the fraction of the chip's lane slots (CUs x 4 SIMDs x 32 lanes x clock) that did work
            e.update({"bound": "valu", "frac": valu, "achieved": valu * VALU_PEAK_TLANEOPS, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                      "frac_is": "recorded VALU issue share x lanes active per instruction"})
        else:
            e.update({"bound": "hbm", "frac": hbm_frac, "achieved": hbm_frac * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac_is": "recorded HBM bytes per launch / live launch time" if e.get("hbm_frac_counter") else "algorithmic bytes per launch / live launch time (no counter pass recorded for this workload)"})
        wt = (rec or {}).get("wave_time")
        if wt:
            # neither roof is close when the waves spend most of their time parked on a fetch: the kernel then runs at the speed
            # of its dependent loads times the waves a CU holds, and `frac` says how far from the nearer roof that leaves it
            e["limited_by"] = ("latency of dependent fetches (waves parked on memory %.0f %% of their time, issuing %.0f %%)" % (100 * wt["waiting_on_memory_or_barrier"], 100 * wt["issuing"])
                               if wt["waiting_on_memory_or_barrier"] > wt["issuing"] else "instruction issue (waves issuing %.0f %% of their time)" % (100 * wt["issuing"]))
        e.update(extra)
        return e

    npr, tpr = nodes_c / max(1.0, rc), tris_c / max(1.0, rc)
    # a scene staged into LDS is read from HBM once per block, not once per visit: only the ray records stream
    b_ray = RAY_IO_BYTES + (0.0 if in_lds else node_bytes * npr + TRIANGLE_BYTES * tpr)
    out["roofline"] = roofline("closest", "k_intersect_closest", ms_closest, n_closest, rc / N / max(1, n_closest), b_ray, {
        "nodes_per_ray": npr, "triangles_per_ray": tpr, "node_bytes": node_bytes, "scene_in_lds": in_lds,
        # lanes of a wave that had work in a step (from the device's own step tallies)
        "lanes_busy_node_step": nodes_c / (64.0 * nsteps_c) if nsteps_c else None,
        "triangles_per_tri_step": tris_c / tsteps_c if tsteps_c else None,
        "wave_steps_per_ray": (nsteps_c + tsteps_c) * 64.0 / max(1.0, rc),
        "note": "bytes_per_unit = per ray %d B of path record, hit record and queue words%s" % (
            RAY_IO_BYTES, "; tree and triangles are staged into LDS once per block and cost no HBM bytes per visit" if in_lds
            else " + node_bytes x nodes fetched + 48 B x triangles tested (this build's own traversal, tallied on the device)")})
    if n_shadow:
        nps, tps = nodes_s / max(1.0, rs), tris_s / max(1.0, rs)
        b_sh = 36 + 4 + 16 + (0.0 if in_lds else node_bytes * nps + TRIANGLE_BYTES * tps)  # ray + queue word read, radiance read-modify-write
        stream = (not in_lds) and args.integrator != "volpath"
        out["roofline_shadow"] = roofline("shadow", "k_shadow_tr" if args.integrator == "volpath" else ("k_shadow_stream" if stream else "k_shadow"),
                                          ms_shadow, n_shadow, rs / N / n_shadow, b_sh, dict(
            {"nodes_per_ray": nps, "triangles_per_ray": tps},
            **({"lanes_busy_node_step": nodes_s / (64.0 * nsteps_s), "triangles_per_tri_step": tris_s / tsteps_s if tsteps_s else None,
                "wave_steps_per_ray": (nsteps_s + tsteps_s) * 64.0 / max(1.0, rs)} if nsteps_s else {})))
    if n_shade:
        # the shading kernels of a step together (one per material kind present, + texture / medium / boundary kernels):
        # per shaded path the 64-byte record and queue word read, 48 B of ray, throughput and state written back, the
        # 36-byte shadow ray and two queue words written
        out["roofline_shade"] = roofline("shade", "k_shade<kind> (all kinds of a step)", ms_shade, n_shade, shaded / N / n_shade, 64 + 4 + 48 + 36 + 8,
                                         {"note": "units = path vertices handed to the shading kernels per step (the material queues' sizes, counted on the device)"})

    # ---- CPU baseline: the restated reference renderer on the host cores, a bounded sample of the same workload
    if N == 1 and not args.no_cpu:
        import orc
        avail, quota = host_cpus()
        cores = args.cpu_threads or avail
        # about 10 s of wall time on the threads in use (measured: 0.33 s per Cornell spp and 6 s per BigMesh spp on 16 threads)
        cpu_spp = args.cpu_spp or (max(1, cores // 8) if variant == "bigmesh" else max(8, 2 * cores))
        o = orc.Oracle(scene, camera, film, integ, fast=True, sampler=args.sampler)
        o.render(0, cpu_spp, threads=cores)
        sc = o.stats()
        cpu_rays = sc["rays_closest"] + sc["rays_shadow"] + sc["rays_mis"]
        cpu_s = sc["ms_render"] * 1e-3
        ref_img = o.image()
        o.close()
        ref_nodes = sc["nodes_closest"] / max(1, sc["rays_closest"])
        ref_tris = sc["tris_closest"] / max(1, sc["rays_closest"])
        out["cpu_baseline"] = {"value": cpu_rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "msamples_per_s": sc["samples"] / cpu_s / 1e6, "seconds": cpu_s,
                               "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "usable_cpus": avail, "cgroup_cpu_quota": quota,
                               "sample": "same scene/camera/film, sample numbers [0,%d) of every 1080p pixel (%.1f M samples), "
                                         "oracle built -O3 -march=x86-64-v3 -ffp-contract=off, the reference's HLBVH / 128-primitive tree, 32x32 tile tasks on %d threads"
                                         % (cpu_spp, sc["samples"] / 1e6, cores)}
        # what the REFERENCE's traversal would have to stream for the same rays (its 32-byte nodes, up to 127 primitives per leaf)
        out["roofline"]["ref_nodes_per_ray"] = ref_nodes
        out["roofline"]["ref_tris_per_ray"] = ref_tris
        out["roofline"]["ref_bytes_per_ray"] = 32.0 * ref_nodes + 48.0 * ref_tris + 48.0
        # image agreement on exactly that sample range (GPU re-render, untimed)
        r.clear()
        r.render(0, cpu_spp)
        gpu_img = r.image()
        diff = gpu_img - ref_img
        out["rmse_vs_cpu"] = float(np.sqrt(np.mean(diff * diff)))
        out["mre_vs_cpu"] = float(np.mean(np.abs(diff) / np.maximum(ref_img, 1e-3)))
    r.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # no torch, no GPU in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    watchdog = arm_watchdog() if world > 1 else None  # spawn_ranks has the same deadline; a foreign launcher has none

    import torch
    import torch.distributed as dist

    # Rehearsal knobs for a box with fewer GPUs than ranks (the control flow of N ranks, not a measurement):
    # ENIGMA_BENCH_ONE_GPU=1 puts every rank on device 0, ENIGMA_BENCH_BACKEND=gloo carries torch's collectives without
    # RCCL (which refuses two ranks on one device, as the library's own communicator then does: the fallback path runs).
    if os.environ.get("ENIGMA_BENCH_ONE_GPU"):
        local_rank = 0
    backend = os.environ.get("ENIGMA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = {"rank": rank, "local_rank": local_rank, "world": world, "dev": dev}

    first = "microfacet" if args.variant == "default" else args.variant
    main_res = run_workload(args, first, args.steps, ctx)
    big_res = None
    if args.variant == "default" and args.integrator == "path":
        # the north star's job itself: 1024 spp of the 1.18 M-triangle scene at 1080p = 64 steps, shared by the ranks, one
        # film reduce at the end; measured whole (4.2 s on one GPU), not scaled up from a few steps
        job_steps = JOB_SPP.get("bigmesh", 1024) // SPP_PER_STEP
        big_res = run_workload(args, "bigmesh", max(1, job_steps // world), ctx)

    if rank == 0:
        line = {"metric": "Mrays/s", "value": main_res.pop("value"), "unit": main_res.pop("unit"),
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res.pop("ms_per_step"),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        main_res.pop("steps")
        line.update(main_res)
        if big_res is not None:
            big_res["metric"] = "Mrays/s"
            line["bigmesh"] = big_res
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
