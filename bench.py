#!/usr/bin/env python3
"""Throughput of the path-tracing pass on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload casa_pt|casa_isect|synth_pt]

One step = one pass of the hot path over one batch:
  casa_pt    one pt_pass_draw (1 sample per pixel) of the casa demo scene at 1920x1080,
             MAX_DEPTH 8, sky NEE + BSDF sampling (the "casa 1080p" configuration the
             headline metric is quoted on; 64 steps = the 64-spp frame)            [default]
  casa_isect one pt_pass_intersect over 16 Mi random incoherent rays vs the casa BVH
  synth_pt   one draw of a procedural 10 M-triangle scene at 3840x2160, MAX_DEPTH 8

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds the
whole scene and renders its own slice of the samples (global sample s belongs to rank
(s-1) % N); accumulators are kept as sums and combined with ONE all-reduce (RCCL) at the
end of the timed region.  Scaling is strong by default then: the K steps are the samples of
ONE frame, cut over the ranks (the multi-GPU benchmark configuration: a fixed-spp frame over
2/4/8 GPUs); --scaling weak gives every rank K steps.  The line carries the time of the
combine (allreduce_ms) and every rank's draw time, so that a flat curve can be attributed.

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="casa_pt", choices=["casa_pt", "casa_isect", "synth_pt"])
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tris", type=int, default=10_000_000, help="synth_pt: triangle count")
    ap.add_argument("--dielectric", type=float, default=0.0, help="synth_pt: fraction of rough dielectric instances")
    ap.add_argument("--textures", type=int, default=0, help="synth_pt: number of procedural RGBA8 albedo maps")
    ap.add_argument("--texture-size", type=int, default=4096)
    ap.add_argument("--rays", type=int, default=1 << 24, help="casa_isect: rays per step")
    ap.add_argument("--pipeline", type=int, default=3, help="draws in flight on the device (pt_pass_set_pipeline); HIP's default of 4 "
                                                            "hardware queues per process holds three and the context's own stream")
    ap.add_argument("--batch", type=int, default=0,
                    help="samples per pt_pass_draw_batch call (1..16): the samples of a frame share the launches of one draw; 0 (default) "
                         "= the rank's samples cut evenly over the draws in flight, at most 16 per draw")
    ap.add_argument("--wide-nodes", type=int, default=-1, choices=[-1, 0, 1],
                    help="node format of the ordered traversal (pt_pass_set_wide_nodes): -1 library default, 0 fp32, 1 quantised")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process; 0 (default) leaves the environment as it is (HIP's default of 4 "
                         "holds 3 draws in flight + the main stream; more draws in flight than that need more).  Read once, when "
                         "the HIP runtime starts, so under rocprofv3 export the variable instead.  Reported in config.")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default) = --steps samples in all, sliced over the ranks (the multi-GPU benchmark "
                         "configuration: a fixed-spp frame cut over 2/4/8 GPUs); weak = every rank draws --steps samples (N x steps in all)")
    ap.add_argument("--collective", default="capi", choices=["capi", "torch"],
                    help="N > 1: combine through the library's own pt_pass_allreduce (RCCL, default) or through torch.distributed")
    ap.add_argument("--traversal", type=int, default=0, choices=[0, 1, 2, 3],
                    help="pt_pass_set_traversal: 0 the library's choice, 1 reference order, 2 ordered over the two levels, "
                         "3 ordered over the flattened tree whatever the scene's size")
    ap.add_argument("--tail", type=int, default=-1,
                    help="pt_pass_set_tail_threshold for the timed region: -1 the library's choice, 0 no tail launch (the counter "
                         "passes of tools/run_pmc.sh use --pipeline 1 --tail 0: the launches the roofline leg times)")
    ap.add_argument("--material-grouping", type=int, default=-1, choices=[-1, 0, 1],
                    help="pt_pass_set_material_grouping: -1 the library's choice (a lone draw: escaped paths apart from vertex 1 on), 0 no grouping, 1 by material type")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank control flow on ONE GPU: all ranks use device 0, gloo backend, collectives staged through the host")
    ap.add_argument("--batches", default=None,
                    help="explicit draw sizes of the timed region, e.g. 4,8,8 (a tuning aid; overrides --batch for the timed region)")
    ap.add_argument("--one-scaling", action="store_true", help="with --gpus N > 1: do not time the other scaling mode as well")
    ap.add_argument("--dump", default=None, help="rank 0 writes the final output buffers to this .npz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel events (no roofline)")
    return ap.parse_args()


def dev_tensor(torch, ptr, n_floats):
    """float32 torch view of device memory owned by the library (no copy)."""
    class _Arr:
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(a, device="cuda")


def main():
    args = parse()
    if args.hw_queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)   # before torch / the HIP runtime start
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start one rank per GPU as child processes (before anything here
        # touches the GPU) and pass rank 0's JSON line through
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ...)" % (args.gpus, world, args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the pass has no CPU path)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from raytracing_amd import device, multigpu, scene

    wl = args.workload
    steps = args.steps if args.steps is not None else (96 if wl == "casa_pt" else (8 if wl == "casa_isect" else 18))
    auto_batch = args.batch <= 0

    def batch_for(n_mine):
        """Samples per draw for a rank that draws n_mine samples: cut evenly over the draws in flight, 16 at most."""
        return max(1, min(16, -(-n_mine // max(args.pipeline, 1)))) if auto_batch else args.batch
    if auto_batch:
        args.batch = batch_for(steps if (world == 1 or args.scaling == "weak") else -(-steps // world))
    warmup = args.warmup if args.warmup is not None else {"casa_pt": 16, "casa_isect": 2}.get(wl, 4)

    t_setup = time.time()
    # large synthetic scenes take minutes to generate and build: say so now and then
    import threading
    setup_done = threading.Event()

    def heartbeat():
        while not setup_done.wait(30.0):
            print("[bench] setting up the scene, %.0f s" % (time.time() - t_setup), file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    if wl == "synth_pt":
        sd = scene.synthetic_scene(args.tris, dielectric_fraction=args.dielectric, textures=args.textures, texture_size=args.texture_size)
        W, H = args.width or 3840, args.height or 2160
        v2w = scene.synthetic_camera()
    else:
        sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
        W, H = args.width or 1920, args.height or 1080
        v2w = scene.casa_camera()
    sky = scene.procedural_sky(2048, 1024)
    ctx = device.Context(local_rank)
    host_build_s = time.time() - t_setup
    # host -> device: the scene arrays and sky textures as the boundary takes them (host pointers),
    # and the re-layout + upload of the traversal arrays (pt_pass_prepare; the first draw would do it otherwise)
    t_up = time.perf_counter()
    sb = device.SceneBuffers(ctx, sd, sky)
    npix = W * H
    p = device.PathTracingPass(ctx, W if wl != "casa_isect" else 16, H if wl != "casa_isect" else 16)
    p.bind_scene(sb)
    p.set_wide_nodes(args.wide_nodes)
    p.set_traversal(args.traversal)
    p.set_material_grouping(args.material_grouping)
    p.set_tail_threshold(args.tail)
    if wl != "casa_isect":
        p.set_pipeline(args.pipeline)
        p.reserve_batch(max(args.batch, batch_for(steps) if world > 1 else 1))
    p.prepare()   # (traversal arrays and the path state of every draw in flight: nothing is allocated inside the timed region)
    ctx.synchronize()
    upload_s = time.perf_counter() - t_up
    upload_bytes = sum(b.nbytes for b in sb.buffers()) + 2 * sky.rgb.nbytes + sum(t.nbytes for t in sd.textures)
    setup_s = time.time() - t_setup
    setup_done.set()

    def note(msg):
        if rank == 0:
            print("[bench] %s (%.1f s)" % (msg, time.time() - t_setup), file=sys.stderr, flush=True)
    note("scene resident on the device")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    comm = collective = total_spp = draw_s = combine_s = None
    my_samples = []
    if wl == "casa_isect":
        lo, hi = sd.world_box()
        rays = scene.random_rays(lo, hi, args.rays, 7 + rank)
        rb = device.Buffer(ctx, rays)
        hb = device.Buffer(ctx, nbytes=args.rays * device.HIT_DTYPE.itemsize)
        for _ in range(warmup):
            p.intersect(rb, args.rays, hb)
        barrier()
        if not args.no_profile:
            p.set_profiling(True)
            p.profile(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            p.intersect(rb, args.rays, hb)
        barrier()
        dt = time.perf_counter() - t0
        total_rays = args.rays * steps
        units_per_rank = dict(rays=total_rays, samples=0, rays_extend=total_rays, rays_shadow=0)
        prof = p.profile() if not args.no_profile else None
        dominant = ("intersect_fast_kernel + intersect_kernel", prof["intersect_ms"], prof["intersect_launches"], total_rays) if prof else None
    else:
        total_spp = steps * world if args.scaling == "weak" else steps
        jits = {total_spp: scene.mt19937_jitters(total_spp)}   # per frame length: the last sample of a frame sits at the pixel centre
        jit = jits[total_spp]
        my_samples = multigpu.sample_slice(total_spp, rank, world)  # 1-based global sample indices

        def params_of(global_s, accum_mode, n_spp=None):
            # currentspp carries the GLOBAL sample index (it seeds the sampler); with sums
            # (accum_mode 1) the accumulation weight does not depend on it
            n_spp = n_spp or total_spp
            if n_spp not in jits:
                jits[n_spp] = scene.mt19937_jitters(n_spp)
            return scene.make_params(W, H, v2w, sky, spp=global_s, frame=1, jitter=jits[n_spp][(global_s - 1) % n_spp],
                                     max_depth=args.depth, light_count=len(sd.lights), accum_mode=accum_mode)

        def draw(global_s, accum_mode):
            p.set_params(params_of(global_s, accum_mode))
            p.draw()

        def prepare_params(samples, accum_mode, n_spp=None):
            """The parameter sets of the samples, made ahead of the draws (in the reference the host sets twenty uniforms per
            sample, main.cpp:185-205; here each set is a Python object with two matrix products behind it)."""
            return [params_of(s, accum_mode, n_spp) for s in samples]

        def draw_all(prms, batch=None):
            """The samples in order: one pt_pass_draw each, or pt_pass_draw_batch calls of --batch samples."""
            batch = batch or args.batch
            if args.batches and len(prms) == sum(int(x) for x in args.batches.split(",")):
                k = 0
                for b in (int(x) for x in args.batches.split(",")):
                    p.draw_batch(prms[k:k + b])
                    k += b
                return
            if batch <= 1:
                for prm in prms:
                    p.set_params(prm)
                    p.draw()
                return
            for k in range(0, len(prms), batch):
                p.draw_batch(prms[k:k + batch])

        draw_all(prepare_params(range(1, 1 + warmup), 0))
        note("warm-up draws queued")
        accum_mode = 1 if world > 1 else 0
        if world > 1:
            # The combine: the library's own (RCCL through the C ABI) unless asked otherwise; RCCL takes one
            # rank per GPU, so a rehearsal of several ranks on one GPU goes through torch.distributed (gloo).
            collective = "torch.distributed" if (args.rehearse or args.collective == "torch") else "pt_pass_allreduce"
            if collective == "pt_pass_allreduce":
                try:
                    comm = multigpu.make_comm(dist, ctx, rank, world)
                except Exception as e:  # RCCL missing or refusing: say so and use the other path on every rank
                    print("[bench] rank %d: pt_comm_create failed (%s)" % (rank, e), file=sys.stderr, flush=True)
                ok = torch.tensor([1 if comm is not None else 0], device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if comm is not None:
                        comm.destroy()
                        comm = None
                    collective = "torch.distributed (pt_comm_create failed)"

            def combine(n_spp):
                if comm is not None:
                    multigpu.allreduce_outputs_capi(comm, p, n_spp)
                    ctx.synchronize()
                else:
                    ctx.synchronize()
                    multigpu.allreduce_outputs(torch, dist, p, npix, n_spp, dev_tensor, via_cpu=args.rehearse)
            # warm the collective up too (communicator, channels, staging buffers): its first use
            # costs far more than the later ones and does not belong to the frame
            combine(max(warmup, 1) * world)
            for name in multigpu.ACCUMULATED:   # sums start from zero on every rank
                p.outputs[name].fill(0.0)
        p.set_pipeline(args.pipeline)

        def timed_frame(n_spp, samples, batch=None):
            """One frame of n_spp samples in all, of which this rank draws `samples`: barrier, draws, combine, barrier."""
            prms = prepare_params(samples, accum_mode, n_spp)
            barrier()
            p.stats(reset=True)
            t_a = time.perf_counter()
            draw_all(prms, batch)
            d_s = c_s = None
            if world > 1:
                ctx.synchronize()                      # this rank's slice is drawn
                d_s = time.perf_counter() - t_a
                combine(n_spp)                         # (waits for the collectives)
                c_s = time.perf_counter() - t_a - d_s  # includes waiting for the slowest rank's slice
            barrier()
            return time.perf_counter() - t_a, prms, d_s, c_s, p.stats(reset=True)

        dt, timed_params, draw_s, combine_s, st = timed_frame(total_spp, my_samples)
        note("timed region done")
        if args.dump and rank == 0:
            np.savez(args.dump, **p.download())
        other = None
        if world > 1 and not args.one_scaling:
            # the other way of cutting the work, timed the same way (not `value`): a flat strong-scaling curve of a small
            # frame and a steep weak-scaling one say different things, so the line carries both
            o_mode = "weak" if args.scaling == "strong" else "strong"
            o_spp = steps * world if o_mode == "weak" else steps
            for name in multigpu.ACCUMULATED:
                p.outputs[name].fill(0.0)
            o_samples = multigpu.sample_slice(o_spp, rank, world)
            o_dt, _, o_draw, o_comb, o_st = timed_frame(o_spp, o_samples, batch_for(len(list(o_samples))))
            other = dict(scaling=o_mode, dt=o_dt, total_spp=o_spp, rays=o_st["rays_extend"] + o_st["rays_shadow"] - o_st["rays_shadow_null"],
                         samples=o_st["samples"], draw_ms=o_draw * 1e3, combine_ms=o_comb * 1e3)
            note("the other scaling mode timed")
        # rays = queries that were TRACED: the reference's occlusion queries for light samples that are exactly
        # zero (pt_stats.rays_shadow_null) are decided without a traversal and not counted as work done
        units_per_rank = dict(rays=st["rays_extend"] + st["rays_shadow"] - st["rays_shadow_null"], samples=st["samples"],
                              rays_extend=st["rays_extend"], rays_shadow=st["rays_shadow"] - st["rays_shadow_null"],
                              shadow_null=st["rays_shadow_null"], retraced=st["rays_retraced"])
        # Per-kernel times: the same draws once more, one at a time (overlapping draws would
        # be charged each other's time), bracketed by HIP events on the launching stream.
        prof = dominant = None
        if not args.no_profile:
            p.set_pipeline(1)
            n_prof = min(len(my_samples), 16)
            # (a) what a caller that issues one draw and then consumes it gets: draws one at a time, as shipped
            #     (the thin tail of each draw in one launch), no events around the launches
            prof_params = timed_params[:n_prof]
            for prm in prof_params[:2]:
                p.set_params(prm)
                p.draw()
            ctx.synchronize()
            each = []
            for prm in prof_params:
                t1 = time.perf_counter()
                p.set_params(prm)
                p.draw()
                ctx.synchronize()
                each.append((time.perf_counter() - t1) * 1e3)
            # (the median: one draw that meets a busy host thread or a clock ramp does not move it; the mean rides along)
            serial_ms, serial_mean_ms = float(np.median(each)), float(np.mean(each))
            # (a') the same with the samples drawn as the timed region draws them (batches), one draw in flight
            batch_ms = None
            if args.batch > 1:
                bp = timed_params[:args.batch * 3]
                p.draw_batch(bp[:args.batch])
                ctx.synchronize()
                each_b = []
                for k in range(0, len(bp), args.batch):
                    t1 = time.perf_counter()
                    p.draw_batch(bp[k:k + args.batch])
                    ctx.synchronize()
                    each_b.append((time.perf_counter() - t1) * 1e3 / len(bp[k:k + args.batch]))
                batch_ms = float(np.median(each_b))
            # (b) per-kernel times: the draws of the timed region once more (same batches), one at a time, with events around
            #     every launch (tail launch off, so that every ray is traced by the kernels that are timed)
            p.set_tail_threshold(0)
            p.set_profiling(True)
            p.profile(reset=True)
            p.stats(reset=True)
            ev_params = timed_params[:max(args.batch, 1) * 2] if args.batch > 1 else prof_params
            draw_all(ev_params)
            ctx.synchronize()
            prof = p.profile()
            pst = p.stats(reset=True)
            p.set_profiling(False)
            p.set_tail_threshold(args.tail)
            p.set_pipeline(args.pipeline)
            note("per-kernel profiling pass done")
            prof["samples_timed"] = len(ev_params)
            prof["batch_ms_per_step"] = batch_ms
            prof["draws"] = n_prof
            prof["serial_ms_per_step"] = serial_ms
            prof["serial_mean_ms_per_step"] = serial_mean_ms
            # every BVH traversal launch of the draws: extend at vertex 0, then per vertex one launch
            # tracing shadow(d) + extend(d+1) rays, each followed by the small reference-order re-trace
            dominant = ("path_trace_fast_kernel + extend_fast_kernel + extend_kernel",
                        prof["extend_ms"] + prof["trace_ms"] + prof["shadow_ms"],
                        prof["extend_launches"] + prof["trace_launches"] + prof["shadow_launches"],
                        pst["rays_extend"] + pst["rays_shadow"] - pst["rays_shadow_null"])


    # max time over ranks, sum of units over ranks
    if world > 1:
        red_dev = "cpu" if args.rehearse else "cuda"
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        uu = torch.tensor([units_per_rank["rays"], units_per_rank["samples"]], dtype=torch.float64, device=red_dev)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
        total_rays, total_samples = float(uu[0].item()), float(uu[1].item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"draw_ms": (draw_s or 0.0) * 1e3, "combine_ms": (combine_s or 0.0) * 1e3,
                                          "samples": len(list(my_samples)) if wl != "casa_isect" else None})
        other_all = [None] * world
        dist.all_gather_object(other_all, other if wl != "casa_isect" else None)
    else:
        total_rays, total_samples = float(units_per_rank["rays"]), float(units_per_rank["samples"])

    if rank == 0:
        out = {
            "metric": "Mrays/s",
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "casa_obj.glb (reference demo scene) + procedural 2048x1024 sky" if wl != "synth_pt"
                    else "synthetic procedural scene + procedural 2048x1024 sky",
            "config": {"workload": {"casa_pt": "casa 1920x1080 path trace, 1 spp per step, max depth %d, sky NEE + BSDF sampling" % args.depth,
                                    "casa_isect": "intersection only: %d random incoherent rays vs casa BVH per step" % args.rays,
                                    "synth_pt": "synthetic %d-triangle scene %dx%d path trace, 1 spp per step, max depth %d, %d%% dielectric instances, %d textures of %d^2" % (len(sd.triangles), W, H, args.depth, round(args.dielectric * 100), args.textures, args.texture_size)}[wl],
                       "resolution": [W, H] if wl != "casa_isect" else None,
                       "triangles": int(len(sd.triangles)), "instances": int(len(sd.instances)),
                       "draws_in_flight": args.pipeline if wl != "casa_isect" else None,
                       "samples_per_draw": args.batch if wl != "casa_isect" else None,
                       "traversal": args.traversal, "flat_leaves": p.stats()["flat_leaves"],
                       "parallelism": "sample slicing x%d, one all-reduce" % world if world > 1 else "single GPU",
                       "collective": collective if wl != "casa_isect" else None,
                       "total_spp": total_spp if wl != "casa_isect" else None,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "msamples_per_s": total_samples / dt / 1e6,
            # N > 1: the combine (one all-reduce of the 9 accumulated floats per pixel + one broadcast of the 7 last-write ones + the
            # scale) as the fastest rank saw it -- the others' figures include waiting for the slowest slice -- and every rank's draw time
            "allreduce_ms": min(r["combine_ms"] for r in per_rank) if world > 1 and wl != "casa_isect" else None,
            "per_rank": per_rank if world > 1 else None,
            # N > 1: the same job cut the other way (weak: `steps` samples per rank; strong: `steps` samples in all), same timing
            "other_scaling": ({"scaling": other_all[0]["scaling"], "total_spp": other_all[0]["total_spp"],
                               "value": sum(o["rays"] for o in other_all) / max(o["dt"] for o in other_all) / 1e6,
                               "msamples_per_s": sum(o["samples"] for o in other_all) / max(o["dt"] for o in other_all) / 1e6,
                               "ms_per_step": max(o["dt"] for o in other_all) / steps * 1e3,
                               "allreduce_ms": min(o["combine_ms"] for o in other_all),
                               "draw_ms_per_rank": [o["draw_ms"] for o in other_all]}
                              if world > 1 and other_all[0] else None),
            "retraced_fraction_of_extend_rays": (units_per_rank.get("retraced", 0) / units_per_rank["rays_extend"])
                                                if units_per_rank.get("rays_extend") else None,
            "rays_per_sample": (total_rays / total_samples) if total_samples else None,
            # rank 0's share: occlusion queries the reference issues for light samples that are exactly zero; the
            # pass decides them without a traversal and `value` does not count them
            "reference_shadow_rays_not_traced_per_sample": (units_per_rank.get("shadow_null", 0) / units_per_rank["samples"])
                                                           if units_per_rank.get("samples") else None,
            "setup_s": setup_s,
            # not part of `value`: what it costs to hand the scene over from host memory once
            "host_to_device": {"bytes": int(upload_bytes), "seconds": upload_s,
                               "includes": "pt_buf/pt_tex uploads (PCIe) + host re-layout of the traversal arrays + their upload",
                               "value_including_it": total_rays / (dt + upload_s) / 1e6},
        }
        # ---- CPU baseline + algorithmic bytes per ray (oracle, bounded sample)
        b_ray = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle
            osc = oracle.OracleScene(sd, sky)
            cores = min(16, os.cpu_count() or 1)
            if wl == "casa_isect":
                n_s = min(args.rays, 1 << 21)
                t1 = time.perf_counter()
                hits, cnt = osc.intersect(rays[:n_s], n_threads=cores)
                cdt = time.perf_counter() - t1
                got = hb.download(np.uint8).view(device.HIT_DTYPE)[:n_s]
                parity = bool(np.array_equal(got["triangle"], hits["triangle"]) and np.array_equal(got["instance"], hits["instance"])
                              and np.array_equal(got["t"].view(np.uint32), hits["t"].view(np.uint32)))
                b_ray = oracle.algorithmic_bytes_per_ray(cnt)
                out["cpu_baseline"] = {"value": n_s / cdt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                                       "sample": "first %d of the %d rays of one step" % (n_s, args.rays)}
                out["parity_vs_oracle"] = "bit-exact ids and t on the sample" if parity else "MISMATCH"
            else:
                # the sample: the last spp of the timed frame, rendered again on the CPU
                sw, sh = (W, H) if wl == "casa_pt" else (W // 4, H // 4)
                s_last = steps
                prm = scene.make_params(sw, sh, v2w, sky, spp=1, frame=s_last, jitter=jit[s_last - 1], max_depth=args.depth,
                                        light_count=len(sd.lights))
                note("oracle scene built, rendering the CPU sample")
                t1 = time.perf_counter()
                ref, cnt = osc.render(prm, n_threads=cores)
                cdt = time.perf_counter() - t1
                note("CPU sample done")
                c_rays = cnt["rays_extend"] + cnt["rays_shadow"]
                b_ray = oracle.algorithmic_bytes_per_ray(cnt)  # extend and shadow rays together, as the launches trace them
                if True:
                    # the same sample once more on the GPU, outside the timed region: live parity
                    pc = p if (sw, sh) == (W, H) else device.PathTracingPass(ctx, sw, sh)
                    if pc is not p:
                        pc.bind_scene(sb)
                    pc.set_profiling(False)
                    pc.set_params(prm)
                    pc.draw()
                    got = pc.download()
                    if pc is not p:
                        pc.destroy()
                    same = all(bool(np.all((got[k].view(np.uint32) == ref[k].view(np.uint32)) | (got[k] == ref[k])
                                           | (np.isnan(got[k]) & np.isnan(ref[k])))) for k in ref)
                    out["parity_vs_oracle"] = ("all 7 outputs bit-equal for sample %d at %dx%d" % (s_last, sw, sh)) if same else "MISMATCH"
                # more samples of the same frame, until about ten seconds of CPU work have been timed (at most 16)
                n_cpu = 1
                while cdt < 10.0 and n_cpu < 16 and s_last - n_cpu >= 1:
                    sx = s_last - n_cpu
                    prm_x = scene.make_params(sw, sh, v2w, sky, spp=1, frame=sx, jitter=jit[sx - 1], max_depth=args.depth,
                                              light_count=len(sd.lights))
                    t1 = time.perf_counter()
                    _, cnt_x = osc.render(prm_x, n_threads=cores)
                    cdt += time.perf_counter() - t1
                    c_rays += cnt_x["rays_extend"] + cnt_x["rays_shadow"]
                    n_cpu += 1
                note("CPU baseline: %d samples in %.1f s" % (n_cpu, cdt))
                out["cpu_baseline"] = {"value": c_rays / cdt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                                       "sample": "%d spp of the same frame at %dx%d (%d rays, %.1f s), OpenMP over rows" % (n_cpu, sw, sh, c_rays, cdt)}
                out["algorithmic_bytes"] = {"per_extend_ray": oracle.algorithmic_bytes_per_extend_ray(cnt),
                                            "per_any_ray": oracle.algorithmic_bytes_per_ray(cnt)}
        # ---- roofline of the dominant kernel
        if dominant is not None and dominant[2] > 0:
            name, ms, launches, rays_in_kernel = dominant
            if b_ray is None:
                b_ray = float(os.environ.get("RT_B_RAY", "0")) or None
            if b_ray is not None:
                achieved = rays_in_kernel * b_ray / (ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_unit": "bytes per launch",
                                   "avg_launch_ms": ms / launches, "launches": int(launches),
                                   "algorithmic_bytes_per_ray": b_ray,
                                   "rays_per_launch": rays_in_kernel / launches}
                if wl != "casa_isect" and world == 1:
                    # The same product over the WHOLE timed region (shading and everything else included,
                    # draws overlapping as they do in use): a lower bound of what the traversal kernels
                    # sustain then; `achieved` above is per launch with one draw in flight, as events can time it.
                    whole = total_rays * b_ray / dt / 1e9
                    out["roofline"]["whole_step_with_draws_in_flight"] = {"achieved": whole, "frac": whole / HBM_PEAK_GBS}
            # What the same kernels physically do, from the committed rocprofv3 --pmc passes of this tree
            # (bench.py cannot collect counters on itself): HBM bytes per launch (FETCH_SIZE + WRITE_SIZE),
            # lanes active per vector instruction, vector-ALU issue occupancy, L2 hit rate.
            from raytracing_amd.build import kernel_sources_digest
            pmc_file = next((f for f in (os.path.join(ROOT, "profiles", "r%02d_pmc_%s.json" % (r, wl)) for r in (3, 2, 1))
                             if os.path.exists(f)), None)
            # (the committed passes are of the default configuration of each workload)
            default_cfg = args.depth == 8 and args.width is None and args.height is None and \
                (wl != "synth_pt" or (args.tris == 10_000_000 and args.dielectric == 0.0 and args.textures == 0)) and \
                (wl != "casa_isect" or args.rays == 1 << 24)
            if "roofline" in out and pmc_file and default_cfg:
                rf = out["roofline"]
                pmc = json.load(open(pmc_file))
                measured_on = pmc.get("_meta", {}).get("kernel_sources_sha256")
                if measured_on != kernel_sources_digest():
                    # counters of other kernels than the ones just timed say nothing about them
                    rf["traffic_note"] = ("%s was measured on a different version of the kernel sources: traffic and the "
                                          "counter-derived fields are left null (tools/run_pmc.sh refreshes it)" % os.path.basename(pmc_file))
                    rf.update({"hbm_frac_measured": None, "lanes_active_per_valu_inst": None, "l2_hit": None, "valu_issue_busy": None})
                else:
                    names = ["intersect_fast_kernel", "intersect_kernel"] if wl == "casa_isect" else \
                            ["path_trace_fast_kernel", "extend_fast_kernel", "extend_kernel", "shadow_fast_kernel", "shadow_kernel"]
                    have = [n for n in names if n in pmc and "FETCH_SIZE" in pmc[n] and "WRITE_SIZE" in pmc[n]]
                    main = [n for n in have if n not in ("extend_kernel", "intersect_kernel")]
                    if have and main:
                        # total bytes of those kernels over the profiled run / number of main launches
                        tot = sum((pmc[n]["FETCH_SIZE"]["total"] + pmc[n]["WRITE_SIZE"]["total"]) * 1024.0 for n in have)
                        n_launch = sum(pmc[n]["FETCH_SIZE"]["launches"] for n in main)
                        # every kernel of a frame's sample (traversal, shading, finalize): bytes that crossed the HBM interface,
                        # from the counter passes of the draws as shipped when there are such (the passes behind `traffic` have
                        # every stage in a launch of its own, as the launches timed here)
                        shipped = pmc_file.replace(".json", "_shipped.json")
                        meta = pmc.get("_meta", {})
                        if os.path.exists(shipped):
                            sm = json.load(open(shipped)).get("_meta", {})
                            if sm.get("kernel_sources_sha256") == kernel_sources_digest() and sm.get("hbm_bytes_per_sample"):
                                meta = sm
                        if meta.get("hbm_bytes_per_sample"):
                            out["hbm_bytes_per_sample_measured"] = meta["hbm_bytes_per_sample"]
                        rf["traffic"] = tot / max(n_launch, 1)
                        rf["traffic_source"] = "profiles/" + os.path.basename(pmc_file)
                        pmc_batch = pmc.get("_meta", {}).get("samples_per_draw")
                        if wl != "casa_isect" and pmc_batch and pmc_batch != max(args.batch, 1):
                            # the counter passes drew other batches than this run: a launch carries the rays of `batch` samples
                            # and its bytes follow (path state and queue entries per ray; the tree stays in the caches)
                            rf["traffic"] *= max(args.batch, 1) / float(pmc_batch)
                            rf["traffic_scaled"] = ("counter passes drew %d samples per launch, this run %d: bytes per launch scaled by "
                                                    "that ratio" % (pmc_batch, max(args.batch, 1)))
                        rf["algorithmic_bytes_per_launch"] = rf["rays_per_launch"] * b_ray
                        # the physical figure: bytes really moved per launch over this run's launch time
                        rf["hbm_gbs_measured"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9
                        rf["hbm_frac_measured"] = rf["hbm_gbs_measured"] / HBM_PEAK_GBS
                        m = pmc[main[0]]

                        def tot_of(c):
                            return sum(pmc[n][c]["total"] for n in main if c in pmc[n])
                        if "SQ_INSTS_VALU" in m and "SQ_THREAD_CYCLES_VALU" in m:
                            rf["lanes_active_per_valu_inst"] = tot_of("SQ_THREAD_CYCLES_VALU") / tot_of("SQ_INSTS_VALU")
                        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
                            rf["l2_hit"] = tot_of("TCC_HIT_sum") / (tot_of("TCC_HIT_sum") + tot_of("TCC_MISS_sum"))
                        if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
                            # wave instructions x 4 cycles over (cycles x 1024 SIMDs); GRBM_GUI_ACTIVE sums the 8 XCDs
                            rf["valu_issue_busy"] = tot_of("SQ_INSTS_VALU") * 4.0 / (tot_of("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)
                        if "SQ_INSTS_SALU" in m and "SQ_INSTS_VALU" in m:
                            rf["salu_per_valu_inst"] = tot_of("SQ_INSTS_SALU") / tot_of("SQ_INSTS_VALU")
                        if rf.get("valu_issue_busy") and rf.get("lanes_active_per_valu_inst"):
                            # the roof this kernel is under: vector lanes.  Share of the chip's lane-cycles (1024 SIMDs x 16 lanes)
                            # that did work = share of cycles a SIMD issued a vector instruction x share of its 64 lanes active
                            rf["valu_lane_frac"] = rf["valu_issue_busy"] * rf["lanes_active_per_valu_inst"] / 64.0
                        if rf["hbm_frac_measured"] < 0.3:
                            rf["bound"] = "valu-issue/latency"
                        rf["note"] = ("frac = the REFERENCE algorithm's bytes per ray (48-byte nodes in its visiting order, counted by "
                                      "the oracle) x rays / kernel time / 8 TB/s; it is not bandwidth: the kernel answers the same queries "
                                      "with far fewer bytes.  hbm_frac_measured = bytes the kernel really moved (PMC) / kernel time / 8 TB/s; "
                                      "below 0.3 the kernel is bound by vector-ALU issue and latency (valu_issue_busy, "
                                      "lanes_active_per_valu_inst of 64; valu_lane_frac = their product / 64 = the share of the chip's "
                                      "vector lane-cycles that did work), not by HBM.")
            if prof:
                out["kernel_ms"] = {k: prof[k] for k in ("extend_ms", "trace_ms", "shadow_ms", "shade_ms", "other_ms", "intersect_ms", "tail_ms")}
                if "draws" in prof:
                    out["kernel_ms"]["over_draws"] = prof["draws"]
                    out["kernel_ms"]["over_samples"] = prof["samples_timed"]
                    # a caller that draws ONE sample, waits, consumes (the reference's loop shape with Config::SPP = 1) ...
                    out["kernel_ms"]["one_draw_in_flight_ms_per_step"] = prof["serial_ms_per_step"]
                    # ... and one that hands the frame's samples over in batches (pt_pass_draw_batch) and waits for each batch
                    out["kernel_ms"]["one_batch_in_flight_ms_per_step"] = prof["batch_ms_per_step"]
                    out["kernel_ms"]["one_draw_in_flight_mean_ms_per_step"] = prof["serial_mean_ms_per_step"]
                    if b_ray is not None:
                        # what the whole pipelined frame sustains, extend rays only
                        out["roofline"]["measured_on"] = "%d samples drawn again after the timed region, in the same batches, one draw in flight, HIP events around every launch" % prof["samples_timed"]
        print(json.dumps(out), flush=True)

    if wl != "casa_isect" and comm is not None:
        comm.destroy()
    p.destroy()
    sb.release()
    ctx.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
