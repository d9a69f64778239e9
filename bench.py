#!/usr/bin/env python3
"""bench.py -- PIMPLE outer-iterations/s of the fireFoam hot path on the synthetic hex box.

One "step" = one fireFoam time step with nOuterCorrectors 1 (= one PIMPLE outer iteration,
solver/fireFoam.C:97-119): rhoEqn, UEqn (3 PBiCGStab+DILU solves), YEEqn (4 species + h), 2 x pEqn
(PCG+DIC, relTol 0.01 then 0), assembly included, on the synthetic buoyant-plume box of SURVEY 8(d)
(ffm_plume_* in include/ffm.h).  Inputs are resident in HBM before the timed region.  The fvDOM stand-in runs at the
reference case's solverFreq 100, i.e. on step 0 (a warm-up step by default); its cost is measured separately and reported
in config.radiation (sweep_ms, amortised_ms_per_step) -- it is not part of `value` unless a timed step is a multiple of 100.

    python bench.py --gpus 1 --steps 3 --warmup 1            # 400^3 = 64 M cells on one MI355X
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the driver contract fields plus `roofline` (the pEqn PCG SpMV
kernel against the 8 TB/s HBM peak, algorithmic bytes 24 N + 16 F; `traffic` = the PMC figure of
profiles/spmv_traffic.json, quoted for the configuration it was measured on) and `cpu_baseline` (the
oracle port on one host core, bounded sample, scaled by cell count).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--edge", dest="n", type=int, default=400, help="cells per box edge (400 -> 64 M cells, BASELINE.json config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=96)
    ap.add_argument("--cpu-procs", type=int, default=0, help="replicas of the CPU baseline run side by side (0: all cores, at most 16)")
    ap.add_argument("--solvers", choices=["krylov", "steckler"], default="krylov",
                    help="transport equations: PBiCGStab+DILU (default) or smoothSolver+symGaussSeidel maxIter 10 (cases/steckler/system/fvSolution:49-62)")
    ap.add_argument("--radiation-freq", type=int, default=100,
                    help="fvDOM stand-in (32 upwind ray solves) every N steps, counted from step 0; 100 = the reference case "
                         "(cases/steckler/constant/radiationProperties:38); 0 = off")
    ap.add_argument("--no-class-layer", action="store_true", help="skip the extra measurement of the same case through the reference's unchanged equation files")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="rccl: one rank per GPU over xGMI (production); host: ranks share GPUs, halo through gloo (rehearsal)")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: whatever a library prints there while the run is set up (gloo's "[Gloo] Rank 0 is
    # connected to ..." on the host transport) goes to stderr; the line itself is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    requested_transport = args.transport
    dev = local % max(ndev, 1)
    torch.cuda.set_device(dev)
    if world > 1 and args.transport == "rccl" and world > max(ndev, 1):
        # more ranks than devices (a rehearsal on a smaller box): RCCL refuses two ranks on one device, use the host transport
        sys.stderr.write("bench.py: %d ranks on %d device(s): falling back to --transport host\n" % (world, ndev))
        args.transport = "host"
    if world > 1:
        if args.transport == "rccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    from ffm_import import ffm
    ctx = ffm.Context(dev)
    n = args.n
    t0 = time.time()
    if world == 1:
        case = ffm.Plume(ctx, (n, n, n), h=0.05, deltaT=1e-3)
        grid = (1, 1, 1)
    else:
        # strong scaling: the same global box, block-decomposed (2x1x1, 2x2x1, 2x2x2), one block per rank
        transport = args.transport
        if transport == "rccl":
            ids = [ffm.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            ok = 1
            try:
                ctx.comm_init_rccl(rank, world, ids[0])
            except Exception as e:                       # keep the run alive: every rank falls back to the host transport
                ok = 0
                sys.stderr.write("bench.py: rank %d: RCCL communicator failed (%s)\n" % (rank, e))
            t = torch.tensor([ok], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                transport = "host"
                gloo_comm = ffm.gloo_comm
                gloo_comm.GROUP = dist.new_group(backend="gloo")
                ctx.comm_init_host(rank, world, gloo_comm.allreduce, gloo_comm.exchange)
        else:
            gloo_comm = ffm.gloo_comm
            ctx.comm_init_host(rank, world, gloo_comm.allreduce, gloo_comm.exchange)
        grid = ffm.hexmesh.grid_for(world)
        lo, hi, nbr = ffm.hexmesh.block_of_rank((n, n, n), grid, rank)
        case = ffm.Plume(ctx, (n, n, n), h=0.05, deltaT=1e-3, lo=lo, hi=hi, nbrRank=nbr)
    if args.solvers == "steckler":
        case.set_solvers(steckler=True)
    if args.radiation_freq > 0:
        case.set_radiation(solverFreq=args.radiation_freq)
    setup_s = time.time() - t0
    # the start state of the case, kept on the host for the class-layer measurement at the end (the SAME case through the reference's
    # unchanged solver/*.H over include/ffmFoam.H); nothing of it is touched inside the timed region
    snap = None
    if rank == 0 and world == 1 and not args.no_class_layer and ffm.snippets.load() is not None:
        try:
            snap = ffm.snippets.FromPlume(case)
        except Exception as e:
            sys.stderr.write("bench.py: class-layer snapshot failed (%r)\n" % (e,))
    # machine-readable: what carried the halo exchange and the reductions of this run ("none": one rank), and whether that is
    # what was asked for -- an RCCL -> host degradation must not pass for a scaling number
    used_transport = "none" if world == 1 else transport
    N, F = n ** 3, 3 * n * n * (n - 1)          # global cells / faces
    Nloc, Floc = case.nCells, case.nFaces

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        case.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        case.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if args.transport == "rccl" else "cpu")      # (the default group: nccl or gloo)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    solves = case.solves()
    pIters = [pf["nIterations"] for nm, pf in solves if nm == "p_rgh"]

    # ---- roofline: the PCG SpMV kernel (lduMatrix::Amul) on the p_rgh matrix left by the last corrector,
    # HIP events on the library's stream (ffm_bench_spmv), algorithmic bytes 24 N + 16 F
    import ctypes as C
    nExt = L_ncells = ffm.lib().ffm_ldu_ncells(case.ldu_handle())
    x = ctx.to_device(ffm.hexmesh.hash_u(0xF4, __import__("numpy").arange(nExt)))
    y = ctx.empty(nExt)
    ms = C.c_double()
    L = ffm.lib()
    rc = L.ffm_bench_spmv(case.ldu_handle(), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 20, C.byref(ms))
    if rc:
        raise SystemExit("ffm_bench_spmv failed: %s" % L.ffm_last_error().decode())
    alg_bytes = 24 * Nloc + 16 * Floc           # this rank's rows (N > 1: the timing then includes the halo refresh)
    achieved = alg_bytes / (ms.value * 1e-3) / 1e9
    # L2-side traffic per launch from the committed PMC passes (profiles/spmv_traffic.json); only quoted for the configuration
    # it was measured on (rocprofv3 cannot wrap the bench's own timed region)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json")))
        if tj["edge"] == n and tj["n_gpus"] == world:
            traffic = tj["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "k_tile_amul (lduMatrix::Amul, symmetric p_rgh matrix, tile numbering%s)" % ("" if world == 1 else "; + ghost refresh and ghost-face tail"),
                "achieved": round(achieved, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": round(ms.value, 5)}

    # ---- the other half of a PCG iteration, for the record: one DIC application = forward + backward sweep over the same matrix
    # (not the metric kernel; algorithmic bytes per sweep 16 F + 16 N as SURVEY 8d counts the unfused upstream loops, + w = rD*r 24 N)
    ms2 = C.c_double()
    rc = L.ffm_bench_precond(case.ldu_handle(), 1, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 10, C.byref(ms2))
    sweeps = None
    if rc == 0:
        dic_bytes = 2 * (16 * Floc + 16 * Nloc) + 24 * Nloc
        sweeps = {"kernel": "k_tile<FWD> + k_tile<BWD> (DICPreconditioner::precondition, exact dependency order)", "avg_ms_per_application": round(ms2.value, 4),
                  "algorithmic_bytes_per_application": dic_bytes, "achieved": round(dic_bytes / (ms2.value * 1e-3) / 1e9, 1), "unit": "GB/s",
                  "frac": round(dic_bytes / (ms2.value * 1e-3) / 1e9 / 8000.0, 4), "bound": "dependency levels x per-level latency below ~20 M cells, hbm above"}
    # ---- cost of one fvDOM sweep (it falls on steps 0, 100, 200, ...: with --warmup >= 1 outside the timed steps)
    radiation = "off"
    if args.radiation_freq > 0:
        case.set_radiation(solverFreq=1)
        barrier()
        t1 = time.perf_counter()
        case.step()
        barrier()
        sweep_ms = max((time.perf_counter() - t1) * 1e3 - dt / args.steps * 1e3, 0.0)
        rays = [pf["nIterations"] for nm, pf in case.solves() if nm.startswith("I") and nm[1:].isdigit()]
        radiation = {"model": "fvDOM stand-in: 32 rays (nPhi 2, nTheta 4), upwind, PBiCGStab+DILU to 1e-4 in a direction-ordered cell numbering per ray (one block; iterative on decomposed blocks), constant absorption, no coupling into h",
                     "solverFreq": args.radiation_freq,
                     "timed_steps_containing_a_sweep": len([k for k in range(args.warmup, args.warmup + args.steps) if k % args.radiation_freq == 0]),
                     "sweep_ms": round(sweep_ms, 1), "amortised_ms_per_step": round(sweep_ms / args.radiation_freq, 2),
                     "ray_iterations_mean": round(sum(rays) / max(len(rays), 1), 1)}

    # ---- the drop-in path: the same case advanced by the reference's own equation files (solver/rhoEqn.H, UEqn.H, YEEqn.H, pEqn.H,
    # unchanged) over the Foam layer -- one kernel and one temporary per operator -- on the same matrix and mesh, from the same start
    # state; an extra key, never `value`
    class_layer = None
    if snap is not None:
        try:
            slib = ffm.snippets.load()
            os.environ["FFM_FOAM_QUIET"] = "1"
            solver = slib.firefoam_snippets_create(ctx.h, case.ldu_handle(), case.mesh().h, C.byref(snap.cs))
            slib.firefoam_snippets_advance(solver, C.byref(snap.cs), 0)          # warm-up (first use of every temporary size)
            barrier()
            t1 = time.perf_counter()
            ncl = 2
            for _ in range(ncl):
                nsol = slib.firefoam_snippets_advance(solver, C.byref(snap.cs), 0)
            barrier()
            cl_ms = (time.perf_counter() - t1) / ncl * 1e3
            its = list(snap.nit[:nsol])
            slib.firefoam_snippets_destroy(solver)
            class_layer = {"path": "reference solver/{rhoEqn,UEqn,YEEqn,pEqn}.H unchanged over include/ffmFoam.H (libffm_refsnippets.so), same case, same mesh and matrix, start state of the compiled run",
                           "ms_per_step": round(cl_ms, 2), "steps": ncl, "warmup": 1, "ratio_to_compiled": round(cl_ms / (dt / args.steps * 1e3), 3),
                           "p_rgh_iterations_last_step": its[-2:]}
        except Exception as e:
            class_layer = {"error": repr(e)}
        snap = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import plume as oplume       # test infrastructure, used here only as the timed CPU baseline
        cn = args.cpu_n
        ref = oplume.Plume((cn, cn, cn))
        ref.step()
        t1 = time.perf_counter()
        nst = 2
        for _ in range(nst):
            ref.step()
        cdt = (time.perf_counter() - t1) / nst
        cell_steps = cn ** 3 / cdt
        del ref
        # all host cores of this GPU's share: P single-core replicas of the same sample side by side (no halo exchange, so an
        # upper bound for a domain-decomposed CPU run of the reference; they do contend for the memory bandwidth, as its ranks would)
        import subprocess
        P = max(1, min(args.cpu_procs if args.cpu_procs > 0 else (os.cpu_count() or 1), 16))
        code = ("import sys, time; sys.path.insert(0, %r); from oracle import plume as P_\n"
                "c = P_.Plume((%d, %d, %d)); c.step(); t = time.perf_counter(); c.step(); print(time.perf_counter() - t)" % (os.path.dirname(os.path.abspath(__file__)), cn, cn, cn))
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(P)]
        times = []
        for pr in procs:
            o, _ = pr.communicate()
            try:
                times.append(float(o.strip().splitlines()[-1]))
            except Exception:
                pass
        agg = sum(cn ** 3 / t for t in times) if times else cell_steps
        cpu = {"value": round(agg / N, 6), "unit": "outer-iterations/s", "cores": max(len(times), 1), "kind": "port",
               "sample": "oracle/plume.py (numpy assembly + C solvers): %d single-core replicas side by side, each %d^3 cells, 1 warm-up + 1 timed step, "
                         "%.2f s/step on average = %.3g cell-steps/s in aggregate, scaled by cell count to %d cells" % (len(times), cn, sum(times) / max(len(times), 1), agg, N),
               "one_core": {"value": round(cell_steps / N, 6), "s_per_step": round(cdt, 3), "cells": cn ** 3, "steps": nst}}
        # the linear-algebra kernels of the path on the host: Amul GB/s on one core, and the synthetic p_rgh PCG
        # solve on one core against P block subdomains on P threads (block-Jacobi DIC, oracle/ffo_multi.c)
        try:
            import numpy as np
            from oracle import oracle as OO, multi
            H = ffm.hexmesh
            sn = 160
            blk = H.HexBlock((sn, sn, sn)); sy = H.synth_p_rgh(blk)
            Ao = OO.Ldu(blk.nCells, blk.l, blk.u).set_coeffs(sy["diag"], sy["upper"])
            xs = H.hash_u(0xF4, np.arange(blk.nCells))
            Ao.amul(xs)
            t1 = time.perf_counter(); reps = 5
            for _ in range(reps):
                Ao.amul(xs)
            tsp = (time.perf_counter() - t1) / reps
            cpu["spmv_GBps_1core"] = round((24 * blk.nCells + 16 * len(blk.l)) / tsp / 1e9, 2)
            t1 = time.perf_counter()
            _, pf1 = Ao.solve(OO.PCG, OO.DIC, np.zeros(blk.nCells), sy["source"], tolerance=1e-6, relTol=0.0)      # p_rghFinal controls
            t1core = time.perf_counter() - t1
            # P = the largest power of two <= the host's cores (at most 64 blocks: 160^3 in 40^3 blocks), one thread per block
            hc = os.cpu_count() or 1
            P = max(p for p in (2, 4, 8, 16, 32, 64) if p <= max(hc, 2))
            grid_c = {64: (4, 4, 4), 32: (4, 4, 2), 16: (4, 2, 2), 8: (2, 2, 2), 4: (2, 2, 1), 2: (2, 1, 1)}[P]
            blocks, nbrRank, nbrPatch, ldus, srcs = multi.decomposed_case(H, (sn, sn, sn), grid_c)
            t1 = time.perf_counter()
            _, pfP = OO.solve_multi(ldus, nbrRank, nbrPatch, OO.PCG, OO.DIC, [np.zeros(b.nCells) for b in blocks], srcs, tolerance=1e-6, relTol=0.0)
            tP = time.perf_counter() - t1
            cpu["pcg_p_rgh_%d3" % sn] = {"one_core_s": round(t1core, 3), "iterations": pf1["nIterations"], "threads": P,
                                         "threads_s": round(tP, 3), "threads_iterations": pfP[0]["nIterations"],
                                         "speedup": round(t1core / tP, 2), "host_cores": os.cpu_count()}
        except Exception as e:                            # the extra figures are informative; the contract fields above stand
            cpu["extra_error"] = repr(e)

    if rank == 0:
        out = {
            "metric": "PIMPLE outer-iterations/sec on 64M-cell hex mesh; pEqn SpMV GB/s vs HBM peak",
            "value": round(args.steps / dt, 4), "unit": "outer-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic %d^3 hex box (%d cells), buoyant plume + 5-species EDC-shaped source, "
                                   "PIMPLE 1/2/0: rhoEqn + UEqn + YEEqn(4 Yi + h) + 2 pEqn per step" % (n, N),
                       "cells": N, "faces": F, "deltaT": 1e-3, "parallelism": "1 GPU" if world == 1 else ("%dx%dx%d block decomposition, one block per GPU, " % grid)
                                      + ("RCCL halo + all-reduce" if transport == "rccl" else "halo + all-reduce through the host (gloo)"),
                       "radiation": radiation, "p_rgh_iterations_last_step": pIters, "setup_s": round(setup_s, 1),
                       "transport_solvers": "PBiCGStab+DILU" if args.solvers == "krylov" else "smoothSolver+symGaussSeidel maxIter 10"},
            "transport": used_transport, "transport_requested": "none" if world == 1 else requested_transport,
            "transport_degraded": bool(world > 1 and used_transport != requested_transport),
            "roofline": roofline, "roofline_dic_sweeps": sweeps, "class_layer": class_layer, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    case.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
