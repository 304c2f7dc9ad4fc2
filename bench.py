#!/usr/bin/env python3
"""bench.py — frames/s @1920x1080 + sorted Gsplats/s on the garden-sized scene, 1..8 MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one
rank per GPU with torch.distributed.run.  A "step" is one frame = one pass of the hot path
(key+cull+project -> radix sort -> tile binning -> per-pixel compositing) over the resident scene, on
the next pose of the fixed 64-pose orbit.  Inputs are resident in HBM before the timed
region.  W untimed warm-up frames, then EXACTLY K frames between barrier+synchronize pairs; rank 0
prints ONE JSON line.

N>1 (replicated splat buffers): `value` is the north_star's partition — every rank renders its tile-row strip of
the SAME frame with the whole path and the strips are exchanged in place with RCCL (libmgs: one grouped collective per
frame on the render stream, mgs_render_gathered; torch.distributed all_gather as the fallback) so that every rank ends
up with the whole frame: strong scaling of a sub-millisecond frame ("scaling": "strong").  Strip boundaries are
cost-balanced from the per-row list lengths of untimed calibration frames (--equal-strips turns that
off).  The throughput partition (alternate frames: rank r renders poses r, r+N, ... with no collective) is measured in
the same run and reported in the secondary `alternate_frames` object.

The JSON line also carries
  roofline      — the LONGEST stage of the frame by measured single-stream time: SURVEY.md 8d's algorithmic bytes / its HIP-event
                  duration, its PMC HBM bytes (`traffic`), and — when that stage is the compositor, which is not HBM-bound — its VALU
                  issue fraction with the calibration caveat (`valu`); roofline_project, roofline_sort, roofline_bin, roofline_frame
                  (bytes actually moved) and roofline_composite (VALU) beside it
  value_single_frame — frames/s with one frame in flight (value: --inflight frames, default 3)
  cpu_baseline  — the oracle's restatement of the reference's CPU sorter (splat_sorter_async.cpp:92-141),
                  timed on this box's host cores (rank 0, N=1 only); a baseline, not a target.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _first_profile(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return n
    return names[0]


# committed rocprofv3 --pmc passes of this command (the newest round's that exists)
PMC_SQ_FILE = _first_profile("r6_z_pmc_sq_composite.json", "r5_z_pmc_sq_composite.json")  # {"k_composite": {"SQ_INSTS_VALU": per-launch mean, ...}}
PMC_FILE = _first_profile("r6_z_pmc_hbm_traffic.json", "r5_z_pmc_hbm_traffic.json")  # written by tools/pmc_traffic.py
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
STAGES = ["project", "sort", "bin", "pairsort", "composite", "total"]


def composite_roofline(ms, alg_bytes, world, N, args):
    """k_composite is bound by fp32 VALU issue, not by HBM.  Its fraction is the VALU-busy fraction of the SIMDs over the
    kernel's duration: SQ_ACTIVE_INST_VALU (quad-cycles, summed over the chip; committed rocprofv3 --pmc pass of this
    command) x 4 / (launch duration x 2.4 GHz x 1024 SIMDs) — a counter ratio, no assumed issue rate.  Calibrated in round 5
    (tools/micro/valu_rate.hip, profiles/r5_valu_rate.log): the counter advances 1.00 quad-cycle per plain wave64 VALU instruction
    and 2.00 per transcendental, and such an instruction really holds its SIMD for four cycles; 2.4 GHz is the nominal clock —
    under a VALU-dense load the chip ran at 2.13 GHz (GRBM_GUI_ACTIVE), so the fraction at the real clock is up to 1.13x this."""
    out = {"bound": "valu", "unit": "fraction of SIMD cycles issuing VALU", "peak": 1.0, "launch_ms": float(ms), "achieved": None,
           "frac": None, "hbm_algorithmic_bytes_per_launch": float(alg_bytes),
           "hbm_achieved_GBps": (alg_bytes / (ms * 1e-3)) / 1e9 if ms > 0 else None}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", PMC_SQ_FILE)))
        if (world == 1 and N == 5_830_000 and args.instances == 1 and ms > 0 and (args.width, args.height) == (1920, 1080)
                and args.scene == "garden" and not args.alpha_sum and args.pipeline == 0 and not args.ply and not args.stochastic):
            busy = pj["k_composite"]["SQ_ACTIVE_INST_VALU"] * 4.0
            out["valu_busy_quad_cycles_per_launch"] = pj["k_composite"]["SQ_ACTIVE_INST_VALU"]
            out["valu_insts_per_launch"] = pj["k_composite"]["SQ_INSTS_VALU"]
            out["achieved"] = busy / (ms * 1e-3 * 2.4e9 * 1024)
            out["frac"] = out["achieved"]
            out["clock_note"] = ("`frac` is issue-slot occupancy under THIS BUILD'S calibration: SQ_ACTIVE_INST_VALU advances one quad-cycle per plain "
                                 "wave64 VALU instruction and such an instruction holds its SIMD ~4 cycles (tools/micro/valu_rate.hip, profiles/r5_valu_rate.log: "
                                 "v_fma_f32 540 wave-instr/SIMD/us = 70.8 TFLOP/s chip-wide, v_pk_fma_f32 428 = 112 TFLOP/s = 0.71 of the 157.3 TFLOP/s fp32 "
                                 "vector spec; v_mov 2.8, v_fma 4.0, v_pk_fma 5.0 real cycles: 4 is an average, not a law).  MI355X_MICROARCH.md's table lists "
                                 "2 cycles for v_fma_f32 (wave64): under that reading the fraction is `frac_if_2_cycles`.  2.4 GHz nominal; at the 2.13 GHz the "
                                 "chip sustains under VALU-dense load the busy fraction is 1.13x `frac`")
            out["frac_if_2_cycles"] = out["frac"] * 0.5
            out["valu_source"] = f"profiles/{PMC_SQ_FILE} (committed rocprofv3 --pmc pass of `bench.py --inflight 1`; the duration is this run's)"
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--splats", type=int, default=5_830_000, help="scene size (default: garden-sized, configs[2])")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--instances", type=int, default=1, help="instances of the scene on a 2 x ceil(k/2) grid, spacing 12 "
                    "(configs[4]: 8 -> ~46.6 M splats under one unified depth sort)")
    ap.add_argument("--sh-format", type=int, default=0, help="0 fp32 (benchmark setting), 1 fp16, 2 uint8")
    ap.add_argument("--rgba-format", type=int, default=0)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="shorthand for the BASELINE.json configs: 1 train-sized 1080p, 2 garden-sized 1080p (the default), "
                         "3 garden-sized 3840x2160, 4 eight garden instances (46.6 M splats) 1080p")
    ap.add_argument("--ply", default=None, help="render this .ply / .spz / .splat (e.g. the real Mip-NeRF360 garden.ply) instead of the "
                    "synthetic stand-in; same code path, `data` says so")
    ap.add_argument("--scene", default="garden", choices=["garden", "fog", "sparse"],
                    help="garden: syn_garden (SURVEY.md 8d, the benchmark workload); fog: the same splats with opacity logits "
                         "shifted to mean -3 (low opacity: regions saturate late or never, the compositor's hard regime); "
                         "sparse: 10 %% of the object splats + the whole background")
    ap.add_argument("--alpha-sum", action="store_true", help="MGS_ALPHA_SUM: the reference's default additive alpha "
                    "(gaussian_splatting.cpp:2083-2084); disables early termination")
    ap.add_argument("--pipeline", type=int, default=0, help="0 3DGS (benchmark), 1 3DGUT")
    ap.add_argument("--stochastic", action="store_true", help="MGS_SORT_STOCHASTIC (stochastic splats, a new frame_sample_id per frame) "
                    "instead of the sorted alpha blend; a secondary number, never the headline")
    ap.add_argument("--dof", type=float, default=0.0, help="3DGUT only: depth of field with this aperture (focus distance 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras of the default run (value_reference_alpha, "
                    "value_uint8_storage): profiling passes want the headline frames only")
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight per GPU (each on its own HIP stream with "
                    "its own working buffers); >1 overlaps one frame's tails/launch gaps/all-gather with the next frame")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (functional check of the N>1 path "
                    "when all ranks share one GPU; strips are staged through host memory)")
    ap.add_argument("--stage-events", action="store_true", help="record the six per-stage HIP events inside the timed frames too "
                    "(plain launches); default: the timed frames are replayed hipGraphs and the per-stage times come from the "
                    "single-stream calibration frames")
    ap.add_argument("--check-gather", action="store_true", help="N>1: verify the gathered frame == a full-frame render")
    ap.add_argument("--equal-strips", action="store_true", help="N>1: equal tile-row strips instead of cost-balanced ones")
    ap.add_argument("--strip-bounds", default=None, help="N>1: the strip table itself, N+1 ascending tile rows \"0,r1,...,rows\" "
                    "(equal neighbours = an empty strip): exercises ragged partitions of the exchange")
    ap.add_argument("--torch-gather", action="store_true", help="N>1: exchange the strips with torch.distributed instead of "
                    "libmgs's own RCCL call (the fallback path)")
    ap.add_argument("--libmgs-gather", action="store_true", help="N>1 with --backend gloo: still exchange the strips through libmgs's own "
                    "RCCL calls (mgs_scene_comm_init / mgs_render_gathered).  With all ranks on one GPU that needs MGS_RCCL_LIB to name the "
                    "test double tests/helpers/libfakerccl.so (real RCCL refuses two ranks on one device)")
    args = ap.parse_args()
    if args.config == 1:
        args.splats = 1_030_000
    elif args.config == 3:
        args.width, args.height = 3840, 2160
    elif args.config == 4:
        args.instances = 8

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=300))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)   # gloo functional check: several ranks may share one GPU
    torch.cuda.set_device(local)

    import vk_gaussian_splatting_amd as mgs
    from vk_gaussian_splatting_amd import capi, synth, multigpu

    W, H, N = args.width, args.height, args.splats
    t0 = time.time()
    if args.ply:
        ss = mgs.SplatSet.load(args.ply)
        a = ss.arrays()
        N = args.splats = int(a["count"])
        sc = {"positions": a["positions"].reshape(-1, 3)}  # the CPU baseline needs the centres only
    else:
        sc = synth.make_scene(N, seed=0xC0FFEE + 2)  # syn_garden; identical on every rank
    if args.ply:
        pass
    elif args.scene == "fog":
        sc["opacity"] = (sc["opacity"] - 2.5).astype(np.float32)  # logit mean -0.5 -> -3
    elif args.scene == "sparse":
        r = np.linalg.norm(sc["positions"], axis=1)
        keep = (r >= 4.0) | (np.random.default_rng(7).random(N) < 0.10)
        sc = {k: np.ascontiguousarray(v[keep]) for k, v in sc.items()}
        N = args.splats = int(keep.sum())
    if not args.ply:
        ss = mgs.SplatSet.from_arrays(**sc)
    K = max(1, args.inflight)
    # ONE resident scene, committed once; frames in flight are frame contexts over it (own stream + working set + graphs),
    # like the reference's single copy of the splat buffers under its frames in flight (gaussian_splatting.cpp:1092-1111)
    scenes, streams = [], []
    scene = mgs.Scene(local)
    for q in range(args.instances):
        if args.instances == 1:
            scene.add_instance(ss)
        else:
            cols = (args.instances + 1) // 2
            M = np.eye(4, dtype=np.float32)
            M[0, 3] = ((q % cols) - (cols - 1) / 2.0) * 12.0
            M[2, 3] = ((q // cols) - 0.5) * 12.0
            scene.add_instance(ss, M)
    scene.commit(args.sh_format, args.rgba_format)
    for c in range(K):
        sc_k = scene if c == 0 else scene.frame_context()
        st_k = torch.cuda.Stream()        # a real (non-null) stream shared by this context's renderer and RCCL (equal priorities: a high-priority context
                                          # starves the others — 3 800 -> 3 630 frames/s, profiles/r6_q_prio.log)
        sc_k.set_stream(st_k.cuda_stream)
        scenes.append(sc_k)
        streams.append(st_k)
    setup_s = time.time() - t0

    poses = []
    for i in range(64):
        eye = synth.orbit_pose(i)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
        p = capi.default_params(W, H)
        capi.set_camera(p, V, P, eye)
        p.collect_timings = 2 if args.stage_events else 0
        p.alpha_mode = capi.ALPHA_SUM if args.alpha_sum else capi.ALPHA_COVERAGE
        p.pipeline = args.pipeline
        if args.stochastic:
            p.sort_mode = capi.SORT_STOCHASTIC
            p.frame_sample_id = i
        if args.dof > 0.0:
            p.dof_mode, p.focus_dist, p.aperture, p.frame_sample_id = capi.DOF_FIXED_FOCUS, 3.0, args.dof, i
        poses.append(p)
    tiles_y = multigpu.tile_rows(H)
    bounds = [multigpu.strip_rows(H, world, r)[0] for r in range(world)] + [tiles_y]  # equal strips
    gather_mode = "none"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def frame(i):
        """alternate-frame / single-GPU step: this rank's i-th whole frame (pose i*world + rank of the orbit)"""
        c = i % K
        with torch.cuda.stream(streams[c]):
            scenes[c].render(poses[(i * world + rank) % 64])

    # ---- N>1: communicators + strip table --------------------------------------------------------------------
    if world > 1:
        if (args.backend == "nccl" or args.libmgs_gather) and not args.torch_gather:
            # libmgs's own RCCL exchange: one communicator per frame context (each has its own stream).  Rank 0 makes the
            # ids; whatever happens there, every rank takes part in the broadcast (nobody is left waiting in it)
            ids = [None] * K
            if rank == 0:
                try:
                    ids = [capi.comm_unique_id() for _ in range(K)]
                except Exception as e:  # noqa: BLE001
                    print(f"[rank 0] mgs_comm_unique_id failed ({type(e).__name__}: {e})", file=sys.stderr)
                    ids = [None] * K
            dist.broadcast_object_list(ids, src=0)
            try:
                if any(i is None for i in ids):
                    raise RuntimeError("no RCCL unique id from rank 0")
                for c in range(K):
                    scenes[c].comm_init(rank, world, ids[c])
                gather_mode = "libmgs: grouped ncclBroadcast of every rank's rows, in place in the frame buffer, on the render stream"
                if os.environ.get("MGS_RCCL_LIB"):
                    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "rccl" in ln.lower() and ".so" in ln]
                    gather_mode += f" [MGS_RCCL_LIB: {os.path.basename(os.environ['MGS_RCCL_LIB'])}, mapped: {sorted(set(os.path.basename(x) for x in loaded))}]"
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] libmgs RCCL exchange unavailable ({type(e).__name__}: {e}); falling back to torch.distributed",
                      file=sys.stderr)
                gather_mode = "none"
        # every rank agrees on the outcome (a rank that failed must not leave the others waiting in a collective)
        flag = torch.tensor([1 if gather_mode.startswith("libmgs") else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if gather_mode.startswith("libmgs"):
                for c in range(K):
                    scenes[c].comm_destroy()
            gather_mode = ("torch.distributed all_gather_into_tensor (RCCL)" if args.backend == "nccl"
                           else f"torch.distributed all_gather over {args.backend}, strips staged through host memory (functional check)")
        if args.strip_bounds:
            bounds = [int(x) for x in args.strip_bounds.split(",")]
            assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == tiles_y and all(a <= b for a, b in zip(bounds, bounds[1:])), \
                f"--strip-bounds needs {world + 1} ascending tile rows from 0 to {tiles_y}"
        elif not args.equal_strips:
            # cost-balanced strips: per-tile-row list lengths of untimed full frames (every rank renders the same
            # frames, so every rank derives the same table; no exchange needed)
            cost = np.zeros(tiles_y, np.float64)
            for i in range(0, 64, 8):
                scenes[0].render(poses[i])
                cost += scenes[0].row_costs(H)
            bounds = multigpu.balanced_bounds(cost, world)
        if gather_mode.startswith("libmgs"):
            for c in range(K):
                scenes[c].set_strip_rows(bounds)
    my_rows = (bounds[rank], bounds[rank + 1]) if world > 1 else (0, 0)
    # what the untimed calibration / counter frames of this rank render: its strip — or, for a rank whose strip is EMPTY (it renders
    # nothing in the timed frames and only joins the exchange), the whole frame, so that its stage times and counters exist at all
    own_rows = my_rows if my_rows[1] > my_rows[0] else (0, 0)
    Rmax = multigpu.padded_strip_rows(bounds) if world > 1 else 0
    strips = [torch.zeros((Rmax, W, 4), dtype=torch.float16, device="cuda") if world > 1 and not gather_mode.startswith("libmgs")
              else None for _ in range(K)]
    gathered = [None] * K

    def strip_frame(i):
        """strip step: this rank's tile rows of frame i + the exchange; afterwards every rank holds the whole frame"""
        p = poses[i % 64]
        c = i % K
        with torch.cuda.stream(streams[c]):
            if gather_mode.startswith("libmgs"):
                scenes[c].render_gathered(p)
                return
            p.strip_row_begin, p.strip_row_end = my_rows
            if my_rows[1] > my_rows[0]:
                scenes[c].render(p)
                scenes[c].copy_strip(strips[c].data_ptr(), (min(my_rows[1] * 16, H) - my_rows[0] * 16) * W * 8)
            p.strip_row_begin, p.strip_row_end = 0, 0
            src = strips[c].cpu() if args.backend != "nccl" else strips[c]
            gathered[c] = multigpu.gather_strips(src, world)  # padded to the tallest strip

    def assembled(c):
        """the whole frame on this rank after strip_frame (host tensor, int16 view of the fp16 pixels)"""
        if gather_mode.startswith("libmgs"):
            pf = capi.default_params(W, H)
            return torch.from_numpy(scenes[c].download_frame(pf).view(np.int16))
        return multigpu.assemble(gathered[c].cpu(), bounds, H).view(torch.int16)

    step = strip_frame if world > 1 else frame
    for i in range(args.warmup):
        step(i)
    fence()
    # calibration (untimed, single stream): which stage dominates a frame when nothing else shares the GPU
    calib = []
    for i in range(8):
        pc = poses[i % 64]
        pc.collect_timings = 2
        pc.strip_row_begin, pc.strip_row_end = own_rows
        with torch.cuda.stream(streams[0]):
            scenes[0].render(pc)
        torch.cuda.synchronize()
        calib.append(scenes[0].timings_all(0))
        pc.collect_timings = 2 if args.stage_events else 0
        pc.strip_row_begin, pc.strip_row_end = 0, 0
    calib_ms = np.array(calib[2:], np.float64).mean(axis=0)
    fence()
    # ... and the serial frame as a consumer gets it: graph-replayed frames one behind the other on ONE stream, HIP events around 24
    # of them (untimed region; the calibration frames above carry six stage events and plain launches each: ~8 us more per frame)
    serial_ms = None
    if world == 1 and not args.stage_events:
        try:
            with torch.cuda.stream(streams[0]):
                for i in range(4):
                    scenes[0].render(poses[i % 64])
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(streams[0])
                for i in range(24):
                    scenes[0].render(poses[(8 + i) % 64])
                ev1.record(streams[0])
            torch.cuda.synchronize()
            serial_ms = ev0.elapsed_time(ev1) / 24.0
        except Exception:  # noqa: BLE001
            serial_ms = None
        fence()
    if world > 1 and args.check_gather:
        strip_frame(0)
        torch.cuda.synchronize()
        got = assembled(0)
        pf = capi.default_params(W, H)
        for kk in range(16):
            pf.view[kk], pf.proj[kk] = poses[0].view[kk], poses[0].proj[kk]
        for kk in range(3):
            pf.camera_pos[kk] = poses[0].camera_pos[kk]
        scenes[K - 1].render(pf) if K > 1 else None
        ref_scene = scenes[K - 1] if K > 1 else mgs.Scene(local)
        if K == 1:
            ref_scene.add_instance(ss)
            ref_scene.commit(args.sh_format, args.rgba_format)
            ref_scene.render(pf)
        full = torch.from_numpy(ref_scene.download_frame(pf).view(np.int16))
        same = torch.equal(got, full)
        print(f"[rank {rank}] gathered frame == full frame: {same}  (strip rows {my_rows}, table {bounds}, {gather_mode})", file=sys.stderr)
        assert same, "the strip exchange does not reproduce the single-GPU frame"
        if K == 1:
            ref_scene.close()
        fence()
    # Pre-roll (untimed, like the warm-up): the calibration frames above are synchronous — the GPU idles between them and falls
    # out of its sustained clocks, and a timed region of 20 frames (5 ms) that starts there reads 6-9 % low (round 6:
    # 3 430-3 560 frames/s against 3 720-3 870 at 128 steps; profiles/r6_p_warmup.log).  MGS_BENCH_PREROLL (default 128) dense frames through the
    # same step() as the timed ones put the device where a running renderer has it; then the fence, then the K timed frames.
    preroll = int(os.environ.get("MGS_BENCH_PREROLL", "128"))
    for i in range(preroll):
        step(args.warmup + i)
    fence()
    # one event per frame end (on the frame's own stream): the intervals between consecutive completions give the
    # p50 / p95 of the frame time as a consumer sees it; 1 us of host work per frame
    done_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
        done_ev[i].record(streams[(args.warmup + i) % K])
    fence()
    elapsed = time.perf_counter() - t1
    intervals = np.array([done_ev[i].elapsed_time(done_ev[i + 1]) for i in range(args.steps - 1)], np.float64) if args.steps > 2 else np.zeros(1)
    if os.environ.get("MGS_BENCH_DUMP_INTERVALS"):  # debugging aid: the completion intervals of the timed frames, in order
        print("INTERVALS", " ".join("%.3f" % x for x in intervals[:64]), file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-stage HIP-event times of the timed frames (ring of 128) + counters per pose --------
    # context c rendered frames c, c+K, ... of the timed region (and possibly more during warm-up/calibration)
    per_ctx = [len(range(c, args.steps, K)) for c in range(K)]
    if args.stage_events:
        st = np.array([scenes[c].timings(b) for c in range(K) for b in range(min(per_ctx[c], 128))], np.float64)  # [*, 6] ms
    else:
        st = np.zeros((0, 6))
    if st.size == 0:
        st = calib_ms[None, :]
    stage_ms = st.mean(axis=0)
    counts = []
    for i in range(min(64, args.steps)):
        pp = poses[(args.warmup + args.steps - 1 - i) % 64]
        # counters are per pose; re-render untimed to read them back (outside the timed region)
        pp.collect_timings = 0
        pp.strip_row_begin, pp.strip_row_end = own_rows  # N>1: this rank's strip (what the stage times describe)
        o = scene.render(pp, want_stats=True)
        counts.append((o.frustum_count, o.sorted_count, o.tile_pairs, o.error_flags, o.shaded_count, o.scanned_entries, o.escape_count))
        pp.collect_timings = 2 if args.stage_events else 0
        pp.strip_row_begin, pp.strip_row_end = 0, 0
    counts = np.array(counts, np.float64)
    Vf, Vs, D = counts[:, 0].mean(), counts[:, 1].mean(), counts[:, 2].mean()
    shaded, scanned = counts[:, 4].mean(), counts[:, 5].mean()
    escapes = counts[:, 6].mean()  # sorted splats whose bin rectangle is stored / gathered by id (the others' ride as codes)
    err = int(counts[:, 3].max())
    if world > 1:
        agg = torch.tensor([Vs, D, Vf], dtype=torch.float64, device="cuda")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        Vs_all = float(agg[0])
    else:
        Vs_all = Vs

    # whole-job aggregate: single GPU = frames of this GPU; N>1 = frames of the strip partition (every step produces ONE
    # frame, complete on every rank)
    fps = args.steps / elapsed
    Ppix = W * H

    alt_out = None
    if world > 1:
        # the throughput partition (no collective) as the secondary measurement: a failure here must not take the
        # headline line with it
        try:
            for i in range(args.warmup):
                frame(i)
            fence()
            t2 = time.perf_counter()
            for i in range(args.steps):
                frame(args.warmup + i)
            fence()
            el2 = time.perf_counter() - t2
            tt = torch.tensor([el2], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
            alt_out = {"value": args.steps * world / el2, "unit": "frames/s", "scaling": "weak", "ms_per_step": 1e3 * el2 / args.steps,
                       "partition": f"alternate frames: rank r renders poses r, r+{world}, ... (no data-path collective)"}
        except Exception as e:  # noqa: BLE001
            alt_out = {"error": f"{type(e).__name__}: {e}"[:300]}
    # algorithmic bytes per launch of each stage (SURVEY.md §8d per-unit figures)
    N = N * args.instances  # total global splats from here on
    alg = {
        # centres of every splat; opacity + covariance of the frustum survivors; 32-B record + rect + (key, id) of the sorted
        # ones (colour, view direction and the 180-B SH records are not touched here: shading is deferred to the compositor)
        # (round 6: the 4-byte rectangle by id is stored for the ESCAPES only — the other rectangles ride through the sort as codes)
        "project": 12 * N + (4 + 24) * Vf + (32 + 8) * Vs + 4 * escapes,
        "sort": 68 * Vs,
        # direct binning: ids + rect gather + sorted rect (count), sorted rect + ids + list append (emit)
        "bin": 20 * Vs + 4 * D,
        "pairsort": 0,
        # list entries actually walked + their 32-B record; colour + centre + the 192-B SH record of the staged ones; RGBA16F frame
        "composite": (4 + 32) * scanned + (16 + 12 + 192) * shaded + 8 * Ppix,
    }
    # `roofline` is SURVEY.md 8d's figure for the longest HBM-bound STAGE of the frame by measured (single-stream) time, chosen
    # at run time among project / sort / bin.  The compositor can be the longest stage, but it is bound by fp32 VALU issue (exp
    # + blend per pixel-splat pair; no matrix contraction, so neither "hbm" nor "mfma" describes it): it gets its own
    # `roofline_composite` object with its VALU utilisation instead of a made-up HBM fraction.  `roofline_project` and
    # `roofline_sort` are always there.
    hbm_stages = [0, 1, 2]
    dom = max(hbm_stages, key=lambda j: calib_ms[j] - (calib_ms[6] if j == 0 else 0.0))
    dom_name = STAGES[dom]
    longest = STAGES[max(range(5), key=lambda j: calib_ms[j] - (calib_ms[6] if j == 0 else 0.0))]
    # kernel duration: HIP events around the stage on an otherwise idle GPU (the untimed calibration frames).  With
    # several frames in flight the event span of a stage also contains queueing behind the other streams' kernels —
    # rocprofv3's per-kernel duration of this same command agrees with the calibration value, not with the span.
    # k_project alone: the project stage minus its head (MGS_STAGE_CULL: the span between the frame's first two event markers,
    # behind the upload — until round 3 a partition-cull kernel ran there, now only the markers' own latency), both from the
    # calibration frames — what rocprofv3 reports as the kernel's own average duration
    cull_ms = float(calib_ms[6])

    def stage_roofline(j):
        ms = float(calib_ms[j]) - (cull_ms if j == 0 else 0.0)
        ach = alg[STAGES[j]] / (ms * 1e-3) if ms > 0 else 0.0
        return {"bound": "hbm", "stage": STAGES[j], "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                "algorithmic_bytes_per_launch": alg[STAGES[j]], "launch_ms": ms}

    # (round 6) `roofline` describes the LONGEST stage of the frame, whatever bounds it: when that is the compositor
    # the object carries its SURVEY 8d byte fraction (`frac`: algorithmic bytes, mostly L2 / MALL hits), its PMC HBM bytes (`traffic`,
    # `traffic_frac`) AND its VALU issue fraction (`valu`), with the calibration caveat in `valu.clock_note`
    if longest == "composite":
        dom, dom_name = 4, "composite"
    dom_ms = float(calib_ms[dom]) - (cull_ms if dom == 0 else 0.0)
    achieved = alg[dom_name] / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
    b_frame = 12 * N + Vs * (16 + 24 + 180) + 8 * Vs + 68 * Vs + 2 * 48 * Vs + 8 * Ppix
    b_moved = alg["project"] + alg["sort"] + alg["bin"] + alg["composite"]  # what this build's kernels move (deferred shading: no 180 B/splat SH stream)
    frame_gpu_ms = stage_ms[5]
    sort_ms = calib_ms[1] if K > 1 else stage_ms[1]  # isolated sort time: overlapped spans are not kernel time

    # HBM traffic of the dominant kernel from the PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs of this same command, corrected as MI355X_MICROARCH.md prescribes; see profiles/*.json)
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        if (dom_name in pj["kernels"] and world == 1 and N == 5_830_000 and args.instances == 1 and args.scene == "garden"
                and args.pipeline == 0 and (W, H) == (1920, 1080) and not args.ply and not args.stochastic):
            traffic = pj["kernels"][dom_name]["traffic_bytes_per_launch_corrected"]
    except Exception:
        pass

    out = {
        "metric": "frames/s @1920x1080 + sorted Gsplats/s, garden-sized scene",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preroll_frames": preroll,  # untimed frames right in front of the timed region (after the synchronous calibration frames)
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.ply else f"file:{os.path.basename(args.ply)}",
        "frames_in_flight": K,
        "submission": "per-stage events + plain launches" if args.stage_events else "hipGraph replay (one upload + one graph launch per frame)",
        "config": {"workload": f"{'syn_garden' if not args.ply else os.path.basename(args.ply)} N={N} x {args.instances} instance(s) SH deg 3, storage sh/rgba format {args.sh_format}/{args.rgba_format}, "
                               f"{W}x{H}, 64-pose orbit r=4 h=1.5 fov60, GPU radix sort, cull at dist (configs[2])"
                               + ("" if args.scene == "garden" else f", scene variant `{args.scene}`") + (", additive alpha (no early termination)" if args.alpha_sum else "")
                               + (", 3DGUT pipeline" if args.pipeline == 1 else "") + (", stochastic splats" if args.stochastic else "")
                               + (f", depth of field aperture {args.dof}" if args.dof > 0.0 else ""),
                   "partition": "single GPU" if world == 1 else
                   f"{world} tile-row strips of every frame ({'equal' if args.equal_strips else 'cost-balanced'} rows {bounds}) + strip exchange: {gather_mode}"},
        "alternate_frames": alt_out,
        "this_rank_strip_rows": list(my_rows) if world > 1 else None,
        "gather_mode": gather_mode,  # which exchange ran (N > 1): libmgs's own RCCL call, or the torch.distributed fallback — never silent
        "sorted_gsplats_per_s": (Vs / (sort_ms * 1e-3)) / 1e9 if sort_ms > 0 else None,
        "sorted_gsplats_per_s_note": ("sorted pairs / time of the frame's sort stage (k_os_prepare + the two sort kernels + the launch that exits). "
                                      "EXCLUDED: the depth keys and the sort's first LSD pass, which the project kernel does while the keys are on chip "
                                      "(its hand-over groups each slot by the key's low byte: ~15 % of k_project's instructions); "
                                      "key_plus_sort_gsplats_per_s is depth key + cull + whole sort end to end"),
        "sorted_gsplats_per_s_in_frame_aggregate": Vs_all * fps / 1e9,  # all ranks' sorted elements x frames/s
        "visible_splats": {"frustum": Vf, "sorted": Vs, "tile_pairs": D, "shaded": shaded, "list_entries_scanned": scanned, "rect_escapes": escapes},
        "stage_ms": {STAGES[j]: float(stage_ms[j]) for j in range(6)},
        "stage_ms_single_stream": {STAGES[j]: float(calib_ms[j]) for j in range(6)},
        "frame_interval_ms_percentiles": {"p50": float(np.percentile(intervals, 50)), "p95": float(np.percentile(intervals, 95)),
                                          "mean": float(intervals.mean()), "max": float(intervals.max()),
                                          "note": "time between consecutive frame completions in the timed region (HIP events at each frame's end)"},
        "frame_span_ms_percentiles": {"p50": float(np.percentile(st[:, 5], 50)), "p95": float(np.percentile(st[:, 5], 95)),
                                      "min": float(st[:, 5].min()), "max": float(st[:, 5].max())},
        # frames/s with ONE frame in flight: graph-replayed serial frames (round 6); the sum of the calibration frames' stage events
        # (plain launches + six events per frame) is `value_single_frame_from_stage_events`
        "value_single_frame": (1e3 / serial_ms) if serial_ms else (1e3 / float(calib_ms[5]) if calib_ms[5] > 0 else None),
        "value_single_frame_from_stage_events": 1e3 / float(calib_ms[5]) if calib_ms[5] > 0 else None,
        "roofline": {"bound": "hbm", "stage": dom_name,
                     "kernels": {"project": "k_project", "sort": "k_os_prepare + 2 x k_os_pass (+ 1 that exits at once); pass 0 is virtual (done in k_project)",
                                 "bin": "k_dbin_count + k_dbin_scan + k_dbin_emit", "composite": "k_composite"}[dom_name],
                     "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "traffic_frac": (traffic / (dom_ms * 1e-3)) / HBM_PEAK if (traffic and dom_ms > 0) else None,
                     "actual_bound": "valu (fp32 vector issue + the serial chain per region; see `valu`)" if dom_name == "composite" else "hbm + valu (co-limited, DESIGN 3)",
                     "valu": composite_roofline(dom_ms, alg["composite"], world, N, args) if dom_name == "composite" else None,
                     "traffic_source": (f"profiles/{PMC_FILE}: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --inflight 1` "
                                        "(2*FETCH + WRITE per launch, MI355X_MICROARCH.md HBM section); not measured in this run") if traffic else None,
                     "algorithmic_bytes_per_launch": alg[dom_name], "launch_ms": float(dom_ms),
                     "note": f"the LONGEST stage of the frame by measured single-stream time (HIP events on the untimed calibration frames): `{longest}`; "
                             "`achieved` / `frac` = SURVEY.md 8d's algorithmic bytes for it / that time, `traffic` = its HBM bytes by the PMC counters"
                             + ("; k_composite is not an HBM kernel (its records and lists are L2 / MALL hits: traffic << algorithmic bytes) — what bounds it "
                                "is in `valu`; the longest HBM-bound stage is in roofline_project / roofline_sort / roofline_bin" if longest == "composite" else "")},
        "roofline_project": dict(stage_roofline(0), stage_ms_incl_head=float(calib_ms[0]), head_ms=cull_ms),
        "roofline_composite": composite_roofline(calib_ms[4] if K > 1 else stage_ms[4], alg["composite"], world, N, args),
        "roofline_sort": {"bound": "hbm", "achieved": (68 * Vs / (sort_ms * 1e-3)) / 1e9 if sort_ms > 0 else None,
                          "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                          "frac": (68 * Vs / (sort_ms * 1e-3)) / HBM_PEAK if sort_ms > 0 else None, "launch_ms": float(sort_ms),
                          "algorithmic_bytes_per_launch": 68 * Vs},
        "roofline_bin": stage_roofline(2),
        # the frame as a whole: the bytes this build's kernels move (sum of the per-stage algorithmic bytes; deferred shading
        # means the 180 B/splat SH stream of SURVEY.md 8d's B_frame does not exist) x frames/s; `frac_single_frame` is the same
        # bytes over the GPU time of one frame.  The B_frame form is kept as `survey_budget_ratio` (a budget comparison, not a fraction).
        "roofline_frame": {"bound": "hbm", "achieved": b_moved * fps / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                           "frac": b_moved * fps / HBM_PEAK if world == 1 else None,
                           "frac_single_frame": (b_moved / (frame_gpu_ms * 1e-3)) / HBM_PEAK if frame_gpu_ms > 0 else None,
                           "bytes_moved_per_frame": b_moved,
                           "survey_bytes_per_frame": b_frame,
                           # (not a fraction of work done: SURVEY's B_frame counts 0.70 GB of SH per frame that deferred shading never moves)
                           "survey_budget_ratio": b_frame * fps / HBM_PEAK if world == 1 else None,
                           "survey_budget_ratio_single_frame": (b_frame / (frame_gpu_ms * 1e-3)) / HBM_PEAK if frame_gpu_ms > 0 else None},
        "error_flags": err,
        "setup_s": setup_s,
    }

    if (rank == 0 and world == 1 and not args.ply and args.scene == "garden" and args.pipeline == 0 and args.instances == 1
            and not args.stochastic and args.splats == 5_830_000 and (W, H) == (1920, 1080) and args.sh_format == 0 and args.rgba_format == 0):
        # parity of THIS run's frames against the CPU oracle (SURVEY.md 8d "PSNR / max-abs vs oracle on 4 poses"): the oracle's
        # answers for the benchmark scene are the committed fixture tests/golden/full_size_garden.npz (sorted-stream hashes
        # + 256x256 crops, generator next to it) — data, not code: the oracle itself is not touched here
        try:
            import hashlib
            fx = np.load(os.path.join(ROOT, "tests", "golden", "full_size_garden.npz"))
            par = {"poses": [], "keys_bit_exact": True, "ids_match": True, "psnr_db_min": None, "max_abs_rgb": 0.0,
                   "source": "tests/golden/full_size_garden.npz (oracle key/id stream SHA-1 + crops of its frames, 4 poses)"}
            psnrs = []
            for v in range(4):
                pose = int(fx[f"v{v}_pose"])
                pp = poses[pose]
                so = scene.sort_keys(pp)
                gk, gi = scene.sort_download(so.count)
                sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
                par["keys_bit_exact"] &= bool(sha(gk) == str(fx[f"v{v}_sha_keys"]))
                par["ids_match"] &= bool(sha(gi[np.lexsort((gi, gk))]) == str(fx[f"v{v}_sha_ids"]))
                scene.render(pp)
                img = scene.download_frame(pp).astype(np.float32)
                for wi, (x0, y0, x1, y1) in enumerate(fx[f"v{v}_windows"]):
                    want = fx[f"v{v}_crop{wi}"].astype(np.float32)[..., :3]
                    got = img[y0:y1 + 1, x0:x1 + 1, :3]
                    mse = float(np.mean((got.astype(np.float64) - want) ** 2))
                    psnrs.append(99.99 if mse <= 0 else min(99.99, 10.0 * np.log10(1.0 / mse)))  # image_compare_metric.comp.slang:116-130
                    par["max_abs_rgb"] = max(par["max_abs_rgb"], float(np.abs(got - want).max()))
                par["poses"].append(pose)
            par["psnr_db_min"], par["psnr_db_mean"] = float(min(psnrs)), float(np.mean(psnrs))
            out["parity"] = par
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    if (rank == 0 and world == 1 and not args.ply and args.scene == "garden" and args.pipeline == 0 and args.instances == 1
            and not args.stochastic and not args.alpha_sum and args.dof == 0.0 and args.sh_format == 0 and args.rgba_format == 0
            and not args.no_extras):
        # The same workload in the reference's OTHER defaults, so that the headline cannot be mistaken for them (untimed extras,
        # a few dozen frames each, same contexts and frames in flight as the headline):
        #  * value_reference_alpha: MGS_ALPHA_SUM — A = sum(alpha), every fragment composited, no early termination: the blend
        #    state of the reference's default pipeline (src/gaussian_splatting.cpp:2081-2086); the headline's alpha is 1 - T with
        #    early termination at T < 1e-4, the reference's FRONT_TO_BACK alpha.  Colour is the same to the bit in both.
        #  * value_uint8_storage: SH and colour stored as uint8, the reference's default storage (src/parameters.h:88-89); the
        #    headline uses fp32 storage (the benchmark setting of SURVEY.md 8d).
        def timed_fps(plist, steps, warm, inflight):
            kk = max(1, min(K, inflight))
            best = 0.0
            for rep in range(2):  # best of two: right after a re-commit one run in a few is hit by a one-off ~1 s stall (allocator)
                for i in range(warm):
                    with torch.cuda.stream(streams[i % kk]):
                        scenes[i % kk].render(plist[i % 64])
                fence()
                tq = time.perf_counter()
                for i in range(steps):
                    with torch.cuda.stream(streams[i % kk]):
                        scenes[i % kk].render(plist[(warm + i) % 64])
                fence()
                best = max(best, steps / (time.perf_counter() - tq))
            return best
        try:
            for pp in poses:
                pp.alpha_mode = capi.ALPHA_SUM
            out["value_reference_alpha"] = timed_fps(poses, 24, 4, K)
            out["value_reference_alpha_single_frame"] = timed_fps(poses, 24, 4, 1)
            out["value_reference_alpha_note"] = ("frames/s with alpha_mode = MGS_ALPHA_SUM (additive alpha, no early termination: the reference's "
                                                 "default blend state, gaussian_splatting.cpp:2081-2086); `value` is alpha = 1 - T with early termination")
        except Exception as e:  # noqa: BLE001
            out["value_reference_alpha"] = None
            out["value_reference_alpha_note"] = f"{type(e).__name__}: {e}"[:200]
        finally:  # (whatever happened, what follows runs in the headline's mode again)
            for pp in poses:
                pp.alpha_mode = capi.ALPHA_COVERAGE
        try:
            scene.commit(2, 2)  # contexts re-size their working sets at their next frame
            out["value_uint8_storage"] = timed_fps(poses, 48, 8, K)
            out["value_uint8_storage_single_frame"] = timed_fps(poses, 48, 8, 1)
            out["value_uint8_storage_note"] = "frames/s with SH and colour stored as uint8 (the reference's default storage, src/parameters.h:88-89)"
        except Exception as e:  # noqa: BLE001
            out["value_uint8_storage"] = None
            out["value_uint8_storage_note"] = f"{type(e).__name__}: {e}"[:200]
        finally:
            try:
                scene.commit(args.sh_format, args.rgba_format)
            except Exception:  # noqa: BLE001
                pass

    if rank == 0 and world == 1 and args.pipeline == 0 and not args.stochastic and not args.no_extras:
        # depth key + cull + sort end to end (mgs_sort_keys: k_project<false>, then the whole key sort), the like-for-like GPU figure
        # beside cpu_baseline, which times depth key + sort on the host (sorted_gsplats_per_s is the sort stage alone)
        try:
            best, cnt = None, 0
            for i in range(6):
                so = scenes[0].sort_keys(poses[i % 64])
                tms = float(so.key_ms) + float(so.sort_ms)
                if i >= 2 and tms > 0 and (best is None or tms < best):
                    best, cnt = tms, int(so.count)
            out["key_plus_sort_gsplats_per_s"] = (cnt / (best * 1e-3) / 1e9) if best else None
            out["key_plus_sort_ms"] = best
            out["key_plus_sort_note"] = ("mgs_sort_keys end to end on this workload: depth key + dist-stage cull over all splats (k_project<false>) + the key sort of the "
                                         "survivors; sorted pairs / time, BEST of four calls (an extra: skipped with --no-extras)")
        except Exception as e:  # noqa: BLE001
            out["key_plus_sort_gsplats_per_s"] = None
            out["key_plus_sort_note"] = f"{type(e).__name__}: {e}"[:200]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: restated SplatSorterAsync::innerSort on a bounded sample of the same workload
        from oracle import binding as ob
        nb = min(N, args.splats)  # the whole splat set of the workload (0.45 s at 5.83 M on this box's host)
        V0 = np.array(poses[0].view, np.float32).reshape(4, 4).T
        fwd = -V0[2, :3]
        eye0 = synth.orbit_pose(0)
        reps, best = 2, None
        for _ in range(reps):
            _, _, dms, sms = ob.cpu_sort(fwd, eye0, [(sc["positions"][:nb], None)], threads=0)
            best = (dms + sms) if best is None else min(best, dms + sms)
        out["cpu_baseline"] = {"value": nb / (best * 1e-3) / 1e9, "unit": "Gsplats/s (depth key + sort)",
                               "cores": os.cpu_count(), "kind": "port",
                               "sample": f"all {nb} splats of the workload's splat set, pose 0, best of {reps}; distance loop on all "
                                         f"cores, std::sort(par_unseq) serial unless libstdc++ finds TBB",
                               "ms": best}
    if rank == 0:
        print(json.dumps(out, allow_nan=False))
    for sc_k in reversed(scenes):
        sc_k.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
