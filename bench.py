#!/usr/bin/env python3
"""bench.py -- reconstructed frames/s of the cista-eiflow hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 4
    python bench.py --gpus N ...            (no torchrun environment: starts the N-rank job itself as a child process)
    python bench.py --gpus 8 --strong 32 --height 480 --width 640      (configs[3]: 32 sequences split over the ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one reconstructed frame for each of the B=8 independent event sequences a GPU holds
(BASELINE.json configs[1]: cista-eiflow, 180x240, batch 8): flow net (DCEIFlow, 6 refinement iterations)
-> any() -> warp I / warp Z -> CISTA-LSTC, fed back recurrently exactly like test_with_flow.py:120-156.
Inputs (synthetic event voxel grids) are resident in HBM before the timed region; weights are
seeded random (no checkpoints exist offline).  N GPUs = N x 8 sequences (weak scaling, no data-path
collective; an RCCL all-gather collates the reconstructed frames).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (fp32-MFMA implicit-GEMM conv) -- algorithmic flops / HIP-event time
  "cpu_baseline": the CPU oracle (a port of the reference graph, eager PyTorch fp32) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: dense f16 / bf16 MFMA (CF_PRECISION=f16; f16x3 spends 3 MFMAs per product)


def mfma_peak():
    """Dense MFMA peak for the arithmetic the convolutions run in (algorithmic flops are priced against it)."""
    prec = os.environ.get("CF_PRECISION", "f32")
    return {"f32": PEAK_FP32_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS / 3.0}[prec]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=240)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10, help="timed CPU-baseline frames per batch size (after 2 warm-up frames)")
    ap.add_argument("--repeat", type=int, default=3, help="timed blocks of --steps steps each; the MEDIAN block is reported")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 / 2 / 4 legs (latency regime; N = 1 only)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra f16x3-precision timing leg")
    ap.add_argument("--model", default="eiflow", choices=["eiflow", "eraft", "idnet"],
                    help="flow network (the BASELINE metric is eiflow; the others are extra workloads)")
    ap.add_argument("--strong", type=int, default=0, metavar="SEQUENCES",
                    help="strong scaling: a FIXED total of SEQUENCES (configs[3]: 32) split over the N ranks "
                         "(contiguous shards, parallel.shard_range) instead of --batch sequences per GPU")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the one-process-per-GPU job ourselves as a
    CHILD process (torch.distributed.run), before this process has made any HIP call -- a process that has touched the GPU
    must never be replaced or fork GPU work -- relay rank 0's JSON line and exit with the child's return code.
    --standalone lets the launcher pick its own free rendezvous port (no bind / close / re-bind window in which another
    job on the shared network namespace could take it); the child's output is relayed line by line, so a hang is visible."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CF_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1)
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{"):
            print(ln, flush=True)
        elif ln.strip():
            print(ln, file=sys.stderr, flush=True)
    raise SystemExit(proc.wait())


def pin_rank_to_cores(local_rank, world):
    """One rank per GPU enqueues ~300 launches per 5 ms step from ONE host thread: give every rank its own slice of the cores
    this process may run on (before the first HIP call, so the runtime's helper threads inherit it).  No-op for a single
    rank or where the platform has no sched_setaffinity."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // world
        if per < 1:
            return None
        mine = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return len(mine)
    except OSError:
        return None


def cpu_info():
    """(model string, physical cores this process may use) from /proc/cpuinfo + the affinity mask."""
    model, phys = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        cur = {}
        for ln in open("/proc/cpuinfo"):
            if ":" in ln:
                k, v = [t.strip() for t in ln.split(":", 1)]
                cur[k] = v
            elif cur:
                if "model name" in cur:
                    model = cur["model name"]
                if allowed is None or int(cur.get("processor", -1)) in allowed:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor", "0"))))
                cur = {}
    except OSError:
        pass
    return model, max(len(phys), 1)


def model_args(H, W):
    return argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode="forward", base_channels=64, depth=5, ds=8,
                              is_bi=False)


def cpu_baseline(B, H, W, frames):
    """The CPU oracle (port of the reference's eager fp32 graph) on the same workload, BASELINE.md section 3's protocol:
    2 warm-up + `frames` (>= 10) timed frames of the recurrent loop, MEDIAN frame time, at B = 1 and at B = `B`; CPU model and
    physical-core count in the record.  Threads: the GPU box's CPU share for one GPU is 16 cores (torch's default, all logical
    CPUs of the host, oversubscribes it), so min(16, physical cores this process may use)."""
    import statistics
    import weights_util as wu
    from oracle import cista_oracle as orc
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
    m = DCEIFlowCistaNet(model_args(H, W))
    wu.fill_module(m, 1234)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    model_name, phys = cpu_info()
    cores = min(16, phys)
    torch.set_num_threads(cores)
    warm = 2

    def leg(b):
        states, prev = None, torch.zeros(b, 1, H, W)
        times = []
        with torch.no_grad():
            for t in range(warm + frames):
                ev = wu.synth_events(b, 5, H, W, 1234 + t)
                t0 = time.perf_counter()
                I, _, states = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev}, states)
                times.append(time.perf_counter() - t0)
                prev = I
        med = statistics.median(times[warm:])
        return {"batch": b, "frames_per_s": round(b / med, 3), "median_ms_per_frame_step": round(med * 1e3, 1), "timed_frames": frames}

    legs = [leg(1)] + ([leg(B)] if B != 1 else [])
    return {"value": legs[-1]["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
            "cpu_model": model_name, "physical_cores_available": phys, "legs": legs,
            "sample": "median of %d timed frames after %d warm-up frames of the recurrent %dx%d cista-eiflow loop, at B=1 and B=%d "
                      "(value = the B=%d leg; oracle/cista_oracle.py, eager PyTorch fp32, %d threads)" % (frames, warm, H, W, B, B, cores)}


PEAK_HBM_TBS = 8.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable with a float4 copy)


def roofline_pass(model, step, B, dev, nprof):
    """Separate pass: every launch of the step bracketed by HIP events on its launch stream, side streams folded into
    the caller's stream (each kernel alone on the chip) with the SAME grids as the timed step (the two half-batch
    CISTA chains are kept and issued back to back).  Contraction kernels are priced against the fp32 MFMA peak with
    algorithmic flops (2*M*N*K, un-padded K), the rest against the HBM peak with algorithmic bytes (each input read
    once, each output written once)."""
    h = model._be().get(B, dev)
    h.profile_enable(True)
    with torch.no_grad():
        for _ in range(nprof):
            step(gather=False)       # rank 0 only: no collective may be issued in this leg
    torch.cuda.synchronize()
    recs = h.profile_read()
    rows = h.profile_rows()
    h.profile_enable(False)
    if os.environ.get("CF_LAYER_REPORT"):
        with open(os.environ["CF_LAYER_REPORT"], "w") as f:
            f.write("# per-launch-site timing over %d steps (HIP events on the launch stream, kernels serialised, same grids as the timed step)\n" % nprof
                    + h.profile_report())
        with open(os.environ["CF_LAYER_REPORT"] + ".json", "w") as f:
            json.dump({"steps": nprof, "rows": rows}, f)

    def agg(sel):
        out = {}
        for r in rows:
            if not sel(r):
                continue
            a = out.setdefault(r["kernel"], {"ms": 0.0, "work": 0.0, "launches": 0})
            a["ms"] += r["ms"]; a["work"] += r["work"]; a["launches"] += r["launches"]
        return out

    def executed(r):
        """flops the matrix cores execute for a launch-site row = algorithmic flops x the library's own per-tile ratio
        (cf_conv_tile_mfma_ratio: 4/9 Winograd F(2x2,3x3), 0.6 F(2,5), 0.25 F(4x4,3x3), 1 for the direct kernels)"""
        return r["work"] * r.get("mfma_ratio", 1.0)

    mf = agg(lambda r: r["class"] == "mfma")
    hb = agg(lambda r: r["class"] == "hbm")
    dom_name = max(mf, key=lambda k: mf[k]["ms"])
    dom = mf[dom_name]
    ach = dom["work"] / (dom["ms"] * 1e-3) / 1e12                     # algorithmic TFLOP/s
    dom_exec = sum(executed(r) for r in rows if r["class"] == "mfma" and r["kernel"] == dom_name)
    ach_x = dom_exec / (dom["ms"] * 1e-3) / 1e12                      # executed on the matrix cores
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(dom_name) or tj.get(dom_name.replace(",3>", ">"))
            if ent:
                traffic = {"bytes_per_launch": ent.get("bytes_per_launch"), "static": True, "source": "profiles/hbm_traffic.json",
                           "collected": tj.get("_meta", {}).get("collected", "round 1 build (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the guide)"),
                           "note": "NOT measured by this run: PMC counters need their own rocprofv3 passes (tools/make_profiles.sh)"}
        except Exception:
            traffic = None
    mf_ms = sum(v["ms"] for v in mf.values())
    mf_work = sum(v["work"] for v in mf.values())
    hb_ms = sum(v["ms"] for v in hb.values())
    hb_work = sum(v["work"] for v in hb.values())

    def cls(sel):
        m_ms = sum(r["ms"] for r in rows if sel(r) and r["class"] == "mfma")
        m_w = sum(r["work"] for r in rows if sel(r) and r["class"] == "mfma")
        m_x = sum(executed(r) for r in rows if sel(r) and r["class"] == "mfma")
        b_ms = sum(r["ms"] for r in rows if sel(r) and r["class"] == "hbm")
        b_w = sum(r["work"] for r in rows if sel(r) and r["class"] == "hbm")
        tot = m_ms + b_ms
        mfr = (m_w / (m_ms * 1e-3) / 1e12 / mfma_peak()) if m_ms else None          # algorithmic flops (may pass 1: Winograd)
        mxr = (m_x / (m_ms * 1e-3) / 1e12 / mfma_peak()) if m_ms else None          # executed on the matrix cores (<= 1)
        hfr = (b_w / (b_ms * 1e-3) / 1e12 / PEAK_HBM_TBS) if b_ms else None
        comb = ((m_ms * (mxr or 0) + b_ms * (hfr or 0)) / tot) if tot else None
        return {"ms_per_step": round(tot / nprof, 3), "mfma_ms_per_step": round(m_ms / nprof, 3), "hbm_ms_per_step": round(b_ms / nprof, 3),
                # roofline fractions (<= 1): executed matrix-core flops against the fp32 MFMA peak, algorithmic bytes against the HBM peak,
                # and their time-weighted combination; the algorithmic-flop figure stands beside them
                "mfma_executed_frac": None if mxr is None else round(mxr, 4),
                "mfma_algorithmic_frac": None if mfr is None else round(mfr, 4),
                "hbm_frac": None if hfr is None else round(hfr, 4),
                "time_weighted_frac": None if comb is None else round(comb, 4)}

    return {
        # `achieved` / `frac`: flops the matrix cores EXECUTE for the dominant kernel (a roofline fraction, <= 1); the algorithmic
        # figure (2*M*N*9*Cin of the 3x3 convolution the reference computes; > executed for the Winograd kernels) stands beside it
        "bound": "mfma", "kernel": dom_name, "achieved": round(ach_x, 2), "peak": round(mfma_peak(), 1),
        "unit": "TFLOP/s", "frac": round(ach_x / mfma_peak(), 4),
        "algorithmic_achieved": round(ach, 2), "algorithmic_frac": round(ach / mfma_peak(), 4),
        "mfma_ratio": round(dom_exec / dom["work"], 4), "traffic": traffic,
        "note": "separate pass, HIP events on the launch stream, side streams folded into one stream (each kernel alone on the "
                "chip), SAME launch grids as the timed step (half-batch CISTA chains kept, issued back to back)",
        "mfma_executed_tflops": round(ach_x, 2), "mfma_executed_frac": round(ach_x / mfma_peak(), 4),     # (= achieved / frac; kept for continuity with r02 / r03 lines)
        "launches_per_step": dom["launches"] / nprof,
        "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
        "flops_per_launch": dom["work"] / dom["launches"],
        "all_conv": {"achieved": round(mf_work / (mf_ms * 1e-3) / 1e12, 2),
                     "executed": round(sum(executed(r) for r in rows if r["class"] == "mfma") / (mf_ms * 1e-3) / 1e12, 2),
                     "ms_per_step": round(mf_ms / nprof, 3),
                     "launches_per_step": sum(v["launches"] for v in mf.values()) / nprof, "gflop_per_step": round(mf_work / nprof / 1e9, 2)},
        "by_kernel": {k: {"ms_per_step": round(v["ms"] / nprof, 3), "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                          "executed_tflops": round(sum(executed(r) for r in rows if r["class"] == "mfma" and r["kernel"] == k) / (v["ms"] * 1e-3) / 1e12, 2),
                          "launches_per_step": v["launches"] / nprof} for k, v in sorted(mf.items(), key=lambda kv: -kv[1]["ms"])},
        "launches_per_step_all": sum(r["launches"] for r in rows) / nprof,
        # the HBM class: algorithmic bytes / HIP-event time against the 8 TB/s spec peak, per kernel and time-weighted
        "hbm": {"peak": PEAK_HBM_TBS, "unit": "TB/s", "achieved": round(hb_work / (hb_ms * 1e-3) / 1e12, 3) if hb_ms else None,
                "frac": round(hb_work / (hb_ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4) if hb_ms else None,
                # the same over the launches that move >= 8 MB (the others last 5-7 us whatever they move: launch-latency-bound)
                "streaming_frac": (lambda big: round(sum(v["work"] for v in big) / (sum(v["ms"] for v in big) * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)
                                   if big else None)([r for r in rows if r["class"] == "hbm" and r["work"] / r["launches"] >= 8e6]),
                "ms_per_step": round(hb_ms / nprof, 3), "launches_per_step": sum(v["launches"] for v in hb.values()) / nprof,
                "by_kernel": {k: {"ms_per_step": round(v["ms"] / nprof, 4), "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                                  "mbytes_per_launch": round(v["work"] / v["launches"] / 1e6, 3),
                                  "tbs": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 3),
                                  "frac": round(v["work"] / (v["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 4),
                                  "launches_per_step": v["launches"] / nprof} for k, v in sorted(hb.items(), key=lambda kv: -kv[1]["ms"])}},
        # north-star step "CISTA unrolled iterations + flow warp": both classes and their time-weighted combination
        "cista_warp_step": cls(lambda r: r["tag"].startswith("cista.") or r["tag"].startswith("warp.")),
        "whole_step": cls(lambda r: True),
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)                      # never returns
    if a.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (a.gpus, world, a.gpus))
    pinned = pin_rank_to_cores(local_rank, world)       # before the first HIP call (torch.cuda.device_count() makes none on this image)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # CF_BENCH_REHEARSE=1: control-flow rehearsal of the N > 1 path on a box with ONE GPU (all ranks share cuda:0,
    # collectives go through gloo on host tensors).  Numbers from it mean nothing; it exists because the real
    # RCCL run only happens on the driver's 8-GPU node.
    # CF_BENCH_REHEARSE=nccl: the same with the REAL backend -- init_process_group("nccl", device_id=cuda:0) on every rank
    # and device-side all_gather_into_tensor on the side stream -- where RCCL accepts two ranks on one device.
    rmode = os.environ.get("CF_BENCH_REHEARSE", "")
    rehearse = rmode == "1"                       # gloo + host tensors
    shared = rmode in ("1", "nccl")               # all ranks share cuda:0
    dev = torch.device("cuda", 0 if shared else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # CF_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, side-stream all_gather_into_tensor with
    # record_stream, barriers, MAX all-reduce) with a world of ONE rank: RCCL refuses two ranks on one device
    # ("Duplicate GPU detected"), so this is as much of the real backend as a one-GPU box can exercise
    force_dist = os.environ.get("CF_BENCH_FORCE_DIST") == "1" and world == 1
    dist_on = world > 1 or force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearse else dev      # where the timing all-reduce lives

    import weights_util as wu
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet, ERAFTCistaNet, IDCistaNet
    from cista_flow_amd.parallel import collate_frames, shard_range
    B, H, W = a.batch, a.height, a.width
    if a.strong:
        s0, s1 = shard_range(a.strong, rank, world)
        B = s1 - s0
        if B < 1:
            raise SystemExit("--strong %d leaves rank %d of %d without a sequence" % (a.strong, rank, world))
    total_seq = a.strong if a.strong else world * B
    cls = {"eiflow": DCEIFlowCistaNet, "eraft": ERAFTCistaNet, "idnet": IDCistaNet}[a.model]
    model = cls(model_args(H, W)).eval()
    wu.fill_module(model, 1234)
    model = model.to(dev)
    if a.model != "idnet":
        model.event_flownet.return_flow_preds = True  # like the reference: every iteration's up-flow is produced
    if a.model == "eraft":
        # the driver loop below carries `event_voxel_old = previous event_voxel` as the SAME tensor object, never written
        # in between (test_with_flow.py:144-149): the aliasing contract the fnet feature reuse needs
        model.reuse_prev_features = True
    R = 8
    evs = [wu.synth_events(B, 5, H, W, 1234 + 100 * rank + i).to(dev) for i in range(R)]
    side = torch.cuda.Stream(device=dev) if dist_on else None

    state = {"prev": torch.zeros(B, 1, H, W, device=dev), "states": None, "i": 0, "flow_init": None}

    def step(gather=True):
        ev = evs[state["i"] % R]
        if a.model == "eiflow":
            I, bf, st = model({"event_voxel": ev, "rec_img0": state["prev"]}, state["states"], {})
        elif a.model == "eraft":     # test_with_flow.py:144-149 (the previous voxel grid; a non-degenerate one on frame 0)
            I, bf, st = model({"event_voxel": ev, "event_voxel_old": evs[(state["i"] - 1) % R], "rec_img0": state["prev"]},
                              state["states"], {})
        else:                        # test_with_flow.py:150-154
            I, bf, st = model({"event_voxel": ev, "rec_img0": state["prev"]}, state["states"], state["flow_init"], {})
            state["flow_init"] = bf["next_flow"]
        state["prev"], state["states"] = I, st
        state["i"] += 1
        if dist_on and gather:
            # collate the reconstructed frames of all ranks (RCCL all-gather over xGMI) off the critical path
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                I.record_stream(side)          # I was allocated on the main stream; keep it alive for the gather
                state["gathered"] = collate_frames(I.cpu() if rehearse else I, n_sequences=a.strong or None, force=force_dist)
        return I

    # W untimed warm-up steps, then `repeat` timed blocks of EXACTLY K steps each, every block bracketed by barrier +
    # synchronize on both sides and reduced with MAX over the ranks; the MEDIAN block is the reported one (boxes and runs
    # differ by +-2.5 %: one block of 20 steps = 100 ms is a thin sample)
    blocks, my_blocks = [], []
    with torch.no_grad():
        for _ in range(max(a.warmup, 0)):
            step()
        for _ in range(max(a.repeat, 1)):
            if dist_on:
                torch.cuda.current_stream().wait_stream(side)
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            if dist_on:
                torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            my_blocks.append(el)
            if dist_on:
                t = torch.tensor([el], device=cdev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            blocks.append(el)
    el = sorted(blocks)[(len(blocks) - 1) // 2]           # median block (lower median for an even count)
    # per-rank view of the same blocks (each rank's own clock around its median block): min / max over the ranks
    my_ms = sorted(my_blocks)[(len(my_blocks) - 1) // 2] / a.steps * 1e3
    rank_ms = [my_ms]
    if dist_on:
        t = torch.tensor([my_ms, -my_ms], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rank_ms = [-float(t[1].item()), float(t[0].item())]
    assert torch.isfinite(state["prev"]).all(), "non-finite reconstruction"

    alt = None
    if not a.no_alt and os.environ.get("CF_PRECISION") is None:
        # the same workload with the opt-in f16x3 arithmetic (3 x f16 MFMA per product, fp32 tensors / accumulate,
        # ~2e-5 end-to-end vs the reference); reported next to the exact-fp32 headline, never instead of it
        m2 = cls(model_args(H, W)).eval()
        m2.precision = "f16x3"
        wu.fill_module(m2, 1234)
        m2 = m2.to(dev)
        st2 = {"prev": torch.zeros(B, 1, H, W, device=dev), "states": None, "i": 0, "flow_init": None}

        def step2():
            ev = evs[st2["i"] % R]
            if a.model == "eiflow":
                I, bf, st = m2({"event_voxel": ev, "rec_img0": st2["prev"]}, st2["states"], {})
            elif a.model == "eraft":
                I, bf, st = m2({"event_voxel": ev, "event_voxel_old": evs[(st2["i"] - 1) % R], "rec_img0": st2["prev"]}, st2["states"], {})
            else:
                I, bf, st = m2({"event_voxel": ev, "rec_img0": st2["prev"]}, st2["states"], st2["flow_init"], {})
                st2["flow_init"] = bf["next_flow"]
            st2["prev"], st2["states"] = I, st
            st2["i"] += 1

        with torch.no_grad():
            for _ in range(max(a.warmup, 2)):
                step2()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step2()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el2], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        # agreement of the two arithmetic modes on the last frame (both ran the same sequence length)
        alt = {"precision": "f16x3 (operands split hi+lo into f16, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate; 3x3 layers on the exact-fp32 Winograd kernel)",
               "value": round(total_seq * a.steps / el2, 2), "unit": "frames/s", "ms_per_step": round(el2 / a.steps * 1e3, 3),
               "opt_in": "module.precision = 'f16x3' or CF_PRECISION=f16x3"}
        del m2

    # rank 0's roofline pass (no collectives inside) comes AFTER every timed leg and its barriers, so the other ranks only
    # wait for it at the final barrier before tearing the process group down
    roofline = None
    if not a.no_roofline and rank == 0:
        roofline = roofline_pass(model, step, B, dev, min(a.steps, 10))

    # the latency regime the reference's own drivers run in (test_with_flow.py: B = 1): same step, smaller batches, N = 1 only
    latency = None
    if rank == 0 and world == 1 and not a.no_latency and not a.strong and B > 1:
        latency = {}
        for b in (1, 2, 4):
            if b >= B:
                continue
            evb = [e[:b].contiguous() for e in evs]
            stb = {"prev": torch.zeros(b, 1, H, W, device=dev), "states": None, "i": 0, "flow_init": None}

            def stepb():
                ev = evb[stb["i"] % R]
                if a.model == "eiflow":
                    I, bf, st = model({"event_voxel": ev, "rec_img0": stb["prev"]}, stb["states"], {})
                elif a.model == "eraft":
                    I, bf, st = model({"event_voxel": ev, "event_voxel_old": evb[(stb["i"] - 1) % R], "rec_img0": stb["prev"]}, stb["states"], {})
                else:
                    I, bf, st = model({"event_voxel": ev, "rec_img0": stb["prev"]}, stb["states"], stb["flow_init"], {})
                    stb["flow_init"] = bf["next_flow"]
                stb["prev"], stb["states"] = I, st
                stb["i"] += 1

            with torch.no_grad():
                for _ in range(max(a.warmup, 2)):
                    stepb()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    stepb()
                torch.cuda.synchronize()
                elb = time.perf_counter() - t0
            latency["b%d" % b] = {"frames_per_s": round(b * a.steps / elb, 1), "ms_per_step": round(elb / a.steps * 1e3, 3)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.model == "eiflow":
        cpu = cpu_baseline(B, H, W, max(a.cpu_frames, 10))

    if rank == 0:
        value = total_seq * a.steps / el
        if a.strong:      # ragged contiguous shards: say the range, not rank 0's size
            sizes = [shard_range(a.strong, r, world)[1] - shard_range(a.strong, r, world)[0] for r in range(world)]
            shard_desc = str(sizes[0]) if min(sizes) == max(sizes) else "%d..%d" % (min(sizes), max(sizes))
        else:
            shard_desc = str(B)
        rccl = None
        if dist_on and not rehearse:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        out = {
            "metric": "reconstructed frames/sec at %dx%d, cista-%s" % (H, W, a.model), "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3),
            "repeat": len(blocks), "block_ms_per_step": [round(b / a.steps * 1e3, 3) for b in blocks],
            "rank_ms_per_step": {"min": round(rank_ms[0], 3), "max": round(rank_ms[-1], 3)}, "cores_pinned_per_rank": pinned,
            "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None,
            "n_ranks_seen": dist.get_world_size() if dist_on else 1, "collective_backend": (("gloo (rehearsal)" if rehearse else "rccl " + str(rccl)) if dist_on else None),
            "dtype": {"f32": "f32", "f16x3": "f16x3->f32acc", "f16": "f16->f32acc"}[os.environ.get("CF_PRECISION", "f32")],
            "data": "synthetic",
            "config": {"workload": "cista-%s %dx%d batch=%s sequences per GPU (BASELINE configs[%d]), flow iters %d, "
                                   "CISTA depth 5, seeded random weights" % (a.model, H, W, shard_desc,
                                                                            {"eiflow": 3 if (H, W) == (480, 640) else 1, "eraft": 2, "idnet": 4}[a.model],
                                                                            model.flow_iters),
                       "sequences_per_gpu": int(shard_desc) if shard_desc.isdigit() else shard_desc, "total_sequences": total_seq, "height": H, "width": W, "parallelism": "dp%d (independent sequences)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "alt_precision": alt, "latency_regime": latency,
        }
        print(json.dumps(out), flush=True)
    if dist_on:
        if force_dist:
            assert state["gathered"].shape == state["prev"].shape and torch.equal(state["gathered"], state["prev"])
        dist.barrier()          # rank 0 may still be in its roofline / printing leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
