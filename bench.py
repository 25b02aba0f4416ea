#!/usr/bin/env python
"""bench.py -- throughput of the per-block Vorbis encode analysis on MI355X.

A "step" is one pass of the hot path over one batch of synthetic 44.1 kHz stereo white-noise
blocks that is already resident in HBM.  Default workload (BASELINE.json config 4, per-GPU
shard): 131 072 stereo 2048-sample blocks per GPU through the FULL mapping0_forward analysis
(window + MDCT + FFT + noise/tone masking + floor1 fit + couple/quantise), q=0.4 tables,
(lW,W,nW)=(1,1,1), blocktype LONG, ampmax_in=-9999 for every block (SURVEY.md 8d "C4").
`--workload c3` stops after the masking curves (config 3), `--workload c2` is mdct_forward only
(config 2).

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; rank 0 loads the setup blob and broadcasts its bytes over
RCCL (the only collective: blocks are independent); every rank then analyses its
own shard (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The ROCm runtime maps HIP streams onto 4 hardware queues unless told otherwise; the host-fed farm keeps five groups
# in flight, each on a stream of its own (+ the library's side streams), and streams that share a queue serialise.
# A deployment knob of the runtime, read when it initialises -- set before torch loads it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vorbis_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

# algorithmic HBM bytes per unit, fp32/int32, every tensor touched once, tables excluded
# (restated in DESIGN.md 5)
ALG_BYTES = {
    "c5": None,    # mixed sizes: 5248 B per short (256-sample) and 41216 B per long stereo block, see StreamRunner
    "c2": 12288,   # per channel-frame: 8192 in + 4096 out
    "c3": 40960,   # per stereo block: 16384 in + mdct 8192 + noise 8192 + tone 8192
    "c4": 41216,   # per stereo block: 16384 in + mdct 8192 + logmask 8192 + iwork 8192 + 256 posts/flags
}


def stage_bytes(stage, n, ch=2):
    """What a stage of the pipeline as launched must itself move per block of n samples and ch channels (every tensor it
    reads or writes once; tables excluded): the 'own bytes' of dominant_kernel.  Scales with the block size, so a
    mixed-size workload adds its size classes up (StreamRunner.stage_bytes_total)."""
    n2 = n // 2
    nlp = {256: 592, 2048: 784}.get(n, n2)       # octave lines of the 44.1 kHz psy setups, padded to 16
    nrp = {256: 100, 2048: 316}.get(n, n2 // 3)  # runs of bins sharing an octave line: logfft travels as one peak per run
    per_ch = {
        "transform": 4 * n + 4 * n2 + 4 * nrp + 5,                # pcm in; spectrum + run peaks out; local ampmax, status
        "ampmax": 6,
        "noisemask": 4 * n2 + 4 * n2,                             # spectrum in, noise curve out
        "tonemask": 4 * nrp + 8 + 4 * nlp + 4 * nlp + 2 * nlp + 4,  # seed: run peaks in, seed lines out; chase: lines in, survivors out
        "floor": 4 * n2 + 4 * n2 + 4 * nlp + 2 * nlp + 4 * n2 + 4 * n2 + n2 + 136,  # noise, spectrum, lines, survivors in; mixed spectrum, mask, byte curve, posts out
        "couple": 4 * n2 + n2 + 4 * n2 + 4,                       # mixed spectrum + byte curve in, residue out
    }
    return per_ch.get(stage, 12288) * ch


def source_hash():
    """sha256 over the kernel sources and the ABI headers: what a PMC profile must have been taken from to
    still describe the library this run loads."""
    import hashlib
    h = hashlib.sha256()
    for d in ("vorbis_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".h", ".hip", ".inc")):
                h.update(f.encode())
                h.update(open(os.path.join(ROOT, d, f), "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_PROFILE = "profiles/r06_pmc_traffic.json"
TRAFFIC_PROFILE_C5 = "profiles/r06_c5_pmc_traffic.json"
VALU_PROFILE = "profiles/r06_pmc_valu.json"
VALU_PROFILE_C5 = "profiles/r06_c5_pmc_valu.json"


def measured_valu(workload, units, stage_ms, clock_ghz=None):
    """The other roofline: the path is bound by vector-instruction issue, not by HBM.  From the committed counter
    pass (SQ_INSTS_VALU per kernel and its dynamic mix, hash-guarded like the traffic profile) and the price list of
    tools/micro/chip_rate.hip: vector instructions per unit x SIMD cycles per instruction / (1024 SIMDs x clock) = the time the
    instructions alone need on this chip, per stage and for the step, against the HIP-event time of the same stage."""
    if workload == "c5":
        return measured_valu_c5(units, stage_ms, clock_ghz)
    if workload not in ("c3", "c4"):
        return None
    try:
        t = json.load(open(os.path.join(ROOT, VALU_PROFILE)))
    except Exception:
        return {"value": None, "note": "no committed VALU profile"}
    if t.get("source_hash") != source_hash():
        return {"value": None, "note": "VALU profile %s was taken from other sources (%s, now %s): re-run tools/profile.sh"
                                       % (VALU_PROFILE, t.get("source_hash"), source_hash())}
    stage_of = {"k_transform": "transform", "k_noise": "noisemask", "k_floor": "floor", "k_floor_pair": "floor", "k_couple": "couple", "k_couple_norm": "couple",
                "k_tone_seed": "tonemask", "k_tone_chase": "tonemask", "k_tone_fold": "tonemask"}
    # SIMD cycles per second, whole chip: at the shader clock MEASURED over this run's timed region when there is one
    # (clock_probe below; the chip runs at 2.0-2.43 GHz depending on load), else at the profile's nominal figure
    ghz = clock_ghz or t["clock_ghz"]
    rate = t["simds"] * ghz * 1e9
    per_stage, insts = {}, 0.0
    for k, v in t["per_kernel"].items():
        st = stage_of.get(k.split("<")[0])
        if st is None or (workload == "c3" and st in ("floor", "couple")):
            continue
        d = per_stage.setdefault(st, {"valu_insts_per_unit": 0.0, "issue_ms": 0.0, "lane_insts_per_unit": 0.0})
        d["valu_insts_per_unit"] += v["valu_per_stereo_block"]
        d["lane_insts_per_unit"] += v["valu_per_stereo_block"] * v.get("mean_lanes_live", 0.0)
        d["issue_ms"] += v["valu_per_stereo_block"] * units * v["cycles_per_inst_model"] / rate * 1e3
        insts += v["valu_per_stereo_block"]
    # the tone chain runs beside the noise mask: the pair is one segment of the step as far as issue slots go
    for st, d in per_stage.items():
        ms = stage_ms.get(st)
        if st == "noisemask":
            pair = d["issue_ms"] + per_stage.get("tonemask", {}).get("issue_ms", 0.0)
            both = stage_ms.get("noisemask", 0.0) + stage_ms.get("tonemask", 0.0)
            d["stage_ms"], d["frac_valu"] = both, (pair / both if both else None)
            d["note"] = "with the tone chain, which runs beside it: %.3f ms of issue time in %.3f ms" % (pair, both)
        elif st == "tonemask":
            d["stage_ms"], d["frac_valu"] = ms, None
        else:
            d["stage_ms"], d["frac_valu"] = ms, (d["issue_ms"] / ms if ms else None)
    # lane utilisation: of the 64 lanes a vector instruction could drive, the mean fraction whose exec
    # bit was set -- SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU), calibrated on k_calib_copy (all lanes live: exactly 64)
    lanes = 0.0
    for d in per_stage.values():
        li = d.pop("lane_insts_per_unit")
        lanes += li
        d["lane_utilisation"] = li / (64.0 * d["valu_insts_per_unit"]) if d["valu_insts_per_unit"] else None
    total = sum(d["issue_ms"] for d in per_stage.values())
    return {"valu_insts_per_unit": insts, "lane_utilisation": lanes / (64.0 * insts) if insts else None, "cycles_per_inst": total * 1e-3 * rate / max(insts * units, 1.0), "simds": t["simds"],
            "clock_ghz": ghz, "clock_source": ("measured over the timed region (s_memtime / s_memrealtime)" if clock_ghz else "nominal"),
            "issue_ms_per_step": total, "frac_valu": total / max(sum(stage_ms.values()), 1e-9),
            "per_stage": per_stage, "source": t["source"],
            "definition": "vector instructions per unit (SQ_INSTS_VALU) x modelled SIMD cycles per instruction (the kernel's dynamic "
                          "mix priced with the measured whole-chip cost of each class) / (SIMDs x clock), against the summed stage events"}


VALU_STAGE_OF = {"k_transform": "transform", "k_noise": "noisemask", "k_floor": "floor", "k_floor_pair": "floor", "k_couple": "couple",
                 "k_couple_norm": "couple", "k_tone_seed": "tonemask", "k_tone_chase": "tonemask", "k_tone_fold": "tonemask",
                 "k_tone_seed_chase": "tonemask", "k_ampmax_streams_mixed": "ampmax"}


def measured_valu_c5(units, stage_ms, clock_ghz=None):
    """roofline.valu for the mixed-size workload: one step of this very workload was counted (tools/prof_run_c5.py; a run with
    other stream counts scales by its units).  Stages as the library's events cut them; the block-switching detector and the
    plan (k_env_*, k_plan_*), which have no stage event, are listed under "detector_plan" with their issue time alone."""
    try:
        t = json.load(open(os.path.join(ROOT, VALU_PROFILE_C5)))
    except Exception:
        return {"value": None, "note": "no committed C5 VALU profile"}
    if t.get("source_hash") != source_hash():
        return {"value": None, "note": "VALU profile %s was taken from other sources (%s, now %s): re-run tools/profile.sh"
                                       % (VALU_PROFILE_C5, t.get("source_hash"), source_hash())}
    scale = units / float(t["short_blocks"] + t["long_blocks"])
    ghz = clock_ghz or t["clock_ghz"]
    rate = t["simds"] * ghz * 1e9
    per_stage, insts, lanes = {}, 0.0, 0.0
    for k, v in t["per_kernel"].items():
        base = k.split("<")[0]
        st = VALU_STAGE_OF.get(base, "detector_plan" if base.startswith(("k_env_", "k_plan_")) else None)
        if st is None:
            continue
        d = per_stage.setdefault(st, {"valu_insts_per_step": 0.0, "issue_ms": 0.0, "lane_insts": 0.0})
        n = v["valu_per_step"] * scale
        d["valu_insts_per_step"] += n
        d["issue_ms"] += n * v["cycles_per_inst_model"] / rate * 1e3
        d["lane_insts"] += n * v.get("mean_lanes_live", 0.0)
        insts += n
        lanes += n * v.get("mean_lanes_live", 0.0)
    for st, d in per_stage.items():
        li = d.pop("lane_insts")
        d["lane_utilisation"] = li / (64.0 * d["valu_insts_per_step"]) if d["valu_insts_per_step"] else None
        ms = stage_ms.get(st)
        d["stage_ms"], d["frac_valu"] = ms, (d["issue_ms"] / ms if ms else None)
    if "noisemask" in per_stage and "tonemask" in per_stage:   # the tone chain runs beside the noise mask: one segment of the step
        both = (stage_ms.get("noisemask") or 0.0) + (stage_ms.get("tonemask") or 0.0)
        pair = per_stage["noisemask"]["issue_ms"] + per_stage["tonemask"]["issue_ms"]
        per_stage["noisemask"]["stage_ms"], per_stage["noisemask"]["frac_valu"] = both, (pair / both if both else None)
        per_stage["noisemask"]["note"] = "with the tone chain, which runs beside it"
        per_stage["tonemask"]["frac_valu"] = None
    total = sum(d["issue_ms"] for d in per_stage.values())
    return {"valu_insts_per_unit": insts / max(units, 1), "lane_utilisation": lanes / (64.0 * insts) if insts else None, "simds": t["simds"],
            "clock_ghz": ghz, "clock_source": ("measured over the timed region (s_memtime / s_memrealtime)" if clock_ghz else "nominal"),
            "issue_ms_per_step": total, "per_stage": per_stage, "source": t["source"],
            "definition": "vector instructions of one step (SQ_INSTS_VALU x waves per kernel) x modelled SIMD cycles per instruction / "
                          "(SIMDs x clock); frac_valu per stage against that stage's event; the whole step's: issue_ms_per_step / ms_per_step"}


def measured_traffic(workload, units, alg_bytes=None):
    """HBM bytes per step from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, FETCH_SIZE doubled per the gfx950 calibration; tools/profile.sh + tools/make_profiles.py).  The
    profile carries the hash of the sources it was taken from; if the sources have changed since, the figure
    is withheld (None, {}, reason) rather than reported stale.  Returns (bytes per step, per-kernel dict, note)."""
    prof = TRAFFIC_PROFILE_C5 if workload == "c5" else TRAFFIC_PROFILE
    try:
        t = json.load(open(os.path.join(ROOT, prof)))
    except Exception:
        return None, {}, "no committed PMC profile"
    if t.get("source_hash") != source_hash():
        return None, {}, "PMC profile %s was taken from other sources (%s, now %s): re-run tools/profile.sh" % (
            prof, t.get("source_hash"), source_hash())
    note = "%s (source hash %s matches this build)" % (t["source"], t["source_hash"])
    if workload == "c2":
        return t["mdct_only_B_per_frame"] * units, {}, note
    if workload == "c5":
        # one step of this very workload was counted; a run with other stream counts scales by its algorithmic bytes
        scale = (alg_bytes / t["alg_bytes_per_step"]) if alg_bytes else 1.0
        per5 = {k: (v["read_B_per_step"] + v["write_B_per_step"]) * scale for k, v in t["per_kernel"].items()}
        return sum(per5.values()), per5, note
    per = {k: (v["read_B_per_stereo_block"] + v["write_B_per_stereo_block"]) * units for k, v in t["per_kernel"].items()}
    if workload == "c3":
        per = {k: v for k, v in per.items() if k in ("k_transform", "k_noise", "k_tone_seed", "k_tone_chase", "k_tone_fold", "k_tone")}
    return sum(per.values()), per, note


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("c2", "c3", "c4", "c5"), default="c4")
    ap.add_argument("--blocks", type=int, default=None, help="stereo blocks per GPU (default 131072; c2/c3: 65536)")
    ap.add_argument("--setup", default=None, help="setup blob name (default 44k_stereo_q4; c5: 44k_stereo_q9)")
    ap.add_argument("--streams", type=int, default=1024, help="c5: streams per GPU")
    ap.add_argument("--stream-samples", type=int, default=131072, help="c5: samples per channel and stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-neighbours", action="store_true", help="skip the informational extra stages (profiling runs)")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip the post-run oracle check of the timed batch")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not sample the shader clock beside the timed steps")
    ap.add_argument("--host-fed-only", default=None, help="development aid: run only host_fed for 'c4' or 'c5' and print its dict")
    ap.add_argument("--feed-streams", type=int, default=512, help="host_fed: streams per group")
    ap.add_argument("--feed-groups", type=int, default=100, help="host_fed: groups in the timed region")
    ap.add_argument("--feed-lanes", type=int, default=5, help="host_fed: groups in flight")
    ap.add_argument("--no-host-fed", action="store_true", help="default run only: skip the host-fed (PCIe-inclusive) figures")
    ap.add_argument("--no-workloads", action="store_true",
                    help="default (c4) run only: skip the other BASELINE configs (c2, c3, c5) that are measured after the headline")
    ap.add_argument("--parity-blocks", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for the CPU rehearsal of the rank logic)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the multi-rank path on ONE device: every rank runs the real GPU runner on cuda:0 (with "
                         "--backend gloo).  Exercises the blob broadcast from a device tensor, per-rank seeds, shard ranges, "
                         "per-rank parity samples and the aggregated line -- everything but RCCL itself; the line says share_gpu")
    ap.add_argument("--runner", default=None,
                    help="test aid: 'module:Class' of a runner that replaces the GPU runner (the CPU rehearsal of the rank logic "
                         "names a stub that does no analysis; the line then carries \"runner\" and is not a measurement)")
    a = ap.parse_args(argv)
    if a.setup is None:
        a.setup = "44k_stereo_q9" if a.workload == "c5" else "44k_stereo_q4"
    return a


def host_cpus():
    """What this process may use: (hardware threads the OS shows, threads in the affinity mask, CPUs the cgroup's
    quota allows or None).  A container is often held to a CPU-time quota far below the threads it sees -- round 3's
    "771 blocks/s per core at 128 threads against 7 769 on one" was 128 threads time-slicing a 16-CPU quota."""
    seen = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = seen
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return seen, aff, quota


def cpu_baseline(setup_name, seconds):
    """Reference libvorbis (oracle/_ref, unmodified sources) -- or the C port if the prebuilt reference is absent --
    timed on this box's host cores on a bounded sample of the same white-noise workload: one thread alone, then two
    legs over all the CPUs this process may use -- threads in this process (one encoder state each) and one PROCESS
    per CPU (tools/cpu_ref_worker.py; the reference's per-block arena, lib/block.c:102-146, mallocs and frees several
    KB per block, and separate address spaces take allocator locks out of the picture).  The better leg is the figure
    reported, with its scaling efficiency against the single thread.  Reported, not a target."""
    import subprocess
    from tests import checker
    from oracle import ref
    ch, rate, q = checker.SETUPS[setup_name]
    seen, aff, quota = host_cpus()
    usable = max(1, min(aff, int(quota + 0.5) if quota else aff))
    sample_blocks = 256
    rng = np.random.default_rng(99)
    pcm = (rng.random((sample_blocks, ch, 2048), dtype=np.float32) - 0.5).astype(np.float32)
    if ref.available():
        kind = "reference"
        make = lambda: ref.RefEncoder(ch, rate, q)  # noqa: E731
    else:
        from oracle import port
        kind = "port"
        blob = np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % setup_name), dtype=np.uint8)
        make = lambda: port.PortEncoder(blob)  # noqa: E731
    nthreads = min(usable, 256)
    encs = [make() for _ in range(nthreads)]
    # one thread on an otherwise idle host first: what a core does unshared
    t1 = time.time()
    n1 = 0
    while time.time() - t1 < min(2.0, seconds / 5):
        encs[0].time_dsp(pcm, 1)
        n1 += sample_blocks
    single = n1 / (time.time() - t1)
    leg_s = (seconds - min(2.0, seconds / 5)) / 2
    legs = []
    # leg 1: threads of this process (ctypes releases the GIL inside the library)
    done = [0] * nthreads
    deadline = time.time() + leg_s

    def work(i):
        e = encs[i]
        while time.time() < deadline:
            e.time_dsp(pcm, 1)
            done[i] += sample_blocks

    t0 = time.time()
    c0 = time.process_time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.time() - t0
    legs.append({"kind": "threads", "workers": nthreads, "value": sum(done) / wall, "blocks": sum(done), "wall": wall,
                 "cpu_seconds": time.process_time() - c0})
    # leg 2: one process per usable CPU
    try:
        start = time.time() + 2.5   # every worker is up (numpy + ctypes only) and waiting by then
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_ref_worker.py"), setup_name, repr(start),
                                   repr(leg_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(usable)]
        outs = [p.communicate(timeout=leg_s + 60)[0].split() for p in procs]
        blocks = sum(int(o[0]) for o in outs if len(o) == 3)
        walls = [float(o[1]) for o in outs if len(o) == 3]
        cpus = sum(float(o[2]) for o in outs if len(o) == 3)
        if walls:
            legs.append({"kind": "processes", "workers": len(walls), "value": blocks / max(walls), "blocks": blocks,
                         "wall": max(walls), "cpu_seconds": cpus})
    except Exception as e:  # the thread leg stands on its own
        legs.append({"kind": "processes", "workers": 0, "value": 0.0, "blocks": 0, "wall": 0.0, "cpu_seconds": 0.0, "error": repr(e)})
    best = max(legs, key=lambda g: g["value"])
    return {
        "value": best["value"], "unit": "stereo blocks/s", "cores": best["workers"], "kind": kind, "leg": best["kind"],
        "single_thread_value": single, "scaling_efficiency": best["value"] / (single * best["workers"]),
        "per_cpu_second": best["blocks"] / max(best["cpu_seconds"], 1e-9),
        "host": {"hardware_threads": seen, "affinity": aff, "cgroup_cpu_quota": quota, "usable_cpus": usable},
        "legs": [{"kind": g["kind"], "workers": g["workers"], "value": g["value"],
                  "cpus_busy": g["cpu_seconds"] / max(g["wall"], 1e-9)} for g in legs],
        "sample": "%d %s x repeated passes over seeded white-noise stereo 2048-blocks for %.0f s wall (%d blocks total); "
                  "window+MDCT+FFT+psy+floor1 fit/encode+couple/quantise of mapping0_forward (the part the GPU path "
                  "computes; residue VQ/Huffman excluded)" % (best["workers"], best["kind"], best["wall"], best["blocks"]),
    }


def neighbour_stages(an, pcm, outs, nb):
    """Outside the timed region and outside the metric: throughput of the analysis with the residue
    back-end's search enabled, of the whole block encode (PCM in, packets out), and of the block-switching
    detector on a slice of the same samples."""
    import vorbis_amd
    res = dict(outs)
    res.update(an.alloc_outputs(1, nb, ("res_class", "res_entries", "res_count")))
    an.analyze(pcm, outs=res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        an.analyze(pcm, outs=res)
    torch.cuda.synchronize()
    with_res = 3 * nb / (time.perf_counter() - t0)
    entries = float(res["res_count"][:, 1].float().mean().item())
    del res
    # PCM in, finished packets out (residue search + packet assembly; intermediate tensors internal)
    pk = an.alloc_outputs(1, nb, ("ampmax_out", "packets", "packet_bits"))
    an.analyze(pcm, outs=pk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        an.analyze(pcm, outs=pk)
    torch.cuda.synchronize()
    to_packets = 3 * nb / (time.perf_counter() - t0)
    packet_bytes = float(pk["packet_bits"].float().mean().item()) / 8
    del pk
    ns = 256
    streams = pcm[: ns * 64].reshape(ns, 64, pcm.shape[1], pcm.shape[2]).permute(0, 2, 1, 3).reshape(ns, pcm.shape[1], -1)
    streams = streams.contiguous()                      # 256 streams of 64 blocks' samples
    win, step = an.envelope_geometry()
    steps = (streams.shape[2] - win) // step + 1
    ret, st = an.envelope_search_batch(streams, steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        st.zero_()
        an.envelope_search_batch(streams, steps, states=st, ret=ret)
    torch.cuda.synchronize()
    det = 3 * ns * steps / (time.perf_counter() - t0)
    # the 5.1 layout (six channels, two submaps, four coupling steps), PCM in, packets out
    surround = None
    try:
        an6 = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_51_q3"), an.device)
        nb6 = max(1024, nb // 8)
        pcm6 = torch.rand((nb6, 6, an6.blocksizes[1]), device=pcm.device) - 0.5
        pk6 = an6.alloc_outputs(1, nb6, ("ampmax_out", "packets", "packet_bits"))
        an6.analyze(pcm6, outs=pk6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            an6.analyze(pcm6, outs=pk6)
        torch.cuda.synchronize()
        surround = {"value": 3 * nb6 / (time.perf_counter() - t0), "unit": "six-channel blocks/s (q 0.3 tables, PCM to packets)",
                    "mean_packet_bytes": float(pk6["packet_bits"].float().mean().item()) / 8}
        del pk6, pcm6
        an6.close()
    except Exception as e:  # informational only
        surround = {"error": repr(e)}
    return {"surround_5_1": surround, "analysis_with_residue_search": {"value": with_res, "unit": "stereo blocks/s", "mean_entries_per_block": entries},
            "pcm_to_packets": {"value": to_packets, "unit": "stereo blocks/s", "mean_packet_bytes": packet_bytes},
            "block_switching_detector": {"value": det, "unit": "stereo detector steps/s (one per 64 samples)",
                                         "streams": ns, "steps_per_stream": int(steps)}}


class GpuRunner:
    """One rank's share of the workload on its GPU: owns the context, the resident inputs and outputs."""

    def __init__(self, a, blob, dev, rank, world):
        import vorbis_amd
        self.a, self.dev, self.rank = a, dev, rank
        self.vorbis_amd = vorbis_amd
        self.an = an = vorbis_amd.Analyzer(blob, device=dev.index)
        ch, n = an.channels, an.blocksizes[1]
        nb = a.blocks or (131072 if a.workload == "c4" else 65536)
        # weak scaling: `nb` units per rank; rank r owns [lo, hi) of the job's nb * world units
        self.lo, self.hi = sharding.shard_range(nb * world, rank, world)
        self.nb = nb = self.hi - self.lo
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        if a.workload == "c2":
            self.frames = torch.rand((nb, n), generator=g, device=dev, dtype=torch.float32) - 0.5
            self.out = torch.empty((nb, n // 2), device=dev, dtype=torch.float32)
            self.units, self.unit_name = nb, "2048-sample frames/s"
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        else:
            self.pcm = torch.rand((nb, ch, n), generator=g, device=dev, dtype=torch.float32) - 0.5
            self.level = vorbis_amd.LEVEL_FULL if a.workload == "c4" else vorbis_amd.LEVEL_PSY
            want = ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out") if a.workload == "c4" \
                else ("mdct_raw", "noise", "tone")
            self.outs = an.alloc_outputs(1, nb, want)
            an.reserve(1, nb)
            self.units, self.unit_name = nb, "stereo blocks/s"

    def step(self):
        if self.a.workload == "c2":
            self.an.mdct_forward(1, self.frames, out=self.out)
        else:
            self.an.analyze(self.pcm, W=1, lW=1, nW=1, blocktype=1, ampmax_in=-9999.0, level=self.level, outs=self.outs)

    def sync(self):
        torch.cuda.synchronize()

    def timed_begin(self):
        if self.a.workload != "c2":
            self.an.profile(True)
        else:
            self.ev0.record()

    def timed_end(self):
        if self.a.workload == "c2":
            self.ev1.record()

    def stage_bytes_total(self, stage):
        if self.a.workload == "c2":
            return ALG_BYTES["c2"] * self.units
        return stage_bytes(stage, self.an.blocksizes[1], self.an.channels) * self.units

    def stage_ms(self, steps):
        """per-kernel durations from HIP events recorded on the launch stream inside the timed region"""
        if self.a.workload != "c2":
            ms, runs = self.an.stage_ms()
            self.an.profile(False)
            return {k: v / max(runs, 1) for k, v in ms.items() if v > 0}
        return {"mdct_forward": self.ev0.elapsed_time(self.ev1) / steps}

    def parity_sample(self, count):
        """After the timed region: `count` randomly indexed units of the very batch that was timed, every output
        the run produced for them compared bit-for-bit with the CPU checker (oracle/_ref when its prebuilt library
        is here, else the pinned C restatement).  Returns (units checked, mismatching units, checker kind)."""
        from tests import checker
        chk = checker.Checker(self.a.setup)
        rng = np.random.default_rng(4242 + self.rank)
        idx = np.sort(rng.choice(self.nb, size=min(count, self.nb), replace=False))
        sel = torch.from_numpy(idx).to(self.dev)
        bad = 0
        if self.a.workload == "c2":
            x, y = self.frames[sel].cpu().numpy(), self.out[sel].cpu().numpy()
            for k in range(len(idx)):
                bad += not np.array_equal(chk.mdct_forward(1, x[k]).view(np.uint32), y[k].view(np.uint32))
            return len(idx), bad, chk.kind
        pcm = self.pcm[sel].cpu().numpy()
        got = {k: v[sel].cpu().numpy() for k, v in self.outs.items()}
        nposts = self.an.posts[1]
        for k in range(len(idx)):
            ref = chk.tap_block(pcm[k])
            bad += checker.compare_block(ref, {kk: v[k] for kk, v in got.items()}, nposts) != 0
        return len(idx), bad, chk.kind

    def workload_text(self):
        nb = self.nb
        return {"c4": "C4 full mapping0_forward analysis (window+MDCT+FFT+noise/tone mask+floor1 fit+"
                      "couple/quantise), %d stereo 2048-blocks per GPU, 44.1 kHz q=0.4 tables, "
                      "white noise, independent frames, inputs resident in HBM" % nb,
                "c3": "C3 MDCT + _vp_noisemask/_vp_tonemask, %d stereo 2048-blocks per GPU" % nb,
                "c2": "C2 batched mdct_forward only, %d x n=2048 frames per GPU" % nb}[self.a.workload]

    def neighbours(self):
        return neighbour_stages(self.an, self.pcm, self.outs, self.nb)


class StreamRunner:
    """BASELINE config 5: many gated-noise streams resident in HBM; a step plans them on the device (block-switching
    detector + the blockout decisions, vamd_plan_streams), gathers the planned blocks of both sizes and analyses them
    with each stream's ampmax chain (vamd_analyze_streams_mixed).  Units = blocks of either size."""

    def __init__(self, a, blob, dev, rank, world):
        import vorbis_amd
        self.a, self.dev, self.rank = a, dev, rank
        self.an = an = vorbis_amd.Analyzer(blob, device=dev.index)
        ch = an.channels
        ns, ln = a.streams, a.stream_samples & ~3
        g = torch.Generator(device=dev)
        g.manual_seed(4321 + rank)
        # noise gated by bursts at stream-dependent periods: bursts trigger short blocks, the quiet stretches run long
        t = torch.arange(ln, device=dev)
        period = (6000 + 977 * torch.arange(ns, device=dev) % 9000).view(ns, 1)
        gate = torch.where((t.view(1, ln) % period) < 600, 0.5, 0.0005).view(ns, 1, ln)
        self.streams = ((torch.rand((ns, ch, ln), generator=g, device=dev) - 0.5) * 2 * gate).contiguous()
        self.want = ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")
        self.unit_name = "stereo blocks/s (256- and 2048-sample)"
        self._run(alloc=True)
        self.units = int(self.plan.nblocks[0] + self.plan.nblocks[1])

    def _run(self, alloc=False):
        an = self.an
        self.plan, _ = an.plan_streams(self.streams)
        if alloc:
            self.outs = [an.alloc_outputs(W, self.plan.nblocks[W], self.want) for W in (0, 1)]
            for W in (0, 1):
                an.reserve(W, max(1, self.plan.nblocks[W]))
            self.amp = torch.empty(self.streams.shape[0], device=self.dev)
        self.amp.fill_(-9999.0)
        # the planned blocks are analysed where they lie in the stream buffers (the plan's offsets): no gathered copy
        an.analyze_plan(self.plan, None, self.outs, self.amp, streams=self.streams)

    def step(self):
        self._run()

    def sync(self):
        torch.cuda.synchronize()

    def timed_begin(self):
        self.an.profile(True)

    def timed_end(self):
        pass

    def stage_ms(self, steps):
        ms, runs = self.an.stage_ms()
        self.an.profile(False)
        return {k: v / max(runs, 1) for k, v in ms.items() if v > 0}

    def alg_bytes(self):
        """"if the input is read as a 50 %-overlapped stream rather than pre-cut frames, PCM-in halves; report
        which".  The streams are analysed where they lie (vamd_batch_io::pcm_src): every sample is charged ONCE -- the
        stream buffers' bytes -- plus the per-block outputs of 8d (long: mdct + logmask + iwork 3 x 8192 + 256; short:
        3 x 2 x 512 + 128)."""
        ns, ch, ln = self.streams.shape
        return ns * ch * ln * 4 + 3200 * int(self.plan.nblocks[0]) + 24832 * int(self.plan.nblocks[1])

    def alg_rule(self):
        return ("overlapped streams: PCM-in = the stream buffers once (%d B), not 2048 / 16384 B per pre-cut short / long "
                "block; outputs 3200 / 24832 B per short / long block (SURVEY.md 8d)" % (self.streams.numel() * 4))

    def stage_bytes_total(self, stage):
        an = self.an
        return sum(stage_bytes(stage, an.blocksizes[W], an.channels) * int(self.plan.nblocks[W]) for W in (0, 1))

    def parity_sample(self, count):
        """Randomly chosen planned blocks: window flags, block type and every output against the CPU checker fed the
        gathered block and the chain's incoming ampmax (decay of the previous block's, from the timed outputs)."""
        from tests import checker
        chk = checker.Checker(self.a.setup)
        L = self.an.plan_lists(self.plan)
        rng = np.random.default_rng(99 + self.rank)
        order, start = L["order"], L["stream_start"]
        pick = np.sort(rng.choice(len(order), size=min(count, len(order)), replace=False))
        bad = 0
        amp_out = [self.outs[W]["ampmax_out"].cpu().numpy() for W in (0, 1)]
        # the sampled blocks' samples and outputs come over in one copy per tensor and size class (after the clock has
        # stopped; round 3 fetched them block by block -- two thousand small copies that a kernel trace then showed)
        idx = [[], []]
        for k in pick:
            idx[(int(order[k]) >> 30) & 1].append(int(order[k]) & 0x3fffffff)
        got, pcm, where = [None, None], [None, None], [{}, {}]
        for W in (0, 1):
            if not idx[W]:
                continue
            sel = torch.tensor(idx[W], device=self.dev, dtype=torch.int64)
            blocks = self.an.gather_blocks(self.plan, W, self.streams)   # (the checker's copy of the blocks, not the timed path's)
            pcm[W] = blocks[sel].cpu().numpy()
            del blocks
            got[W] = {kk: v[sel].cpu().numpy() for kk, v in self.outs[W].items()}
            where[W] = {i: j for j, i in enumerate(idx[W])}
        for k in pick:
            s = int(np.searchsorted(start, k, side="right") - 1)
            W, i = (int(order[k]) >> 30) & 1, int(order[k]) & 0x3fffffff
            prev = -9999.0
            if k > start[s]:
                pW, pi = (int(order[k - 1]) >> 30) & 1, int(order[k - 1]) & 0x3fffffff
                prev = float(amp_out[pW][pi])
            amp_in = chk.enc.ampmax_decay(prev, W)
            j = where[W][i]
            ref = chk.tap_block(pcm[W][j], int(L["lW"][W][i]), W, int(L["nW"][W][i]), int(L["blocktype"][W][i]), amp_in)
            bad += checker.compare_block(ref, {kk: v[j] for kk, v in got[W].items()}, self.an.posts[W]) != 0
        return len(pick), bad, chk.kind

    def workload_text(self):
        return ("C5 mixed short/long-block streams: %d gated-noise stereo streams x %d samples per GPU, 44.1 kHz q=0.9 tables; per step: "
                "block-switching detector + blockout decisions on the device (vamd_plan_streams), full analysis of the "
                "%d short and %d long blocks where they lie in the stream buffers, with per-stream ampmax chains" % (self.streams.shape[0], self.streams.shape[2],
                                                                               self.plan.nblocks[0], self.plan.nblocks[1]))


def host_fed(setup, kind, device, streams_per_group=512, frames=131072, groups=100, lanes=5, parity_streams=2, seed=0):
    """The H2D/D2H-inclusive figure: whole streams from PINNED HOST memory as 16-bit interleaved samples
    in, finished packets back in host memory, through vamd_feed (include/vorbis_amd.h) -- upload, 16-bit -> float, both
    stream ends, detector, block walk, full analysis, residue search, packet assembly and the packets' way home all inside
    the clock, `lanes` groups in flight so that one group's upload runs beside another's kernels.  kind "c4": white noise
    (long blocks only: the C4 shape as streams); "c5": the gated noise of config 5 (mixed sizes).  The clock runs from
    the first vamd_feed_wrote() to the last group's packets, after one untimed group per lane (allocations).  Returns the
    dict that goes into the line as host_fed[kind]; its parity sample re-encodes `parity_streams` of the timed streams with
    the reference's own application loop fed the same 16-bit samples."""
    import vorbis_amd
    blob = vorbis_amd.default_setup_blob(setup)
    ch = 2
    feed = vorbis_amd.Feed(blob, devices=[device], lanes_per_device=lanes, max_streams=streams_per_group, max_frames=frames, fmt=vorbis_amd.FEED_S16)
    rng = np.random.default_rng(20260 + (kind == "c5") + 17 * seed)
    n = streams_per_group * frames * ch
    # one synthetic group per lane, generated straight into the lane's pinned arena (a caller's decoder writes there)
    slots = []
    for _ in range(lanes):
        slot, buf = feed.buffer(ch)
        x = rng.random(n, dtype=np.float32) - np.float32(0.5)
        if kind == "c5":
            t = np.arange(frames, dtype=np.int64)
            period = (6000 + 977 * np.arange(streams_per_group, dtype=np.int64) % 9000)[:, None]
            gate = np.where((t[None, :] % period) < 600, 1.0, 0.001).astype(np.float32)
            x = (x.reshape(streams_per_group, frames, ch) * gate[:, :, None]).reshape(-1)
        buf[:n] = np.round(x * np.float32(32767.0)).astype(np.int16)
        if not slots:
            keep_pcm = buf[:n].reshape(streams_per_group, frames, ch)[:parity_streams].copy()
        slots.append(slot)
    for slot in slots:                                  # untimed: every lane allocates its HBM and grows its arenas
        feed.wrote(slot, streams_per_group, frames)
    keep = None
    for slot in slots:
        r = feed.packets(slot, copy=False)
        if keep is None:
            keep = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in r.items()}
    # the timed run: `groups` groups, at most `lanes` in flight
    t0 = time.perf_counter()
    inflight, blocks, out_bytes, up_ms, dev_ms, issued = [], 0, 0, 0.0, 0.0, 0
    for slot in slots:
        feed.release(slot)
    while issued < groups or inflight:
        while issued < groups and len(inflight) < lanes:
            slot, _ = feed.buffer(ch)          # (the lane's arena still holds its samples: a caller would refill it here)
            feed.wrote(slot, streams_per_group, frames)
            inflight.append(slot)
            issued += 1
        slot = inflight.pop(0)
        r = feed.packets(slot, copy=False)
        blocks += r["nblocks"]
        out_bytes += r["total_bytes"]
        up_ms += r["upload_ms"]
        dev_ms += r["device_ms"]
        feed.release(slot)
    elapsed = time.perf_counter() - t0
    in_bytes = groups * n * 2
    d = {"value": blocks / elapsed, "unit": "stereo blocks/s, host s16 in -> host packets out", "seconds": elapsed, "groups": groups,
         "streams_per_group": streams_per_group, "frames_per_stream": frames, "lanes": lanes, "blocks": int(blocks),
         "input": "%d B/s16 interleaved stereo from pinned host memory, %.1f MB per group" % (2, n * 2 / 1e6),
         "output": "packets end to end in pinned host memory, %.1f bytes per block" % (out_bytes / max(blocks, 1)),
         "pcie_GBps": {"up_sustained": in_bytes / elapsed / 1e9, "up_while_copying": in_bytes / max(up_ms * 1e-3, 1e-9) / 1e9,
                       "down_sustained": out_bytes / elapsed / 1e9},
         "device_ms_per_group": dev_ms / groups, "upload_ms_per_group": up_ms / groups, "setup": setup}
    # parity: the reference's application loop over the same 16-bit samples
    try:
        from tests import checker
        from oracle import ref
        if ref.available() and keep is not None and keep_pcm is not None:
            chn, rate, q = checker.SETUPS[setup]
            bad = npk = 0
            for s_ in range(parity_streams):
                planar = np.ascontiguousarray((keep_pcm[s_].astype(np.float32) / np.float32(32768.0)).T)
                want = ref.RefEncoder(chn, rate, q).encode_stream(planar)
                lo, hi = int(keep["stream_start"][s_]), int(keep["stream_start"][s_ + 1])
                npk += len(want)
                if hi - lo != len(want):
                    bad += abs(hi - lo - len(want)) + 1
                    continue
                for k, w in enumerate(want):
                    o, bits = int(keep["offset"][lo + k]), int(keep["bits"][lo + k])
                    bad += bytes(keep["bytes"][o:o + (bits + 7) // 8]) != w["packet"] or int(keep["granulepos"][lo + k]) != w["granulepos"]
            d["parity_sample"] = {"streams": parity_streams, "packets": npk, "mismatches": int(bad), "checker": "reference",
                                  "compared": "every packet of whole streams (bytes, granulepos), first block to last, against the "
                                              "reference's application loop fed the same 16-bit samples"}
    except Exception as e:
        d["parity_sample"] = {"packets": 0, "mismatches": None, "error": repr(e)}
    feed.close()
    return d


def host_fed_all(a, dev, rank, world):
    """host_fed on every rank's own device at once (each GPU has its own link; the host's memory system is shared), the
    job's figure = all ranks' blocks / the slowest rank's seconds.  One rank: the C4 shape and the C5 shape; more: C4."""
    out = {}
    for kind, setup in (("c4", "44k_stereo_q4"), ("c5", "44k_stereo_q9")):
        if world > 1 and kind != "c4":
            continue
        try:
            sharding.barrier()
            d = host_fed(setup, kind, dev.index or 0, streams_per_group=a.feed_streams, frames=a.stream_samples, groups=a.feed_groups,
                         lanes=a.feed_lanes, seed=rank)
            err = 0
        except Exception as e:
            d, err = {"error": repr(e), "blocks": 0, "seconds": 0.0}, 1
        blocks = sharding.sum_over_ranks(d.get("blocks", 0), dev)
        secs = sharding.max_over_ranks(d.get("seconds", 0.0), dev)
        bad = sharding.sum_over_ranks(err, dev)
        if world > 1 and not bad:
            d["rank0_value"] = d["value"]
            d["value"] = blocks / max(secs, 1e-9)
            d["blocks"], d["seconds"], d["n_gpus"] = blocks, secs, world
        out[kind] = d
    return out


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` with no process group in the environment: launch the N ranks ourselves, exactly as
    the driver does (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), and hand back their exit
    code.  Rank 0 of the children prints the line."""
    import subprocess
    # --standalone: torch.distributed.run opens the rendezvous store itself on a port of its own choosing (no
    # pick-a-free-port-then-hope race); --local-addr keeps it on 127.0.0.1 (the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), os.path.abspath(__file__)] + list(argv)
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT")}
    return subprocess.call(cmd, env=env)


class ClockProbe:
    """The shader clock over a timed region: vamd_clock_probe -- the first wave of every step's tone
    stack walk, which runs beside the noise mask, samples s_memtime against s_memrealtime over its own life."""

    def __init__(self, R, dev):
        self.an = getattr(R, "an", None)
        self.acc = torch.zeros(3, dtype=torch.int64, device=dev) if (self.an is not None and hasattr(self.an, "clock_probe")) else None

    def begin(self):
        if self.acc is not None:
            try:
                self.an.clock_probe(self.acc)
            except Exception:
                self.acc = None

    def result(self):
        """{"ghz", "samples"} or None; call after a device synchronise"""
        if self.acc is None:
            return None
        torch.cuda.synchronize()
        self.an.clock_probe(None)
        t, w, n = (int(x) for x in self.acc.cpu().tolist())
        if n == 0 or w == 0:
            return None
        return {"ghz": t / w * 0.1, "samples": n,
                "method": "s_memtime ticks per s_memrealtime (100 MHz) tick over the life of the first wave of each step's "
                          "tone stack walk (k_tone_chase: beside the noise mask)"}


def timed_run(a, R, dev, world, probe=None):
    """Warm up, then time EXACTLY a.steps steps between barriers + device synchronisations; max over ranks.
    Returns (elapsed seconds, per-stage ms per step from HIP events on the launch stream)."""
    for _ in range(a.warmup):
        R.step()
    R.sync()
    sharding.barrier()
    R.sync()
    t0 = time.perf_counter()
    R.timed_begin()
    if probe is not None:
        probe.begin()
    for _ in range(a.steps):
        R.step()
    R.timed_end()
    R.sync()
    sharding.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    return elapsed, R.stage_ms(a.steps)


def parity_of(a, R, dev, world, count):
    try:
        n_chk, n_bad, kind = R.parity_sample(max(1, count // world))
        return {"blocks": sharding.sum_over_ranks(n_chk, dev), "mismatches": sharding.sum_over_ranks(n_bad, dev),
                "checker": kind, "compared": "every output tensor of the timed batch for randomly indexed units, bit-exact"}
    except Exception as e:  # a missing checker must not lose the GPU number -- but it is said, not hidden
        sharding.sum_over_ranks(0, dev), sharding.sum_over_ranks(0, dev)
        return {"blocks": 0, "mismatches": None, "error": repr(e)}


def roofline_of(a, R, stage_ms, clock, ms_per_step):
    """frac = algorithmic bytes of one step / the step's WALL time (ms_per_step, the same clock `value` is quoted on) / peak:
    alg_bytes / ms_per_step / peak reproduces it for every workload of the line.  (Until round 5 the divisor was the sum of
    the stage events, which for C5 leaves out the detector and the plan -- VERDICT r05 weak 3.)  The stage events stay in
    the line as kernels_ms_per_step."""
    units = R.units
    kernels_ms = sum(stage_ms.values())
    dom = max(stage_ms, key=stage_ms.get)
    alg = R.alg_bytes() if a.workload == "c5" else ALG_BYTES[a.workload] * units   # bytes per step per GPU, algorithmic
    achieved = alg / (ms_per_step * 1e-3) / 1e9               # GB/s over the step
    dom_bytes = R.stage_bytes_total(dom)
    traffic, traffic_per, traffic_note = measured_traffic(a.workload, units, alg)
    stage_of = {"k_transform": "transform", "k_noise": "noisemask", "k_floor": "floor", "k_floor_pair": "floor", "k_couple": "couple", "k_couple_norm": "couple",
                "k_tone_seed": "tonemask", "k_tone_chase": "tonemask", "k_tone_fold": "tonemask", "k_tone": "tonemask"}
    dom_traffic = sum(v for k, v in traffic_per.items() if stage_of.get(k.split("<")[0]) == dom) or None
    d = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
        "alg_bytes_per_step": alg,
        "definition": "algorithmic bytes of the whole path per step (%d B over %d units) / the step's wall time "
                      "(ms_per_step) / peak" % (alg, units),
        "kernels_ms_per_step": stage_ms, "kernels_ms_sum": kernels_ms,
        "dominant_kernel": {"name": dom, "ms": stage_ms[dom], "own_bytes_per_step": dom_bytes,
                            "own_GBps": dom_bytes / (stage_ms[dom] * 1e-3) / 1e9, "traffic": dom_traffic},
        "valu": measured_valu(a.workload, units, stage_ms, clock["ghz"] if clock else None),
    }
    if a.workload == "c5":
        d["alg_bytes_rule"] = R.alg_rule()
        if d["valu"] and d["valu"].get("issue_ms_per_step"):
            d["valu"]["frac_valu"] = d["valu"]["issue_ms_per_step"] / ms_per_step   # (the step's wall time: detector and plan included)
    return d


# the other BASELINE configs, measured after the headline of a default run at their BASELINE sizes (VERDICT r04 next 4:
# "put every BASELINE config in the driver's one line"): (workload, steps, warmup)
# steps chosen so that every timed region lasts >= 0.5 s (c2 alone measured 333 M, inside a 40 ms
# region 370 M): c2 0.2 ms a step, c3 ~4.9, c5 ~11
EXTRA_WORKLOADS = (("c2", 3000, 100), ("c3", 120, 5), ("c5", 50, 3))


def extra_workloads(a, blobs, dev, rank, world):
    """c2, c3 and c5 after the headline: value, ms per step, roofline (frac, traffic) and a 64-unit parity sample each.
    Every rank runs them (their timing is the job's: barrier, max over ranks); with more than one rank only c5 -- the
    BASELINE config that is quoted on 8 GPUs beside c4."""
    out = {}
    for w, steps, warmup in EXTRA_WORKLOADS:
        if world > 1 and w != "c5":
            continue
        t_w = time.perf_counter()
        b = argparse.Namespace(**vars(a))
        b.workload, b.steps, b.warmup, b.blocks = w, steps, warmup, None
        b.setup = "44k_stereo_q9" if w == "c5" else "44k_stereo_q4"
        try:
            R = (StreamRunner if w == "c5" else GpuRunner)(b, blobs(b.setup), dev, rank, world)
            probe = None if a.no_clock_probe else ClockProbe(R, dev)
            elapsed, stage_ms = timed_run(b, R, dev, world, probe)
            clock = probe.result() if probe else None
            d = {"value": world * R.units * steps / elapsed, "unit": R.unit_name, "ms_per_step": elapsed / steps * 1e3,
                 "steps": steps, "warmup": warmup, "units_per_gpu": R.units, "setup": b.setup, "workload": R.workload_text(),
                 "roofline": roofline_of(b, R, stage_ms, clock, elapsed / steps * 1e3)}
            if clock:
                d["shader_clock"] = clock
            if not a.no_parity_sample:
                d["parity_sample"] = parity_of(b, R, dev, world, 64)
            if hasattr(R, "an"):
                R.an.close()
            del R
            torch.cuda.empty_cache()
        except Exception as e:  # informational beside the headline: an error is said, the headline stands
            d = {"error": repr(e)}
        d["seconds"] = time.perf_counter() - t_w
        out[w] = d
    return out


def main(argv=None, make_runner=None):
    """`make_runner(a, blob, dev, rank, world)` replaces the GPU runner; only the CPU rehearsal of the rank logic
    (tests/test_abi_and_host.py, gloo, world size 2) passes one (or names one with --runner)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.host_fed_only:
        print(json.dumps(host_fed("44k_stereo_q9" if a.host_fed_only == "c5" else "44k_stereo_q4", a.host_fed_only, 0,
                                  streams_per_group=a.feed_streams, frames=a.stream_samples, groups=a.feed_groups, lanes=a.feed_lanes)))
        return 0
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        return spawn_ranks(a, argv)           # --gpus N means N ranks: launch them
    if env_world is not None and a.gpus == 1 and not any(x == "--gpus" or x.startswith("--gpus=") for x in argv):
        # a launcher set the world and the command did not repeat it: the launcher is right
        print("bench: WORLD_SIZE=%s and no --gpus: running as %s ranks" % (env_world, env_world), file=sys.stderr)
        a.gpus = int(env_world)
    if int(env_world or 1) != a.gpus:
        print("bench: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d (or leave WORLD_SIZE unset and "
              "bench.py starts the ranks itself)" % (a.gpus, env_world, a.gpus), file=sys.stderr)
        return 2
    if a.runner and make_runner is None:
        import importlib
        mod, _, cls = a.runner.partition(":")
        make_runner = getattr(importlib.import_module(mod), cls)
    rank, world, dev = sharding.init_from_env(a.backend, use_cuda=make_runner is None, share_gpu=a.share_gpu)
    ranks_seen = sharding.sum_over_ranks(1, dev)   # an all-reduce of 1 over the group the job really has
    import vorbis_amd
    # rank 0 owns the setup blob; everyone else receives it over RCCL -- the job's only collective besides timing
    blob = sharding.broadcast_blob(vorbis_amd.default_setup_blob(a.setup) if rank == 0 else None, dev)
    R = (make_runner or (StreamRunner if a.workload == "c5" else GpuRunner))(a, blob, dev, rank, world)

    probe = ClockProbe(R, dev) if (make_runner is None and not a.no_clock_probe) else None
    elapsed, stage_ms = timed_run(a, R, dev, world, probe)
    clock = probe.result() if probe else None
    parity = None if a.no_parity_sample else parity_of(a, R, dev, world, a.parity_blocks)

    rc = 0
    units, unit_name, workload_text = R.units, R.unit_name, R.workload_text()
    roof = roofline_of(a, R, stage_ms, clock, elapsed / a.steps * 1e3) if rank == 0 else None
    default_run = a.workload == "c4" and a.blocks is None and make_runner is None
    neighbours = None
    if rank == 0 and world == 1 and a.workload == "c4" and not a.no_neighbours and make_runner is None:
        try:  # informational: the stages either side of the metric's path
            neighbours = R.neighbours()
        except Exception as e:
            neighbours = {"error": repr(e)}
    workloads = hostfed = None
    if default_run and not a.no_workloads:
        # the other BASELINE configs at their BASELINE sizes and the host-fed figures, after the headline (its tensors
        # released first); every rank takes part
        if hasattr(R, "an"):
            R.an.close()
        del R
        torch.cuda.empty_cache()
        workloads = extra_workloads(a, lambda name: sharding.broadcast_blob(vorbis_amd.default_setup_blob(name) if rank == 0 else None, dev),
                                    dev, rank, world)
        if any(d.get("parity_sample", {}).get("mismatches") for d in workloads.values()):
            rc = 3
        if not a.no_host_fed:
            hostfed = host_fed_all(a, dev, rank, world)
            if any((d.get("parity_sample") or {}).get("mismatches") for d in hostfed.values() if isinstance(d, dict)):
                rc = 3
    sharding.barrier()
    if rank == 0:
        value = world * units * a.steps / elapsed
        line = {
            "metric": "audio blocks/s (2048-sample MDCT+psy) @1/2/4/8 GPU; % HBM roofline",
            "value": value, "unit": unit_name, "n_gpus": world, "world": world, "rccl_ranks_seen": ranks_seen,
            "backend": (a.backend if world > 1 else None), "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "timed_region_s": elapsed, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload_text,
                "blocks_per_gpu": units, "setup": a.setup, "parallelism": "blocks sharded x%d, no data-path collective" % world,
            },
            "roofline": roof,
        }
        if clock:
            line["shader_clock"] = clock
        if a.share_gpu:
            line["share_gpu"] = True             # a rehearsal of the rank logic on ONE device: not a scaling measurement
            line["collectives_on"] = sharding.collectives_on()
        if a.runner:
            line["runner"] = a.runner            # not the GPU runner: a rehearsal, not a measurement
        if parity is not None:
            line["parity_sample"] = parity
            if parity.get("mismatches"):
                rc = 3
        if hostfed is not None:
            line["host_fed"] = hostfed
        if neighbours is not None:
            line["neighbours"] = neighbours
        if workloads is not None:
            line["workloads"] = workloads
        if not a.no_cpu_baseline and make_runner is None:
            # rank 0 only, after every GPU figure is in (the other ranks are past their last collective and leaving)
            try:
                line["cpu_baseline"] = cpu_baseline(a.setup, a.cpu_seconds)
            except Exception as e:  # a missing checker must not lose the GPU number
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(line))
        if rc:
            print("bench: a parity_sample found mismatching units -- the timed outputs differ from the oracle", file=sys.stderr)
    sharding.finish()
    return rc


if __name__ == "__main__":
    sys.exit(main())
