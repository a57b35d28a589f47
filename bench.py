#!/usr/bin/env python3
"""Headline benchmark: 40 ms packets/s, encode + decode round trip.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs ONE RANK PER GPU.  Started under a launcher (the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`) it reads RANK / LOCAL_RANK / WORLD_SIZE; started plainly (`python bench.py --gpus N`) it launches the N ranks itself
(solo_amd.dist.self_launch) -- and refuses with a non-zero exit code when fewer than N devices are visible.  WORLD_SIZE must equal
--gpus in every case: a line can never report another job than the one that ran.

One "step" = every stream of the batch encodes `--packets` consecutive 40 ms packets (AGR_Sate_Encoder_Encode semantics: QMF split,
SILK analysis, 3-track delayed-decision NSQ, range coding of both descriptions, high band) and decodes them again
(AGR_Sate_Decoder_Decode, both descriptions received, BWE resynthesis + QMF), through the C ABI of solo_amd/libsolo_mi355x.so.
Inputs are resident in HBM before the timed region; codec state stays in HBM between steps.

  N = 1   BASELINE.json configs[2]: 4096 synthetic streams, round trip                     (the configuration `metric` is quoted on)
          + two more timed legs in `extra`: configs[1] (the same 4096 streams, encode only) and configs[3] (8192 streams, decode
          only, every description lost with probability 0.3, first packet kept: single-description decoding + concealment)
  N > 1   BASELINE.json configs[4]: 8192 streams PER GPU (65 536 on 8), encode + decode per rank, streams sharded contiguously
          over the ranks with NO data-path collective; RCCL carries the barriers, the max-over-ranks time and ONE all_gather of the
          per-rank record {packets, seconds, payload_bytes, block hashes}.

Prints ONE JSON line (rank 0):
  value          whole-job packets/s = packets all ranks processed / max-over-ranks time of the K timed steps
  parity_checked the first step (freshly reset streams) of every rank is hashed per block of 4096 streams and compared with the
                 hashes the COMPILED REFERENCE produced for the same streams (tests/golden/bench_blocks.json, made by
                 tests/golden/make_bench_golden.py): true = every block of every rank equals the reference bit for bit
  roofline       SURVEY 8(d): the dominant kernel's ALGORITHMIC HBM bytes per launch -- what crosses the boundary: 1280 B PCM +
                 payload + 4 B of lengths per packet -- / its average launch duration, measured live with HIP events that the
                 library records around every launch on the launch stream (solo_batch_set_timing), against 8 TB/s.  The
                 kernel-to-kernel hand-over records and the per-stream state are implementation traffic: `traffic` (PMC: 2 x
                 FETCH_SIZE + WRITE_SIZE per launch, profiles/hbm_traffic.json), not `achieved`.
  valu_issue     the limiter that actually binds (serial fixed-point recursions): VALU wave-instructions per second of the whole
                 step (SQ_INSTS_VALU per packet from the rocprofv3 counter pass in profiles/, x measured packets/s) against the
                 chip's VALU issue peak (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)
  profile_matches_head   the counter-derived fields (traffic, valu_issue) come from profiles/*.json; true when those were collected
                 from the same kernel sources (solo_amd.kernel_source_hash) as the library that just ran
  cpu_baseline   the compiled reference (oracle/_ref, fixed-point tree; -O3 like the reference's own Release build, and the -O2
                 checker build beside it) on the host cores this process may use (affinity and cgroup quota respected)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

# Runtime knob (read by the HIP runtime when it initialises, i.e. before the first torch.cuda / library call): the number of
# hardware queues the process's HIP streams are spread over.  The encoder runs its three stages on three HIP streams, the decoder
# its two on two more; with the runtime's default (4 queues) consecutive encode / decode calls that are enqueued without a host
# synchronisation in between lose ~6 % (measured: 91.0 ms per step against 85.5 ms with 2 queues, 4096 streams x 50 packets; calls
# that are synchronised one by one take the same time either way, and 1 queue serialises the encoder's pipeline: 132 ms).
# Recorded in the JSON line (config.runtime_env); a value set by the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD, CLOCK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 2.0       # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
PCM_BYTES = 1280.0             # 640 int16 samples per 40 ms packet
LOSS_SEED, LOSS_P = 4242, 0.3  # configs[3]: Bernoulli description loss (also used by tests/golden/make_bench_golden.py)


def loss_mask(n_streams, n_packets, p=LOSS_P, seed=LOSS_SEED):
    """uint8 [N, P]: bit0 = MD1 received, bit1 = MD2 received; every description is lost with probability p, packet 0 is kept."""
    rng = np.random.default_rng(seed)
    m = (rng.random((n_streams, n_packets)) >= p).astype(np.uint8) | ((rng.random((n_streams, n_packets)) >= p).astype(np.uint8) << 1)
    m[:, 0] = 3
    return m


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the compiled reference on the host cores (fresh interpreter, no torch / HIP)
# ---------------------------------------------------------------------------------------------------------------------
def effective_cores():
    """CPUs this process can actually use: the scheduler affinity mask, capped by the cgroup CPU quota (cpu.max / cfs_quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info = {"os_cpu_count": os.cpu_count(), "affinity": n, "cgroup_quota_cpus": None}
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota:
        info["cgroup_quota_cpus"] = round(quota, 2)
        n = max(1, min(n, int(math.floor(quota + 1e-9))))
    info["effective"] = n
    return n, info


def _cpu_worker_main(args):
    """One worker process: round-trips whole streams of the synthetic workload through the compiled reference encoder + decoder
    (both descriptions received) for `passes` passes over its streams.  Returns packets, wall and CPU seconds."""
    kind, widx, n_streams, P, passes, t_start = args
    sys.path.insert(0, os.path.join(HERE, "oracle"))
    import refcodec as R
    from solo_amd.synth import synth_stream
    pcm = [synth_stream(widx * n_streams + i, P) for i in range(n_streams)]

    def run(x):
        e, d = R.RefEncoder(kind), R.RefDecoder(kind)
        for p in range(x.shape[0]):
            pl, n0, n1 = e.encode(x[p])
            d.decode(*R.map_loss(pl, n0, n1, False, False))
        e.close(); d.close()
        return x.shape[0]
    run(pcm[0][:8])                                           # library load, page-in
    while time.time() < t_start:                              # common start line
        time.sleep(0.001)
    w0, c0 = time.perf_counter(), time.process_time()
    done = 0
    for _ in range(passes):
        for x in pcm:
            done += run(x)
    return done, time.perf_counter() - w0, time.process_time() - c0


def _cpu_leg(kind, cores, target_seconds, P):
    import multiprocessing as mp
    S = 2
    t = _cpu_worker_main((kind, 0, 1, min(P, 100), 1, 0.0))    # calibrate one pass on one core, then size every worker's job
    per_packet = t[1] / t[0]
    passes = max(1, int(math.ceil(target_seconds / (per_packet * P * S))))
    with mp.get_context("fork").Pool(cores) as pool:
        t_start = time.time() + 1.5 + 0.02 * cores
        res = pool.map(_cpu_worker_main, [(kind, w, S, P, passes, t_start) for w in range(cores)], chunksize=1)
    packets = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    cpu_s = [r[2] for r in res]
    return {"value": round(packets / wall, 1), "per_core_packets_per_s": round(float(np.mean([r[0] / r[2] for r in res])), 1),
            "worker_cpu_seconds": [round(min(cpu_s), 2), round(float(np.mean(cpu_s)), 2), round(max(cpu_s), 2)], "wall_seconds": round(wall, 2),
            "packets": packets, "passes": passes, "streams_per_worker": S}


def cpu_worker(target_seconds, packets_per_stream):
    sys.path.insert(0, os.path.join(HERE, "oracle"))
    import refcodec as R
    if not R.have_ref("fix"):
        print(json.dumps(None))
        return
    cores, info = effective_cores()
    P = packets_per_stream
    have_o3 = R.have_ref("fix_O3")
    main = _cpu_leg("fix_O3" if have_o3 else "fix", cores, target_seconds, P)
    out = {"value": main["value"], "unit": "40ms packets/s (encode+decode)", "cores": cores, "kind": "reference",
           "build": "gcc -O3 (oracle/_ref/libsolo_ref_fix_O3.so: the optimisation level of the reference's own CMake Release build)" if have_o3
                    else "gcc -O2 (oracle/_ref/libsolo_ref_fix.so)",
           "per_core_packets_per_s": main["per_core_packets_per_s"], "worker_cpu_seconds": main["worker_cpu_seconds"],
           "wall_seconds": main["wall_seconds"], "host": info,
           "sample": "%d worker processes (= usable host CPUs: affinity %s, cgroup quota %s of %s hardware threads), each %d passes over %d "
                     "streams x %d packets of the same synthetic workload through the compiled fixed-point reference: %d packets in %.1f s wall; "
                     "worker_cpu_seconds = [min, mean, max]" % (cores, info["affinity"], info["cgroup_quota_cpus"], info["os_cpu_count"],
                                                                 main["passes"], main["streams_per_worker"], P, main["packets"], main["wall_seconds"])}
    if have_o3:                                               # the -O2 build is the parity checker of the test suite: timed beside it
        o2 = _cpu_leg("fix", cores, max(3.0, target_seconds / 2), P)
        out["O2_build"] = {"value": o2["value"], "per_core_packets_per_s": o2["per_core_packets_per_s"], "wall_seconds": o2["wall_seconds"],
                           "build": "gcc -O2 (oracle/_ref/libsolo_ref_fix.so, the checker)"}
    print(json.dumps(out))


def cpu_baseline(target_seconds, packets_per_stream):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-seconds", str(target_seconds),
                        "--cpu-packets-per-stream", str(packets_per_stream)], capture_output=True, text=True, timeout=900)
    for line in reversed(r.stdout.strip().splitlines()):
        try:
            return json.loads(line)
        except Exception:
            continue
    return None


def _profile_json(name):
    try:
        return json.load(open(os.path.join(HERE, "profiles", name)))
    except Exception:
        return None


def golden_blocks():
    """tests/golden/bench_blocks.json: reference-generated hashes per block of 4096 streams (data, committed; see the script beside it)"""
    try:
        g = json.load(open(os.path.join(HERE, "tests", "golden", "bench_blocks.json")))
        return g, {b["first_stream"]: b for b in g["blocks"]}
    except Exception:
        return None, {}


def check_blocks(first_stream, hashes, gold, gmap, P, rate, slot, pcm_key="pcm_md5"):
    """-> (checked: True / False / None, detail list).  None = nothing to compare with (no golden for these streams / this shape)."""
    if not hashes or gold is None or P != gold["packets"] or rate != gold["rate_bps"] or slot != gold["slot_bytes"] or first_stream % gold["block_streams"]:
        return None, []
    det, ok, seen = [], True, 0
    for i, h in enumerate(hashes):
        g = gmap.get(first_stream + i * gold["block_streams"])
        if g is None or g.get(pcm_key) is None:
            det.append({"first_stream": first_stream + i * gold["block_streams"], "checked": None})
            continue
        seen += 1
        good = (h["payload_md5"] == g["payload_md5"]) and (h["pcm_md5"] is None or h["pcm_md5"] == g[pcm_key])
        ok = ok and good
        det.append({"first_stream": g["first_stream"], "checked": bool(good), "payload_md5": h["payload_md5"], "pcm_md5": h["pcm_md5"]})
    return (ok if seen == len(hashes) else (False if not ok else None)), det


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: 4096 = configs[2] at N = 1, 8192 = configs[4] at N > 1)")
    ap.add_argument("--first-stream", type=int, default=-1, help="global index of this process's first stream (default: rank * streams)")
    ap.add_argument("--packets", type=int, default=50, help="40 ms packets per stream per step (SURVEY 8(d): P >= 50 = 2 s of audio)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] / configs[3] legs (N = 1 only)")
    ap.add_argument("--no-hash", action="store_true", help="skip the hashes of the first step (saves the device-to-host copies; parity_checked = null)")
    ap.add_argument("--no-overlap", action="store_true", help="strictly sequential schedule: every step's decode finishes before the next step's encode is issued "
                                                             "(default: the decode of step k is issued after the encode of step k+1, see config.schedule)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline: seconds of work per worker process")
    ap.add_argument("--cpu-packets-per-stream", type=int, default=400)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def timed_loop(fn, steps, warmup, barrier):
    """W untimed calls, then exactly `steps` calls between two barrier + synchronize brackets -> seconds"""
    for _ in range(warmup):
        fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    return time.perf_counter() - t0


def main():
    args = parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_seconds, args.cpu_packets_per_stream)
        return
    from solo_amd import dist as sdist
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (never silently time one GPU)
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible on this node" % (args.gpus, have))
        sdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)          # execs; does not return

    world, rank, local_rank = sdist.env_world()
    sdist.check_world(args.gpus, world)
    # the rank's input batch, generated by forked worker processes BEFORE the HIP runtime / RCCL come up in this process
    N = args.streams if args.streams > 0 else (4096 if world == 1 else 8192)
    P, RATE, SLOT = args.packets, 13600, 512
    from solo_amd.synth import synth_batch
    first = args.first_stream if args.first_stream >= 0 else sdist.stream_range(rank, N)[0]
    ncpu, _ = effective_cores()
    workers = max(1, min(16, ncpu // max(1, min(world, 8))))
    # (outside the timed region.  The generator costs ~19 ms per stream of 50 packets on one host core: a rank of the 8-GPU job -- 8192
    # streams, 16 usable CPUs / 8 ranks = 2 workers -- spends ~80 s here, the one-GPU run ~5 s; reported as config.input_generation_s)
    t_gen = time.perf_counter()
    pcm_host = synth_batch(first, N, P, workers=workers)
    t_gen = time.perf_counter() - t_gen
    import torch
    import solo_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the codec has no CPU path")
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d HIP device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    dist = None
    # SOLO_FORCE_DIST=1 (tests/test_gpu_dist_nccl.py): run the N > 1 code path -- process group over RCCL, barriers, all_reduce(MAX),
    # all_gather_object of the per-rank records -- also when the launcher started ONE rank, so that the collectives of the
    # multi-GPU path execute on hardware on a one-GPU box; the record must equal the plain single-process run's
    use_dist = world > 1 or (launched and os.environ.get("SOLO_FORCE_DIST") == "1")
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist = sdist.init("nccl", torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    pcm = torch.from_numpy(pcm_host).to(dev)
    batch = solo_amd.SoloBatch(N, rate=RATE, encoder=True, decoder=True, slot_bytes=SLOT)
    bits = torch.zeros((N, P, SLOT), dtype=torch.uint8, device=dev)
    nb = torch.zeros((N, P, 2), dtype=torch.int16, device=dev)
    st_e = torch.zeros((N,), dtype=torch.int32, device=dev)
    st_d = torch.zeros((N,), dtype=torch.int32, device=dev)
    out = torch.zeros((N, P, 640), dtype=torch.int16, device=dev)
    payload_acc = torch.zeros((1,), dtype=torch.int64, device=dev)      # payload bytes actually produced (summed on the device per step)

    # (per-kernel timing brackets -- HIP events with timestamps around every launch -- are switched on only for the few sampling
    # steps AFTER the timed region: they cost several per cent of throughput)
    kms = {"analysis": [], "quantiser": [], "coding": [], "decode": []}

    # Default schedule (off with --no-overlap): consecutive encode calls are pipelined (solo_batch_set_async_join: the first analysis
    # chunk of step k + 1 starts while the last quantiser / coding chunks of step k still run) and the decode of step k is issued
    # after the encode of step k + 1; the bitstream buffers are double-buffered.  Default: encode then decode, one stream.
    overlap = not args.no_overlap
    bits2, nb2 = [bits, torch.zeros_like(bits) if overlap else bits], [nb, torch.zeros_like(nb) if overlap else nb]
    step_no = [0]
    if overlap:
        batch.set_async_join(True)

    def step():
        if not overlap:
            batch.encode(pcm, bits, nb, st_e)
            batch.decode(bits, nb, None, out, st_d)
            payload_acc.add_(nb[:, :, 0].sum(dtype=torch.int64))
            return
        j = step_no[0] & 1
        batch.encode(pcm, bits2[j], nb2[j], st_e)
        if step_no[0] > 0:                                # the previous step's packets: wait for THAT encode call, then decode
            batch.wait_encode(1)
            batch.decode(bits2[1 - j], nb2[1 - j], None, out, st_d)
            payload_acc.add_(nb2[1 - j][:, :, 0].sum(dtype=torch.int64))
        step_no[0] += 1

    def drain():
        if overlap and step_no[0] > 0:
            j = (step_no[0] - 1) & 1
            batch.wait_encode(0)
            batch.decode(bits2[j], nb2[j], None, out, st_d)
            payload_acc.add_(nb2[j][:, :, 0].sum(dtype=torch.int64))
            step_no[0] = 0

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The first step after a reset is the one the compiled reference reproduces bit for bit from the same input: hash its outputs
    # per block of 4096 streams (outside the timed region).  Later steps continue the streams' state with the same input.
    step()
    drain()
    torch.cuda.synchronize()
    mean_payload = float(nb[:, :, 0].float().mean().item())
    first_step = (None, None, None) if args.no_hash else (nb.cpu().numpy(), bits.cpu().numpy(), out.cpu().numpy())
    # Steps 2 and 3 run under the TIMED schedule (pipelined: the encode of step 3 is issued before the decode of step 2) and continue the
    # streams' state; the third step's payloads and PCM are hashed as well (the reference encoded the same 50 packets three times in a row)
    step()
    step()
    jl = (step_no[0] - 1) & 1 if overlap else 0
    drain()
    torch.cuda.synchronize()
    third_step = (None, None, None) if args.no_hash else (nb2[jl].cpu().numpy(), bits2[jl].cpu().numpy(), out.cpu().numpy())
    for _ in range(max(0, args.warmup - 3)):
        step()
    drain()
    barrier()
    payload_acc.zero_()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    drain()                                               # (the last step's decode: every packet is encoded AND decoded in the timed region)
    barrier()
    dt_local = time.perf_counter() - t0
    payload_bytes_timed = int(payload_acc.item())
    # effective shader clock under vector load, right after the timed steps (the boxes of a pool differ by ~10 %: lines are comparable
    # only with it); a ~1 ms probe kernel of the library, outside the timed region
    sclk_mhz = solo_amd.shader_clock_mhz()
    # per-kernel durations: a few extra steps outside the timed region (reading the events synchronises the stream)
    if overlap:
        batch.set_async_join(False)
        overlap = False
    batch.set_timing(True)
    for _ in range(min(3, args.steps)):
        step()
        for name, v in batch.last_kernel_ms().items():
            kms[name].append(v)
    batch.set_timing(False)
    enc_chunks = max(1, batch.last_encode_chunks())
    dt = sdist.max_over_ranks(dt_local, dist, dev) if use_dist else dt_local
    assert int(st_e.abs().max()) == 0 and int(st_d.abs().max()) == 0, "codec status != 0"
    kavg = {n: float(np.mean(v)) for n, v in kms.items()}
    packets_step = N * P
    value = world * packets_step * args.steps / dt

    gold, gmap = golden_blocks()
    hashes = None if args.no_hash else sdist.block_hashes(*first_step)
    checked, detail = check_blocks(first, hashes, gold, gmap, P, RATE, SLOT)
    # the state-continued third step under the pipelined schedule (reference hashes exist for block 0)
    checked3 = None
    if not args.no_hash and gold is not None and first == 0 and gmap.get(0, {}).get("step3_payload_md5"):
        b3 = third_step[1][:4096]
        b3 = np.where(np.arange(b3.shape[2])[None, None, :] < third_step[0][:4096, :, 0:1].astype(np.int64), b3, 0).astype(np.uint8)   # (the slots still hold the tails of step 1's payloads)
        h3 = sdist.block_hashes(third_step[0][:4096], b3, third_step[2][:4096])[0]
        checked3 = bool(h3["payload_md5"] == gmap[0]["step3_payload_md5"] and h3["pcm_md5"] == gmap[0]["step3_pcm_md5"])
        if checked3 is False:
            checked = False

    # ---- extra legs (N = 1): BASELINE configs[1] and configs[3], each its own timed loop ------------------------------------
    extra = {}
    if world == 1 and not args.no_extra and not args.no_overlap:
        # the same round trip with the strictly sequential schedule (decode of a step done before the next encode is issued)
        def seq_step():
            batch.encode(pcm, bits, nb, st_e)
            batch.decode(bits, nb, None, out, st_d)
        ns = max(2, min(5, args.steps))
        dt_s = timed_loop(seq_step, ns, 1, barrier)
        extra["sequential_schedule"] = {"workload": "the headline round trip, encode then decode on one stream (--no-overlap)", "steps": ns, "warmup": 1,
                                        "value": round(packets_step * ns / dt_s, 1), "unit": "40ms packets/s (encode+decode)", "ms_per_step": round(dt_s / ns * 1e3, 3)}
    if world == 1 and not args.no_extra:
        # configs[1]: the same 4096 streams, encode only
        dt_e = timed_loop(lambda: batch.encode(pcm, bits, nb, st_e), args.steps, max(1, args.warmup), barrier)
        assert int(st_e.abs().max()) == 0
        extra["configs[1]"] = {"workload": "BASELINE configs[1]: %d synthetic 16 kHz WB streams, encode only, 13.6 kbps, %d packets/stream/step" % (N, P),
                               "value": round(packets_step * args.steps / dt_e, 1), "unit": "40ms packets/s (encode)", "steps": args.steps,
                               "warmup": max(1, args.warmup), "ms_per_step": round(dt_e / args.steps * 1e3, 3),
                               "parity_checked": checked, "parity_note": "the payload hashes of the main leg's first step ARE this configuration's output"}
        # configs[3]: 8192 streams, decode only, every description lost with probability 0.3 (seeded Bernoulli draws, packet 0 kept)
        N8 = 8192
        more = synth_batch(first + N, N8 - N, P, workers=workers) if N8 > N else None
        pcm8 = torch.cat([pcm, torch.from_numpy(more).to(dev)]) if more is not None else pcm[:N8]
        b8 = solo_amd.SoloBatch(N8, rate=RATE, encoder=True, decoder=True, slot_bytes=SLOT)
        bits8, nb8, st8 = b8.encode(pcm8)
        recv = torch.from_numpy(loss_mask(N8, P)).to(dev)
        out8 = torch.zeros((N8, P, 640), dtype=torch.int16, device=dev)
        st8d = torch.zeros((N8,), dtype=torch.int32, device=dev)
        b8.decode(bits8, nb8, recv, out8, st8d)           # first decode of freshly reset decoders: the one the reference hashes describe
        torch.cuda.synchronize()
        assert int(st8.abs().max()) == 0 and int(st8d.abs().max()) == 0
        chk8, det8 = (None, [])
        if not args.no_hash:
            h8 = sdist.block_hashes(nb8.cpu().numpy(), bits8.cpu().numpy(), out8.cpu().numpy())
            chk8, det8 = check_blocks(first, h8, gold, gmap, P, RATE, SLOT, pcm_key="pcm_loss30_md5")
        dt_d = timed_loop(lambda: b8.decode(bits8, nb8, recv, out8, st8d), args.steps, max(1, args.warmup), barrier)
        assert int(st8d.abs().max()) == 0
        m = recv.cpu().numpy()
        extra["configs[3]"] = {"workload": "BASELINE configs[3]: %d streams decode only, descriptions lost independently with probability %.1f "
                                           "(numpy default_rng(%d), packet 0 kept): single-description decoding + concealment, %d packets/stream/step" % (N8, LOSS_P, LOSS_SEED, P),
                               "value": round(N8 * P * args.steps / dt_d, 1), "unit": "40ms packets/s (decode)", "steps": args.steps,
                               "warmup": max(1, args.warmup), "ms_per_step": round(dt_d / args.steps * 1e3, 3),
                               "packets_both_received": int((m == 3).sum()), "packets_one_description": int(((m == 1) | (m == 2)).sum()),
                               "packets_lost": int((m == 0).sum()), "parity_checked": chk8, "parity_blocks": det8,
                               "parity_note": "encoder payloads of the 8192 streams and the PCM of the first decode under this loss pattern, per block of "
                                              "4096 streams, against the compiled reference (bench_blocks.json: payload_md5 / pcm_loss30_md5)"}
        # configs[4], one GPU's share: the SAME per-GPU load as a rank of the 8-GPU job (8192 streams, blocks 0-1 of the 16), encode + decode,
        # freshly reset streams, first step hashed -- so that an N = 1 line and an N > 1 line can be compared per GPU
        b8.reset()
        torch.cuda.synchronize()

        def rt8():
            b8.encode(pcm8, bits8, nb8, st8)
            b8.decode(bits8, nb8, None, out8, st8d)
        rt8()
        torch.cuda.synchronize()
        chk4, det4 = (None, [])
        if not args.no_hash:
            h4 = sdist.block_hashes(nb8.cpu().numpy(), bits8.cpu().numpy(), out8.cpu().numpy())
            chk4, det4 = check_blocks(first, h4, gold, gmap, P, RATE, SLOT)
        n4 = max(2, min(5, args.steps))
        dt_4 = timed_loop(rt8, n4, 1, barrier)
        assert int(st8.abs().max()) == 0 and int(st8d.abs().max()) == 0
        extra["configs[4]_share"] = {"workload": "BASELINE configs[4], ONE GPU's share: %d streams (the load of one rank of the 65536-stream / 8-GPU job), encode then decode "
                                                 "on one stream, %d packets/stream/step" % (N8, P), "value": round(N8 * P * n4 / dt_4, 1), "unit": "40ms packets/s (encode+decode)",
                                     "steps": n4, "warmup": 1, "ms_per_step": round(dt_4 / n4 * 1e3, 3), "parity_checked": chk4, "parity_blocks": det4,
                                     "parity_note": "first step after a reset: payloads and decoded PCM of blocks 0-1 of the 16 reference-hashed blocks (bench_blocks.json); blocks 2-15 "
                                                    "on one GPU: tests/test_gpu_fullsize.py::test_config4_all_blocks_on_one_gpu"}
        del b8, bits8, nb8, out8, pcm8, recv

    if world == 1 and not args.no_extra and P % 2 == 0:
        # the 32 kHz mode (`samplerate = 32000`: 1280-sample packets, SILK wide band inside, 24 kbps): the same input
        # samples taken as 32 kHz audio, P / 2 packets per stream per step, sequential encode then decode
        bw = solo_amd.SoloBatch(N, rate=24000, encoder=True, decoder=True, slot_bytes=SLOT, samplerate=32000)
        Pw = P // 2
        xw = pcm.view(N, Pw, 1280)
        bits_w = torch.zeros((N, Pw, SLOT), dtype=torch.uint8, device=dev); nb_w = torch.zeros((N, Pw, 2), dtype=torch.int16, device=dev)
        out_w = torch.zeros((N, Pw, 1280), dtype=torch.int16, device=dev)
        st_we = torch.zeros((N,), dtype=torch.int32, device=dev); st_wd = torch.zeros((N,), dtype=torch.int32, device=dev)

        def wb_step():
            bw.encode(xw, bits_w, nb_w, st_we)
            bw.decode(bits_w, nb_w, None, out_w, st_wd)
        wb_step()
        torch.cuda.synchronize()
        chk_w = None
        if not args.no_hash and gold is not None and first == 0 and N >= 4096 and gmap.get(0, {}).get("wb_payload_md5"):
            hw = sdist.block_hashes(nb_w[:4096].cpu().numpy(), bits_w[:4096].cpu().numpy(), out_w[:4096].cpu().numpy())[0]
            chk_w = bool(hw["payload_md5"] == gmap[0]["wb_payload_md5"] and hw["pcm_md5"] == gmap[0]["wb_pcm_md5"])
        nw = max(2, min(5, args.steps))
        dt_w = timed_loop(wb_step, nw, 1, barrier)
        assert int(st_we.abs().max()) == 0 and int(st_wd.abs().max()) == 0
        extra["samplerate_32000"] = {"workload": "%d streams in the 32 kHz mode (1280-sample 40 ms packets, SILK wide band + 8-16 kHz high band, 24 kbps), encode then decode, "
                                                 "%d packets/stream/step" % (N, Pw), "value": round(N * Pw * nw / dt_w, 1), "unit": "40ms packets/s (encode+decode, 32 kHz)",
                                     "steps": nw, "warmup": 1, "ms_per_step": round(dt_w / nw * 1e3, 3), "mean_payload_bytes": round(float(nb_w[:, :, 0].float().mean().item()), 2),
                                     "parity_checked": chk_w, "parity_note": "first call after creation: payloads and decoded PCM of block 0 against the compiled reference run with samplerate = 32000 "
                                                                           "(bench_blocks.json: wb_payload_md5 / wb_pcm_md5)"}
        del bw, bits_w, nb_w, out_w

    # ONE all_gather of the per-rank record (RCCL); no other collective besides the barriers and the max time
    record = sdist.result_record(rank, first, N, packets_step * args.steps, dt_local, *first_step)
    record["payload_bytes"] = payload_bytes_timed            # produced in the timed steps (device-side sum of nBytesOut[0])
    record["parity_checked"] = checked
    record["blocks"] = detail
    record["device"] = torch.cuda.get_device_name(dev)
    record["shader_clock_mhz_under_vector_load"] = None if sclk_mhz is None else round(sclk_mhz, 1)
    records = sdist.gather_records(record, dist) if use_dist else [record]

    if rank == 0:
        all_checked = [r["parity_checked"] for r in records]
        parity = (all(c is True for c in all_checked) if all(c is not None for c in all_checked) else (False if any(c is False for c in all_checked) else None))
        # ALGORITHMIC bytes per packet: what crosses the boundary, with the run's mean payload
        enc_alg = PCM_BYTES + mean_payload + 4.0
        dec_alg = mean_payload + 4.0 + PCM_BYTES
        # the three encoder kernels are stages of ONE pass over the boundary data of a packet (PCM in, payload + lengths out): each
        # is priced with the encode-only boundary bytes of the packets it handles per launch; the hand-over records between them are
        # implementation traffic
        alg = {"analysis": enc_alg, "quantiser": enc_alg, "coding": enc_alg, "decode": dec_alg}
        split_dec = os.environ.get("SOLO_DEC_SPLIT", "1") != "0"
        stage_kernels = {"analysis": ["solo_enc_analysis_kernel"], "quantiser": ["solo_nsq_kernel"],
                         "coding": ["solo_enc_coding_kernel", "solo_enc_rc_kernel"],
                         "decode": ["solo_dec_synth_kernel", "solo_dec_extract_kernel"] if split_dec else ["solo_decode_kernel"]}
        kname = {n: v[0] for n, v in stage_kernels.items()}
        traffic = _profile_json("hbm_traffic.json") or {}
        insts = _profile_json("wave_instructions.json") or {}
        src_hash = solo_amd.kernel_source_hash()
        prof_hash = {"hbm_traffic.json": traffic.get("kernel_source_sha16"), "wave_instructions.json": insts.get("kernel_source_sha16")}
        prof_ok = all(v == src_hash for v in prof_hash.values())
        dec_chunk = int(os.environ.get("SOLO_DEC_CHUNK", "64"))       # (solo_api.hip: one chunk up to 64 packets)
        dec_first = int(os.environ.get("SOLO_DEC_FIRST_CHUNK", "64"))
        if split_dec and dec_chunk > 0:
            cp = min(P, dec_chunk)
            c0 = dec_first if (P > 2 * dec_first and cp > dec_first) else cp
            dec_chunks = 1 + (P - c0 + cp - 1) // cp
        else:
            dec_chunks = 1
        nl = {n: (enc_chunks if n != "decode" else dec_chunks) for n in kavg}

        def tr_launch(n):
            v = [traffic.get(k + "_bytes_per_packet") for k in stage_kernels[n]]
            return None if any(x is None for x in v) else int(sum(v) * packets_step / nl[n])
        kernels = {kname[n]: {"launches_per_step": nl[n], "avg_launch_ms": round(kavg[n] / nl[n], 4),
                              "kernels_of_the_stage": stage_kernels[n],
                              "packets_per_launch": packets_step // nl[n],
                              "algorithmic_bytes_per_launch": int(alg[n] * packets_step / nl[n]),
                              "achieved_GBps": round(alg[n] * packets_step / (kavg[n] * 1e-3) / 1e9, 4),
                              "frac_of_hbm_peak": round(alg[n] * packets_step / (kavg[n] * 1e-3) / 1e9 / HBM_PEAK_GBS, 7),
                              "traffic_bytes_per_launch": tr_launch(n),
                              "valu_lane_utilisation": {k: (insts.get("valu_lane_utilisation") or {}).get(k) for k in stage_kernels[n]}} for n in kavg}
        dom = max(kavg, key=kavg.get)
        alg_launch = alg[dom] * packets_step / nl[dom]
        achieved = alg_launch / (kavg[dom] / nl[dom] * 1e-3) / 1e9
        step_alg = (enc_alg + dec_alg) * packets_step
        step_gbs = step_alg / (dt / args.steps) / 1e9
        res = {
            "metric": "40 ms frames/sec (encode+decode) per GPU; concurrent real-time WB streams @1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "40ms packets/s (encode+decode)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32 fixed point (int16 PCM, Q-format arithmetic, bit-exact)", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: %d synthetic 16 kHz WB streams, full encode -> two-description bitstream -> decode "
                                    "round trip with BWE resynthesis, 13.6 kbps, %d packets/stream/step" % (N, P)) if world == 1 else
                                   ("BASELINE configs[4]: %d synthetic 16 kHz WB streams sharded evenly across %d x MI355X (%d per GPU), encode + "
                                    "decode per rank, RCCL gather only, 13.6 kbps, %d packets/stream/step" % (N * world, world, N, P)),
                       "streams_per_gpu": N, "packets_per_stream_per_step": P, "mean_payload_bytes": round(mean_payload, 2),
                       "input_generation_s": round(t_gen, 1), "input_workers": workers,
                       "launch": "torch.distributed.run, one rank per GPU" + (" (started by bench.py itself)" if os.environ.get("SOLO_SELF_LAUNCHED") else "") if world > 1 else "single process",
                       "runtime_env": {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "SOLO_DEC_SPLIT", "SOLO_DEC_CHUNK", "SOLO_DEC_FIRST_CHUNK", "SOLO_ENC_CHUNK") if k in os.environ},
                       "schedule": ("consecutive steps pipelined: encode of step k+1 issued before the decode of step k "
                                    "(solo_batch_set_async_join, double-buffered bitstreams); every packet is encoded AND decoded inside the timed region" if not args.no_overlap
                                    else "encode then decode on one stream")},
            "parity_checked": parity,
            "parity": {"what": "first step (freshly reset streams) hashed per block of 4096 streams: md5(nBytes || payload slots) and md5(decoded PCM), "
                               "compared with the compiled reference's hashes of the same streams (tests/golden/bench_blocks.json); block 0 also after "
                               "steps 2 and 3 run under the timed (pipelined) schedule: the third, state-continued step against the reference's third pass",
                       "third_step_pipelined_checked": checked3,
                       "ranks": [{"rank": r["rank"], "first_stream": r["first_stream"], "checked": r["parity_checked"], "blocks": r["blocks"]} for r in records]},
            "shader_clock_mhz_under_vector_load": None if sclk_mhz is None else round(sclk_mhz, 1),
            "clock_note": "effective shader clock of rank 0 right after the timed steps while every SIMD runs vector instructions (solo_debug_clock: shader-clock "
                          "counter ticks per 100 MHz tick); boxes of a pool differ, compare lines at equal clock",
            "realtime_streams": round(value / 25.0, 1),
            "per_gpu_packets_per_s": [r["packets_per_s"] for r in records],
            "whole_node_packets_per_s": round(value, 1),
            "rccl_ranks": world if use_dist else 0,
            "ranks": [{k: v for k, v in r.items() if k != "blocks"} for r in records],
            "roofline": {"kernel": kname[dom], "bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": tr_launch(dom),
                         "traffic_is": "L2 memory-side (fabric) bytes per launch, rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE: Infinity-Cache hits are included (no counter "
                                       "behind the Infinity Cache is exposed), so an UPPER bound of the HBM bytes; the 32 MB quantiser ring lives in the 256 MiB Infinity Cache",
                         "avg_launch_ms": round(kavg[dom] / nl[dom], 4), "launches_per_step": nl[dom],
                         "packets_per_launch": packets_step // nl[dom],
                         "algorithmic_bytes_per_packet": round(alg[dom], 2),
                         "algorithmic_bytes_per_launch": int(alg_launch),
                         "formula": "achieved = packets_per_launch x (1280 + mean_payload + 4) B / avg_launch_ms; frac = achieved / 8 TB/s "
                                    "(SURVEY 8(d); hand-over records and stream state count as traffic, not as algorithmic bytes).  Every encoder "
                                    "stage in `kernels` is priced with the full encode boundary bytes of its packets (the stages are one pass over "
                                    "a packet): their achieved_GBps are per stage, not additive",
                         "whole_step": {"algorithmic_bytes_per_packet": round(enc_alg + dec_alg, 2), "achieved": round(step_gbs, 4),
                                        "frac": round(step_gbs / HBM_PEAK_GBS, 7)},
                         "note": "serial fixed-point recursions: instruction-issue / latency bound, not HBM bound -- see valu_issue"},
            "kernels": kernels,
            "launches_per_step": {k: nl[n] for n, ks in stage_kernels.items() for k in ks},
            "kernel_source_sha16": src_hash,
            "profile_matches_head": prof_ok,
            "profile_source": {"files": prof_hash, "git_head_of_profile": traffic.get("git_head"), "summary": insts.get("source")},
        }
        if extra:
            res["extra"] = extra
        peak_issue = N_SIMD * CLOCK_HZ / CYCLES_PER_VALU
        if insts.get("valu_per_packet_round_trip"):
            v = insts["valu_per_packet_round_trip"] * (value / world)
            # share of the time a SIMD's vector unit is executing: VALU wave-instructions of this run per SIMD and second (SQ_INSTS_VALU of the
            # committed counter pass x this run's packet rate) x the measured cost of an instruction of this path's class mix (a third of them
            # full rate at 1.15 ns, two thirds half rate at 1.85 ns per SIMD: profiles/r05_issue_model.txt, tools/debug/nsq_row_mix.py)
            busy = v / N_SIMD * (1.15e-9 / 3.0 + 2.0 * 1.85e-9 / 3.0)
            res["valu_issue"] = {"bound": "valu_issue", "achieved": round(v / 1e9, 2), "peak": round(peak_issue / 1e9, 1), "valu_busy_frac": round(busy, 3),
                                 "unit": "G wave-instructions/s (VALU)", "frac": round(v / peak_issue, 4),
                                 "valu_wave_instructions_per_packet": insts.get("valu_per_packet"),
                                 "all_wave_instructions_per_packet": insts.get("all_per_packet"),
                                 "valu_lane_utilisation": insts.get("valu_lane_utilisation"),
                                 "lane_utilisation_is": "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes switched on (EXEC) while a vector instruction executes",
                                 "source": insts.get("source"), "from_this_build": prof_ok,
                                 "note": "SQ_INSTS_VALU per packet of each kernel (rocprofv3 --pmc pass) x packets/s of this run, per GPU; "
                                         "peak = 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (nominal).  Measured on gfx950 "
                                         "(tools/debug/mb_mix.hip, profiles/r05_issue_model.txt; round 4: mb_valu.hip, r04_valu_microbench.txt): a SIMD retires one "
                                         "add / sub / logic / arithmetic-shift instruction with register operands per 1.1 - 1.2 ns and one multiply / left shift / "
                                         "min / max / select / DPP / 3-operand / SGPR-operand instruction per 1.8 - 1.9 ns (a v_mul_hi + v_add pair 3.0, v_mad_i64_i32 2.1); "
                                         "scalar and LDS instructions of OTHER waves overlap with them (a type mix costs ~ the maximum, not the sum); one wave issues "
                                         "a dependent instruction every 3.6 - 4.0 ns, two waves interleave perfectly, four saturate the unit.  The encoder pipeline "
                                         "holds 5 waves per SIMD (registers and LDS full) and keeps the vector unit ~65 % busy: time follows the instruction count"}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_seconds, args.cpu_packets_per_stream)
            if cb and cb.get("per_core_packets_per_s"):
                cb["gpu_equals_reference_cores"] = round(value / cb["per_core_packets_per_s"], 1)
            res["cpu_baseline"] = cb
        print(json.dumps(compact_line(res), separators=(",", ":")), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if checked is False:
        raise SystemExit("bench.py: the first step does NOT equal the compiled reference's hashes (rank %d, first stream %d)" % (rank, first))


# ---- the line as printed: under ~6 KB, the fields a reader looks for first at its END (a driver keeps the tail of stdout) -------------
# What the numbers mean (formulae, where the counters come from, how the CPU sample was taken) is DESIGN.md section 6, not the line.
_DROP = {"note", "parity_note", "clock_note", "formula", "what", "profile_source", "lane_utilisation_is", "kernels_of_the_stage",
         "worker_cpu_seconds", "host", "whole_node_packets_per_s", "per_gpu_packets_per_s", "source", "from_this_build", "algorithmic_bytes_per_launch",
         "packets_per_launch", "achieved_GBps", "traffic_is"}
_TAIL = ("kernel_source_sha16", "profile_matches_head", "third_step_pipelined_checked", "shader_clock_mhz_under_vector_load", "valu_busy_frac", "parity_checked")


def _squeeze(o, depth=0):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in _DROP:
                continue
            if k.endswith("_md5") and isinstance(v, str):
                v = v[:12]
            elif k in ("workload", "sample", "build", "schedule", "launch", "traffic_is") and isinstance(v, str) and len(v) > 100:
                v = v[:97] + "..."
            out[k] = _squeeze(v, depth + 1)
        return out
    if isinstance(o, list):
        return [_squeeze(v, depth + 1) for v in o]
    return o


def compact_line(res):
    r = _squeeze(res)
    if isinstance(r.get("parity"), dict):
        r["third_step_pipelined_checked"] = r["parity"].get("third_step_pipelined_checked")
        ranks = r["parity"].get("ranks") or []
        r["parity"] = {"third_step_pipelined_checked": r["third_step_pipelined_checked"], "blocks_checked": sum(len(x.get("blocks") or []) for x in ranks),
                       "first_block": (ranks[0]["blocks"][0] if ranks and ranks[0].get("blocks") else None)}
    for kv in (r.get("kernels") or {}).values():      # (the lane utilisation of every kernel is in valu_issue)
        if isinstance(kv, dict):
            kv.pop("valu_lane_utilisation", None)
    vi = r.get("valu_issue")
    if isinstance(vi, dict) and vi.get("valu_busy_frac") is not None:
        r["valu_busy_frac"] = vi["valu_busy_frac"]
    for k in _TAIL:                                # re-insert at the end, in this order
        if k in r:
            r[k] = r.pop(k)
    return r


if __name__ == "__main__":
    main()
