#!/usr/bin/env python3
"""Headline benchmark: 40 ms packets/s, encode + decode round trip (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = every stream of the batch encodes `--packets` consecutive 40 ms packets (AGR_Sate_Encoder_Encode
semantics: QMF split, SILK analysis, 3-track delayed-decision NSQ, range coding of both descriptions, high band)
and decodes them again (AGR_Sate_Decoder_Decode, both descriptions received, BWE resynthesis + QMF), through the
C ABI of solo_amd/libsolo_mi355x.so.  Inputs are resident in HBM before the timed region; codec state stays in
HBM between steps.  Streams shard over ranks with no data-path collective ("weak" scaling: 4096 streams per GPU);
RCCL is used only for the barrier and the max-over-ranks time.

Prints ONE JSON line (rank 0).  The encoder is a three-kernel pipeline (analysis -> quantiser -> coding) that the library
runs over chunks of the step's packets on three internal streams, so its kernels overlap in time; `roofline` is for the
kernel with the largest summed duration per step: algorithmic HBM bytes of that kernel per launch /
its average launch duration, measured with HIP events recorded by the library around every launch on its stream
(solo_batch_set_timing).  `kernels` lists all four kernels the same way.  `cpu_baseline` times the compiled reference (oracle/_ref, fixed-point tree)
on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def _cpu_gen(args):
    from solo_amd.synth import synth_stream
    return synth_stream(args[0], args[1])


def _cpu_run(pcm):
    sys.path.insert(0, os.path.join(HERE, "oracle"))
    import refcodec as R
    e, d = R.RefEncoder("fix"), R.RefDecoder("fix")
    for p in range(pcm.shape[0]):
        pl, n0, n1 = e.encode(pcm[p])
        d.decode(*R.map_loss(pl, n0, n1, False, False))
    e.close()
    return pcm.shape[0]


def cpu_worker(n_packets_total, packets_per_stream):
    """Runs in a fresh interpreter (no torch / HIP): one process per host CPU, each looping whole streams through the
    compiled reference encoder + decoder (both descriptions received)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(HERE, "oracle"))
    import refcodec as R
    if not R.have_ref("fix"):
        print(json.dumps(None))
        return
    cores = os.cpu_count() or 1
    P = packets_per_stream
    n_streams = max(cores, n_packets_total // P)
    n_streams = (n_streams + cores - 1) // cores * cores
    with mp.get_context("fork").Pool(cores) as pool:
        pcm = pool.map(_cpu_gen, [(i, P) for i in range(n_streams)])
        pool.map(_cpu_run, [x[:2] for x in pcm[:cores]])      # warm every worker (library load, page-in)
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_run, pcm, chunksize=1))
        dt = time.perf_counter() - t0
    print(json.dumps({"value": round(done / dt, 1), "unit": "40ms packets/s (encode+decode)", "cores": cores, "kind": "reference",
                      "sample": "%d streams x %d packets of the same synthetic workload through the compiled fixed-point "
                                "reference (oracle/_ref/libsolo_ref_fix.so, gcc -O2), %d processes, %.1f s wall"
                                % (n_streams, P, cores, dt)}))


def cpu_baseline(n_packets_total, packets_per_stream):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-packets", str(n_packets_total),
                        "--cpu-packets-per-stream", str(packets_per_stream)], capture_output=True, text=True, timeout=600)
    for line in reversed(r.stdout.strip().splitlines()):
        try:
            return json.loads(line)
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--packets", type=int, default=50, help="40 ms packets per stream per step (SURVEY 8(d): P >= 50 = 2 s of audio)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="decode of step k on a second stream beside the encode of step k+1 (+1 %)")
    ap.add_argument("--cpu-packets", type=int, default=0, help="0: 400 packets per host CPU")
    ap.add_argument("--cpu-packets-per-stream", type=int, default=400)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_packets <= 0:
        args.cpu_packets = 400 * (os.cpu_count() or 1)
    if args.cpu_worker:
        cpu_worker(args.cpu_packets, args.cpu_packets_per_stream)
        return

    import torch
    import solo_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the codec has no CPU path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from solo_amd import dist as sdist
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist = sdist.init("nccl", torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    dev = torch.device("cuda", torch.cuda.current_device())

    N, P = args.streams, args.packets
    from solo_amd.synth import synth_batch
    first = sdist.stream_range(rank, N)[0]
    pcm = torch.from_numpy(synth_batch(first, N, P, workers=min(16, os.cpu_count() or 1))).to(dev)
    batch = solo_amd.SoloBatch(N, rate=13600, encoder=True, decoder=True, slot_bytes=512)
    bits = torch.zeros((N, P, 512), dtype=torch.uint8, device=dev)
    nb = torch.zeros((N, P, 2), dtype=torch.int16, device=dev)
    st_e = torch.zeros((N,), dtype=torch.int32, device=dev)
    st_d = torch.zeros((N,), dtype=torch.int32, device=dev)
    out = torch.zeros((N, P, 640), dtype=torch.int16, device=dev)

    batch.set_timing(True)
    kms = {"analysis": [], "quantiser": [], "coding": [], "decode": []}

    # Optional serving shape (--overlap): consecutive encode calls are pipelined (solo_batch_set_async_join: the first analysis
    # chunk of step k + 1 starts while the last quantiser / coding chunks of step k still run) and the decode of step k is issued
    # after the encode of step k + 1; the bitstream buffers are double-buffered.  Default: encode then decode, one stream.
    overlap = args.overlap
    bits2, nb2 = [bits, torch.zeros_like(bits)], [nb, torch.zeros_like(nb)]
    step_no = [0]
    if overlap:
        batch.set_async_join(True)

    def step(k=None):
        if not overlap:
            batch.encode(pcm, bits, nb, st_e)
            batch.decode(bits, nb, None, out, st_d)
            return
        j = step_no[0] & 1
        batch.encode(pcm, bits2[j], nb2[j], st_e)
        if step_no[0] > 0:                                # the previous step's packets: wait for THAT encode call, then decode
            batch.wait_encode(1)
            batch.decode(bits2[1 - j], nb2[1 - j], None, out, st_d)
        step_no[0] += 1

    def drain():
        if overlap and step_no[0] > 0:
            j = (step_no[0] - 1) & 1
            batch.wait_encode(0)
            batch.decode(bits2[j], nb2[j], None, out, st_d)
            step_no[0] = 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()                                               # (the last step's decode: every packet is encoded AND decoded in the timed region)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel durations: a few extra steps outside the timed region (reading the events synchronises the stream)
    if overlap:
        batch.set_async_join(False)
        overlap = False
    for _ in range(min(3, args.steps)):
        step()
        for name, v in batch.last_kernel_ms().items():
            kms[name].append(v)
    enc_chunks = max(1, batch.last_encode_chunks())
    if world > 1:
        dt = sdist.max_over_ranks(dt, dist, dev)

    assert int(st_e.abs().max()) == 0 and int(st_d.abs().max()) == 0, "codec status != 0"
    kavg = {n: float(np.mean(v)) for n, v in kms.items()}
    dec_ms = kavg["decode"]
    # the encoder's kernels overlap each other (and, with --overlap, the decoder): encode alone is timed separately below
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        batch.encode(pcm, bits, nb, st_e)
    torch.cuda.synchronize()
    enc_only_ms = (time.perf_counter() - t1) / 3 * 1e3
    mean_payload = float(nb[:, :, 0].float().mean().item())
    packets_step = N * P
    value = world * packets_step * args.steps / dt

    if rank == 0:
        # algorithmic HBM bytes per packet of each kernel: stage input + stage output that has to cross HBM
        rec_in, rec_out, rec_code = 660.0, 964.0, 848.0       # sizeof SxNsqIn / SxNsqOut / SxCodeIn
        alg = {"analysis": 1280.0 + 2 * rec_in + rec_code, "quantiser": 2 * rec_in + 2 * rec_out,
               "coding": 2 * rec_out + rec_code + mean_payload + 4.0, "decode": mean_payload + 4.0 + 1280.0}
        kname = {"analysis": "solo_enc_analysis_kernel", "quantiser": "solo_nsq_kernel", "coding": "solo_enc_coding_kernel",
                 "decode": "solo_decode_kernel"}
        # the encoder kernels run as a pipeline over chunks of the step's packets: kavg = sum over the launches of a step
        nl = {n: (enc_chunks if n != "decode" else 1) for n in kavg}
        kernels = {kname[n]: {"launches_per_step": nl[n], "avg_launch_ms": round(kavg[n] / nl[n], 3),
                              "algorithmic_bytes_per_launch": int(alg[n] * packets_step / nl[n]),
                              "achieved_GBps": round(alg[n] * packets_step / (kavg[n] * 1e-3) / 1e9, 4)} for n in kavg}
        dom = max(kavg, key=kavg.get)
        enc_bytes = packets_step * alg[dom] / nl[dom]
        achieved = enc_bytes / (kavg[dom] / nl[dom] * 1e-3) / 1e9
        res = {
            "metric": "40 ms frames/sec (encode+decode) per GPU; concurrent real-time WB streams @1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "40ms packets/s (encode+decode)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32 fixed point (int16 PCM, Q-format arithmetic, bit-exact)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d synthetic 16 kHz WB streams per GPU, full encode -> two-description "
                                   "bitstream -> decode round trip with BWE resynthesis, 13.6 kbps, %d packets/stream/step" % (N, P),
                       "streams_per_gpu": N, "packets_per_stream_per_step": P, "mean_payload_bytes": round(mean_payload, 2),
                       "schedule": ("consecutive steps pipelined: encode of step k+1 issued before the decode of step k "
                                    "(solo_batch_set_async_join, double-buffered bitstreams)" if args.overlap
                                    else "encode then decode on one stream")},
            "realtime_streams": round(value / 25.0, 1),
            "encode_only_packets_per_s": round(packets_step / (enc_only_ms * 1e-3), 1),
            "decode_only_packets_per_s": round(packets_step / (dec_ms * 1e-3), 1),
            "roofline": {"kernel": kname[dom], "bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": None,
                         "avg_launch_ms": round(kavg[dom] / nl[dom], 3), "launches_per_step": nl[dom],
                         "algorithmic_bytes_per_launch": int(enc_bytes),
                         "note": "serial fixed-point recursions: latency / issue bound, not HBM bound (DESIGN.md section 4)"},
            "kernels": kernels,
        }
        tr = os.path.join(HERE, "profiles", "hbm_traffic.json")   # PMC-derived bytes per launch, collected separately
        if os.path.exists(tr):
            try:
                res["roofline"]["traffic"] = int(json.load(open(tr)).get(kname[dom] + "_bytes_per_packet") * packets_step / nl[dom])   # per launch
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_packets, args.cpu_packets_per_stream)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
