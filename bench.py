#!/usr/bin/env python3
"""Headline benchmark: faces/s end-to-end (det + align + embed + match) on MI355X.

Workload (BASELINE.json configs[1]): SCRFD-10G + ArcFace-R50 (fp16 MFMA, fp32 accumulate), batch = 64
synthetic 640x640 frames per GPU, 1k-entry gallery, F = 1 face kept per frame (the reference's
--max-num 1, main.py:55-60).  One process per GPU; frames shard by rank, one RCCL all-gather of the
per-rank unit embeddings before the gallery match (weak scaling: 64 frames per GPU).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

By default TWO batches are in flight per GPU (--streams 2): steps are issued round-robin to two independent
library contexts on separate HIP streams, so the latency-bound kernels of one batch (IResNet at 64 faces has
only 100-400 tiles per layer) overlap the other batch's.  Every step is still one full pass over one batch,
all K steps complete inside the timed region; ms_per_step is elapsed / K.

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline:     all MFMA conv launches of one step (the dominant kernel family conv_mfma_kernel<...>):
                algorithmic FLOPs (2 x MACs of the true channel counts) / their summed device time,
                timed with HIP events on the library's stream, against the 2.5 PFLOP/s dense fp16 peak
  cpu_baseline: the oracle (fp32 torch-CPU restatement of the reference path) on this box's host cores
                for a bounded sample of the same frames (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may actually use (the GPU box grants a share of the host, not all of it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:        # cgroup v2 quota
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("FID_CPU_THREADS", "16"))))


def build(ctx, batch, F, gallery_size, calib_frames):
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias
    det_net = archs.scrfd_10g((640, 640))
    det_P = archs.synth_params(det_net, seed=0)
    det_P, shift = calibrate_detector_bias(ctx, det_net, det_P, calib_frames, target=48)
    rec_net = archs.iresnet50()
    rec_P = archs.synth_params(rec_net, seed=0)
    det = CompiledNet(ctx, det_net, det_P, max_batch=batch)
    rec = CompiledNet(ctx, rec_net, rec_P, max_batch=batch * F)
    g = np.random.default_rng(99).standard_normal((gallery_size, 512)).astype(np.float32)
    gallery = Gallery(ctx, g)
    return det_net, det_P, rec_net, rec_P, det, rec, gallery, g


def mfma_roofline(pipe, frames_dev, batch, F):
    """HIP-event time of every MFMA conv launch of one step vs its algorithmic FLOPs."""
    from scrfd_arcface_facerecognition_amd.lower import OP_CONV, OP_STEM, OP_STEMFUSED
    tot_ms, tot_flop, launches, per_net = 0.0, 0.0, 0, {}
    for name, cn, imgs, n in (("scrfd_10g", pipe.det, frames_dev, batch), ("arcface_r50", pipe.rec, pipe.crops, batch * F)):
        best = None
        for _ in range(3):
            ms = cn.run_profiled(imgs, n)
            best = ms if best is None else np.minimum(best, ms)
        t, fl, k = 0.0, 0.0, 0
        by_name = {nd.name: nd for nd in cn.net.nodes}
        for oi, names in enumerate(cn.low.op_nodes):
            if int(cn.low.ops[oi, 0]) not in (OP_CONV, OP_STEM, OP_STEMFUSED):
                continue
            macs = sum(node_macs(cn.net, by_name[nm]) for nm in names)
            t += float(best[oi]); fl += 2.0 * macs * n; k += 1
        per_net[name] = {"ms": round(t, 4), "tflops": round(fl / t / 1e9, 1), "launches": k,
                         "net_ms_all_ops": round(float(best.sum()), 4)}
        tot_ms += t; tot_flop += fl; launches += k
    ach = tot_flop / tot_ms / 1e9
    traffic = None       # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
        traffic = round(pmc["hbm_bytes_per_launch_corrected"])
    except Exception:
        pass
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP16_TFLOPS, 4), "traffic": traffic,
            "kernel": "MFMA conv kernels of one step: conv_mfma_kernel<*> / conv_mfma_dma_kernel<*> / conv3x3_direct<*> / "
                      "conv3x3_chunked<*> / conv3x3_pc<*> / conv3x3_pcr / scrfd_stem_fused<*> / stem_conv_mfma<*> (per layer the autotuner's pick)", "launches": launches,
            "avg_us_per_launch": round(tot_ms * 1e3 / launches, 2), "gflop_per_step": round(tot_flop / 1e9, 1),
            "per_net": per_net}


_shape_cache = {}


def node_macs(net, node):
    from scrfd_arcface_facerecognition_amd.archs import infer_shapes
    key = id(net)
    if key not in _shape_cache:
        _shape_cache[key] = infer_shapes(net)
    shp = _shape_cache[key]
    if node.kind == "conv":
        _, ho, wo = shp[node.name]
        return ho * wo * node.cout * (node.cin // node.groups) * node.k * node.k
    if node.kind == "fc":
        return node.c * node.h * node.w * node.cout
    if node.kind == "dethead":
        _, h, w = shp[node.name]
        return h * w * node.num_anchors * 15 * node.cin * node.k * node.k
    return 0


def cpu_baseline(frames, det_net, det_P, rec_net, rec_P, gallery, n_frames):
    """The oracle on the host cores, reference structure (frame by frame, face by face, python gallery loop)."""
    import torch
    from oracle import pipeline as opipe
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: warm-up frame on {cores} threads")
    opipe.process_frame(frames[0], det_net, det_P, rec_net, rec_P, gallery, max_num=1)     # warm-up
    t0 = time.perf_counter()
    faces, done = 0, 0
    while done < n_frames:
        faces += len(opipe.process_frame(frames[done % len(frames)], det_net, det_P, rec_net, rec_P, gallery, max_num=1))
        done += 1
        if done % 16 == 0:
            log(f"cpu_baseline: {done} frames, {time.perf_counter() - t0:.1f} s")
        if time.perf_counter() - t0 > 15:          # bounded sample: ~15 s of CPU work
            break
    dt = time.perf_counter() - t0
    return {"value": round(faces / dt, 3), "unit": "faces/s", "cores": cores, "kind": "port",
            "sample": f"{done} frames of the same synthetic stream, frame by frame like the reference: SCRFD-10G + "
                      f"ArcFace-R50 fp32 torch-CPU oracle + python gallery loop (1k), {dt:.1f} s on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--faces-per-frame", type=int, default=1)
    ap.add_argument("--gallery", type=int, default=1000)
    ap.add_argument("--cpu-frames", type=int, default=512, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("FID_BENCH_STREAMS", "2")),
                    help="pipelines in flight per GPU: steps are issued round-robin to this many independent "
                         "contexts/HIP streams so that the small kernels of one batch overlap another batch's")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    log("importing torch")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from scrfd_arcface_facerecognition_amd._lib import Context
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, run_step_distributed
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    stream = torch.cuda.Stream()
    ctx = Context(local_rank, stream.cuda_stream)
    B, F = args.batch, args.faces_per_frame
    calib = np.random.default_rng(1234).integers(0, 256, (8, 640, 640, 3), dtype=np.uint8)   # same on every rank
    frames = np.random.default_rng(1234 + rank).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
    log("building nets (synthetic weights, detector bias calibration)")
    det_net, det_P, rec_net, rec_P, det, rec, gallery, gal_host = build(ctx, B, F, args.gallery, calib)
    log("nets resident")

    class Lane:
        """one independent pipeline: its own HIP stream / library context, nets, buffers"""
        pass

    lanes = []
    for li in range(max(1, args.streams)):
        ln = Lane()
        if li == 0:
            ln.stream, ln.ctx, ln.det, ln.rec, ln.gallery = stream, ctx, det, rec, gallery
        else:
            ln.stream = torch.cuda.Stream()
            ln.ctx = Context(local_rank, ln.stream.cuda_stream)
            ln.det = CompiledNet(ln.ctx, det_net, det_P, max_batch=B)
            ln.rec = CompiledNet(ln.ctx, rec_net, rec_P, max_batch=B * F)
            ln.gallery = Gallery(ln.ctx, gal_host)
        with torch.cuda.stream(ln.stream):
            ln.q_local = torch.empty((B * F, 512), dtype=torch.float16, device="cuda")
            ln.q_all = torch.empty((world * B * F, 512), dtype=torch.float16, device="cuda") if world > 1 else None
            ln.pipe = FacePipeline(ln.ctx, ln.det, ln.rec, batch=B, faces_per_frame=F, q_buffer=ln.q_local)
            ln.frames_dev = ln.ctx.to_device(frames)     # resident in HBM before the timed region
        lanes.append(ln)
    pipe, frames_dev = lanes[0].pipe, lanes[0].frames_dev

    def step(i):
        ln = lanes[i % len(lanes)]
        with torch.cuda.stream(ln.stream):
            if world > 1:
                run_step_distributed(ln.pipe, ln.frames_dev, 640, 640, ln.gallery, 0.4, ln.q_local, ln.q_all, dist)
            else:
                ln.pipe.run_step(ln.frames_dev, 640, 640, ln.gallery, 0.4)

    if True:
        for i in range(len(lanes)):              # every lane tunes its kernels alone on the GPU
            step(i)
            torch.cuda.synchronize()
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        log("warm-up done")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")

        pipe.post.check()
        counts = pipe.post.counts.download()
        faces_step = int(np.minimum(counts, F).sum())
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            fc = torch.tensor([faces_step], dtype=torch.int64, device="cuda")
            dist.all_reduce(fc, op=dist.ReduceOp.SUM)
            faces_total_step = int(fc.item())
        else:
            faces_total_step = faces_step

        out = None
        if rank == 0:
            value = faces_total_step * args.steps / elapsed
            out = {
                "metric": "faces/sec end-to-end (det+align+embed+match)", "value": round(value, 2), "unit": "faces/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                "config": {"workload": "SCRFD-10G + ArcFace-R50, 64 synthetic 640x640 frames per GPU per step, "
                                       f"F={F} face/frame (max_num), {args.gallery}-entry gallery, random-init weights (seed 0)",
                           "frames_per_gpu": B, "faces_per_step": faces_total_step, "gallery": args.gallery,
                           "parallelism": (f"frames sharded over {world} GPU(s), 1 all-gather of embeddings" if world > 1 else "1 GPU")
                                          + (f", {len(lanes)} batches in flight per GPU on separate HIP streams" if len(lanes) > 1 else "")},
            }
            if not args.no_roofline:
                log("roofline: per-op HIP-event timing")
                out["roofline"] = mfma_roofline(pipe, frames_dev, B, F)
        if rank == 0 and world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(frames, det_net, det_P, rec_net, rec_P, gal_host, args.cpu_frames)
        if rank == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
