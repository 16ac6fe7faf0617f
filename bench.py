#!/usr/bin/env python3
"""Headline benchmark: utterances/s on 4 s @ 16 kHz clips (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of the hot path (raw waveform -> logits -> bonafide score)
over one batch of synthetic utterances that is already resident in HBM.  At N=1 the
workload is BASELINE.json configs[1]: the Conformer student (XLS-R first-6 trunk +
4 Conformer blocks, emb 144) at batch 64.  With N>1 every rank scores its own 64
utterances (weak scaling, no data-path collective) and the scores are all-gathered
over RCCL each step, inside the timed region.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (the 128x128 MFMA GEMM): algorithmic FLOPs per
                  launch / its average launch duration, timed with hipEvents on the
                  launch stream in a second, instrumented pass over the same K steps
                  (the timed region itself runs un-instrumented);
  cpu_baseline -- the CPU oracle (kind "port") timed on this box's host cores on a
                  bounded sample of the same workload, plus the GPU-vs-oracle parity
                  of that sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# dense matrix-core peaks, MI355X_MICROARCH.md "Chip-level parameters" (fp32: the exact-mode MFMA, 1/16 of fp16)
MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}
# measured issue ceiling of the instruction the GEMM kernels use (v_mfma_f32_16x16x32_{f16,bf16} issues at half
# the rate of 32x32x16; v_mfma_f32_16x16x4_f32 for exact mode): tools/peak_probe.hip, profiles/r01_peak_probe.txt
MFMA_INSTR_CEILING_TFLOPS = {"fp16": 1316.7, "bf16": 1316.7, "fp32": 135.0}

WORKLOADS = {
    # name: (engine arch, oracle model name, trunk layers, GFLOP per utterance)
    "conformer_student": ("conformer", "ConformerModel", 6, 55.29),
    "xlsr_aasist": ("xlsr_aasist", "XLSR_AASIST", 24, 148.67),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="conformer_student", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default 64 / 16)")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--dtype", default=os.environ.get("AFX_DTYPE", "fp16"), choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--cpu-sample", type=int, default=8, help="utterances timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X-native path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("AFX_FORCE_DIST") == "1"  # the env knob rehearses the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    from afx import engine, synth
    from afx.dist import all_gather_scores

    arch, oname, n_layers, gflop = WORKLOADS[args.workload]
    B = args.batch or (64 if args.workload == "conformer_student" else 16)
    L = int(args.seconds * 16000)
    sd = synth.model_state_dict(oname, n_layers=n_layers)
    eng = engine.Engine(arch, n_layers=n_layers, dtype=args.dtype)
    eng.load_state_dict(sd)
    wave = synth.waveforms(B, L, batch_idx=rank).cuda()  # resident in HBM before the timed region
    idx = torch.arange(rank * B, (rank + 1) * B, dtype=torch.int32, device="cuda")

    def step():
        logits = eng.forward(wave)
        scores = logits[:, 1]
        if use_dist:
            return all_gather_scores(idx, scores, world)
        return idx, scores

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        assert out[0].numel() == world * B
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- instrumented pass: per-kernel-class time from hipEvents on the launch stream ----
    eng.profile_begin()
    for _ in range(args.steps):
        eng.forward(wave)
    prof = eng.profile_end()
    gemm_classes = {k: v for k, v in prof.items() if k.startswith("gemm") and v["launches"]}
    dom = max(gemm_classes, key=lambda k: gemm_classes[k]["ms"])  # the tile instance with the most time
    g = gemm_classes[dom]
    gemm_tflops = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    all_ms = sum(v["ms"] for v in gemm_classes.values())
    all_fl = sum(v["flops"] for v in gemm_classes.values())
    roofline = {
        "bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
        "frac": round(gemm_tflops / MFMA_PEAK_TFLOPS[args.dtype], 4), "traffic": None,
        "kernel": "afx::" + dom.replace("<", f"<{args.dtype},").replace("x", ","),
        "avg_launch_us": round(g["ms"] * 1e3 / max(g["launches"], 1), 2),
        "launches_per_step": g["launches"] // args.steps,
        "gflop_per_launch": round(g["flops"] / max(g["launches"], 1) / 1e9, 3),
        "all_gemm_instances_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0,
        # the QKV and FC1 products run as the 8-phase kernel on the rows that fill whole rounds of CUs plus a
        # 128x128-tile kernel on the remaining rows; the pair is ONE timed launch here, two rows in rocprofv3's
        # kernel stats: avg_launch_us = avg(8-phase) + (remainder launches / 8-phase launches) x avg(remainder)
        "launch_note": "one launch = one GEMM of the path; round-split GEMMs (8-phase kernel + 128x128 remainder kernel) are timed as one",
    }
    # what the MFMA instruction these kernels issue sustains on this chip with nothing else in the loop
    # (tools/peak_probe.hip, random operands, profiles/r01_peak_probe.txt); `peak`/`frac` stay the nominal ones
    if args.dtype in MFMA_INSTR_CEILING_TFLOPS:
        roofline["instr_ceiling"] = MFMA_INSTR_CEILING_TFLOPS[args.dtype]
        roofline["frac_of_instr_ceiling"] = round(gemm_tflops / MFMA_INSTR_CEILING_TFLOPS[args.dtype], 4)
    breakdown = {k: round(v["ms"] / args.steps, 4) for k, v in prof.items() if v["launches"]}
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh: FETCH_SIZE x 2
    # + WRITE_SIZE, mean bytes per launch on this workload); counters cannot be read from inside this process
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.workload == "conformer_student" and B == 64 and args.dtype != "fp32":
        kern = dom.split("<")[0] + "<afx::" + args.dtype.upper() + ", " + dom.split("<")[1].rstrip(">").split(",")[0].replace("x", ", ")
        for name, rec in json.load(open(pmc_path))["kernels"].items():
            if name.startswith("afx::" + kern):
                roofline["traffic"] = round(rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"])
                roofline["traffic_unit"] = "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)"

    result = {
        "metric": "utterances/sec (4 s @ 16 kHz)" if args.seconds == 4.0 else f"utterances/sec ({args.seconds:g} s @ 16 kHz)", "value": round(value, 2), "unit": "utterances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {oname} ({n_layers}-layer XLS-R trunk), batch {B} per GPU, "
                               f"{args.seconds:g} s clips @ 16 kHz, random-init weights",
                   "global_batch": world * B, "samples_per_utterance": L,
                   "parallelism": f"dp{world} (utterance sharding, RCCL score all-gather)" if world > 1 else "single GPU"},
        "model_tflops": round(value * gflop / 1e3, 1),
        "device_ms_per_step": round(ev0.elapsed_time(ev1) / args.steps, 3),
        "roofline": roofline,
        "kernel_ms_per_step": breakdown,
    }

    # ---- the same K steps with the host hand-over inside: pinned fp32 waveform H2D (256 KB per
    # utterance) + forward + D2H of the scores (main.py:209-213).  Reported beside, never as, `value`.
    if world == 1:
        host_wave = wave.cpu().pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            host_scores = eng.forward(host_wave.to("cuda", non_blocking=True))[:, 1].cpu()
        e2e = time.perf_counter() - t0
        result["with_pcie"] = {"value": round(B * args.steps / e2e, 2), "unit": "utterances/s",
                               "note": "H2D of the batch + forward + D2H of the scores every step, one stream, no overlap"}
        # the scoring loop's form (afx.harness.prefetch_to_device): next batch's H2D on a side stream
        from afx.harness import prefetch_to_device
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [eng.forward(x)[:, 1] for _m, x in prefetch_to_device(((i, host_wave) for i in range(args.steps)), "cuda")]
        host_scores = torch.cat(outs).cpu()
        e2e = time.perf_counter() - t0
        result["with_pcie"]["overlapped"] = round(B * args.steps / e2e, 2)

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample (rank 0, N=1) ----
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import models as omodels
        n = min(args.cpu_sample, B)
        # a one-GPU box gets a 16-core share of the host (more threads only oversubscribe it)
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        torch.set_num_threads(cores)
        fwd = omodels.conformer_forward if arch == "conformer" else omodels.xlsr_aasist_forward
        cpu_wave = wave[:n].cpu()
        fwd(sd, cpu_wave[:1])  # warm the thread pool
        reps, cpu_s = 0, 0.0
        while cpu_s < 10.0 and reps < 50:  # a bounded sample of about 10-20 s of CPU work
            t0 = time.perf_counter()
            ref = fwd(sd, cpu_wave)
            cpu_s += time.perf_counter() - t0
            reps += 1
        got = eng.forward(wave)[:n].cpu()
        result["cpu_baseline"] = {
            "value": round(n * reps / cpu_s, 3), "unit": "utterances/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{reps} batched fp32 forward(s) of the CPU oracle (PyTorch CPU) over {n} of the same 4 s "
                      f"utterances, {cpu_s:.1f} s in all",
        }
        result["parity"] = {"max_abs_dlogit_vs_oracle": float((got - ref).abs().max()), "tolerance": 1e-3,
                            "utterances": n}
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
