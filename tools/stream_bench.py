"""Real-time factor of the streaming mode (BASELINE config 5): S concurrent streams per GPU, 250 ms hops, every hop
emits each stream's score of its last 4 s.  Two scorers with bit-identical outputs (afx/streaming.py), reference-exact:
  sliding      the whole model on the window every hop (round 1);
  incremental  conv layers 0-5 cached per absolute frame, only the new 800/400/.../25 frames computed per hop;
and the labelled NON-reference mode config 5 names (a different, block-causal function: oracle/streaming.py):
  kv-cached    cached keys / values of the last 16 chunks, only the chunk's 12-13 new frames through the trunk.

    python tools/stream_bench.py [--workload conformer_student|xlsr_aasist] [--streams 1 64 512 2048]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/stream_bench.py --gpus N ...

With --gpus N every rank pins its own S streams to its GPU (state lives there; nothing is exchanged on the data path);
the hop time reported is the max over ranks (one RCCL all-reduce of a scalar, outside the timed hops), the stream
count the sum."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")]
from afx import engine, synth  # noqa: E402
from afx.streaming import IncrementalScorer, KVCachedScorer, SlidingWindowScorer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="conformer_student", choices=["conformer_student", "xlsr_aasist"])
    ap.add_argument("--streams", type=int, nargs="*", default=[1, 64, 512, 2048])
    ap.add_argument("--hops", type=int, default=10)
    ap.add_argument("--modes", nargs="*", default=["sliding", "incremental", "kv-cached"])
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    arch, oname, nl = ("conformer", "ConformerModel", 6) if args.workload == "conformer_student" else ("xlsr_aasist", "XLSR_AASIST", 24)
    sd = synth.model_state_dict(oname, n_layers=nl)
    eng = engine.Engine(arch, n_layers=nl, dtype="fp16")
    eng.load_state_dict(sd)
    W, H = 64000, 4000
    for S in args.streams:
        line = f"{args.workload}, {world} GPU(s) x {S} streams:"
        for name in args.modes:
            try:
                sc = {"sliding": lambda: SlidingWindowScorer(eng, S, window=W, hop=H), "incremental": lambda: IncrementalScorer(eng, sd, S, window=W, hop=H),
                      "kv-cached": lambda: KVCachedScorer(eng, sd, S, window=W, hop=H)}[name]()
            except Exception as exc:  # (the K / V rings of 24 layers are 38 MB per stream: 2 048 streams of the teacher do not fit beside the rest)
                line += f"  {name} n/a ({str(exc)[:60]})"
                continue
            chunk = (0.1 * torch.randn(S, H, generator=torch.Generator().manual_seed(rank))).cuda()
            for _ in range(W // H + 2):  # fill the window, reach the steady state
                sc.push(chunk)
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.hops):
                sc.push(chunk)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.hops
            if dist:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = t.item()
            line += f"  {name} {dt * 1e3:8.2f} ms/hop RTF {dt / 0.25:6.3f} ({world * S / dt:8.0f} scores/s)"
            del sc
        if rank == 0:
            print(line, flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
