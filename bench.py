#!/usr/bin/env python
"""Headline benchmark: DCCRN decode throughput (utterances/s, RTF) on 16 kHz 4 s clips.

One "step" = one pass of the whole hot path (unit-RMS normalise -> STFT -> compress -> DCCRN -> mask ->
decompress -> iSTFT -> /c) over one batch of B synthetic clips per GPU, inputs resident in HBM, outputs
device-resident on rank 0 after the RCCL gather of enhanced waveforms (N > 1).  BASELINE.json configs[2]:
DCCRN complex-mask, compressed input (p_in 0.5 / p_out 2.0), batch 256 per MI355X, utterances sharded across
ranks (weak scaling: per-GPU batch fixed).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
  bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or from a bare shell - then it starts its own
  ranks that way.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CLIP_SAMPLES = 64000          # 4 s @ 16 kHz
CLIP_SECONDS = 4.0
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense f32 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E peak (MI355X_MICROARCH.md)
DCCRN_GFLOP_PER_UTT = 53.4    # algorithmic 2*MAC per 4 s utterance (T = 501)


def dccrn_conv_bytes(B, T=501):
    """Algorithmic HBM bytes of the 20 tap-table GEMM launches of one DCCRN step (DESIGN.md 'Measurement'): every
    launch reads its input activations once and writes its output once (fp32; weights are < 0.1 %).  Encoder: 6 convs;
    decoder: 5 transposed convs = 2 frequency-parity launches each, both reading the (previous, skip) pair, and the last
    one (64 -> 2 channels) as one launch for both classes; LSTM input / output projections: 3 launches.  Channels are real
    counts (complex = 2 x)."""
    ch = [2, 32, 64, 128, 256, 256, 256]
    F = [257, 129, 65, 33, 17, 9, 5]
    act = [ch[i] * F[i] * T for i in range(7)]                 # floats per utterance at each encoder level
    rd = sum(act[0:6]) + 2 * sum(2 * act[i] for i in range(2, 7)) + 2 * act[1]
    wr = sum(act[1:7]) + sum(act[0:6])
    lstm = 2 * (256 * 5 * T) + 2 * (4 * 256 * T) * 2 + 2 * (256 * 5 * T)    # in-proj, 2 x (gates), out-proj
    return 4.0 * B * (rd + lstm), 4.0 * B * (wr + lstm)


def pmc_traffic():
    """HBM bytes per launch of the gc_kernel family from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_summary.py -> profiles/r01_pmc_dccrn.json); None when the summary is absent."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_dccrn.json')))
    if not cands:
        return None
    with open(cands[-1]) as f:            # the latest round's summary
        d = json.load(f)['gc_family']
    d['source'] = os.path.relpath(cands[-1], ROOT)
    return d


def _cpu_worker(args):
    """One CPU worker = one copy of the reference's batch-1 decode loop pinned to a single BLAS thread."""
    seed, p_in, p_out, clips = args
    import se_amd  # noqa: F401
    from se_amd import synth, schemas
    from oracle import decode as D
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:       # threadpoolctl missing: OMP/BLAS env limits set by the parent still apply
        import contextlib
        ctx = contextlib.nullcontext()
    sd = synth.synth_state_dict(schemas.dccrn_schema(), seed)
    with ctx:
        D.enhance_dccrn(sd, synth.synth_clip(0, 'speech', CLIP_SAMPLES), p_in, p_out)      # warm-up (imports, page-in)
        t0 = time.time()
        for n in range(clips):
            D.enhance_dccrn(sd, synth.synth_clip(n, 'speech', CLIP_SAMPLES), p_in, p_out)
        return time.time() - t0


def cpu_baseline(seed, p_in, p_out):
    """The numpy oracle (a port of the reference decode loop) on the host cores, bounded sample of the same workload:
    whole-path decode of 4 s clips, one at a time like the reference's batch-1 loop.  The port's BLAS calls do not scale
    with threads (a 256-thread run is slower than one thread), so the all-core figure is utterance-parallel: P
    single-thread worker processes (`bench.py --cpu-worker`), each a copy of the reference loop - how the reference
    would be spread over a host."""
    import subprocess
    ncpu = os.cpu_count() or 1
    clips = 2
    cpu_model, cores = 'unknown', set()
    try:
        with open('/proc/cpuinfo') as f:
            phys = None
            for ln in f:
                if ln.startswith('model name') and cpu_model == 'unknown':
                    cpu_model = ln.split(':', 1)[1].strip()
                elif ln.startswith('physical id'):
                    phys = ln.split(':', 1)[1].strip()
                elif ln.startswith('core id'):
                    cores.add((phys, ln.split(':', 1)[1].strip()))
    except OSError:
        pass
    # one single-thread copy of the reference loop per PHYSICAL core: measured on the 2 x 64-core / 256-thread host of the
    # GPU box, 64 workers decode 12.0 utt/s, 256 (one per SMT thread) only 5.3 - the port is memory-bound
    P = len(cores) if cores else max(1, ncpu // 2)
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1',
               HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', str(clips), '--cpu-worker-args',
           f'{seed},{p_in},{p_out}']
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
    spans = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(5.0, 240.0 - (time.time() - t0)))
            spans.append(float(out.strip().splitlines()[-1]))
        except Exception:           # a worker that is late or died is dropped from the sample (and reaped)
            p.kill()
            p.communicate()
    if not spans:
        return {"value": None, "unit": "utt/s", "cores": 0, "kind": "port", "sample": "no CPU worker finished within 240 s",
                "cpu_model": cpu_model}
    wall = max(spans)
    return {"value": round(len(spans) * clips / wall, 3), "unit": "utt/s", "cores": len(spans), "kind": "port",
            "sample": f"{len(spans)} single-thread worker processes x {clips} x 4 s clips, batch-1 loop each, numpy oracle "
                      f"(oracle/decode.py:enhance_dccrn), slowest worker {wall:.1f} s, {time.time() - t0:.1f} s with start-up; "
                      f"host has {ncpu} logical CPUs ({cpu_model})",
            "cpu_model": cpu_model, "logical_cpus": ncpu, "physical_cores": P,
            "value_per_worker_best": round(clips / min(spans), 4)}


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start one rank per GPU under torch.distributed.run (rendezvous on
    127.0.0.1, a free port) with the same arguments and hand back its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='clips per GPU per step')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend of the waveform gather (nccl = RCCL over xGMI; gloo stages through the host)')
    ap.add_argument('--force-pg', action='store_true',
                    help='initialise the process group and run the gather collective even with one rank (RCCL smoke on a 1-GPU box)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--cpu-worker', type=int, default=0, help='internal: decode N clips with the numpy oracle, print seconds')
    ap.add_argument('--cpu-worker-args', type=str, default='14,0.5,2.0')
    args = ap.parse_args()
    if args.cpu_worker:
        seed, p_in, p_out = args.cpu_worker_args.split(',')
        print(_cpu_worker((int(seed), float(p_in), float(p_out), args.cpu_worker)), flush=True)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import se_amd  # noqa: F401
    from se_amd import synth
    from se_amd.models import DCCRN

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # one rank per GPU; SE_BENCH_SHARE_GPU=1 (the 1-GPU test box) folds the ranks onto the devices that exist
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and os.environ.get('SE_BENCH_SHARE_GPU') != '1':
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible")
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    use_pg = world > 1 or args.force_pg
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank) if args.backend == 'nccl' else None)

    B, p_in, p_out, seed = args.batch, 0.5, 2.0, 14
    model = DCCRN(rnn_units=256, masking_mode='E', use_clstm=True, kernel_num=[32, 64, 128, 256, 256, 256],
                  device=local_rank, max_batch=B, max_samples=CLIP_SAMPLES, p_in=p_in, p_out=p_out)
    model.load_synthetic(seed)
    eng = model.engine
    # synthetic clips: 16 distinct speech-like clips tiled to the batch, distinct per rank
    base = synth.synth_batch(16, 'speech', CLIP_SAMPLES, seed0=100 + 16 * rank)
    wav = torch.from_numpy(np.tile(base, ((B + 15) // 16, 1))[:B].copy()).cuda()
    n_out = eng.output_samples(CLIP_SAMPLES)
    from se_amd import shard
    # enhanced waveforms of every rank end up device-resident on rank 0: asynchronous RCCL gather, double-buffered so
    # that the gather of step k overlaps the compute of step k + 1 (se_amd/shard.py:GatherPipe); N = 1: no collective
    pipe = shard.GatherPipe(B, n_out, torch.device('cuda', local_rank), dst=0, single_rank=args.force_pg)
    out = None

    def step():
        nonlocal out
        out = pipe.slot()
        eng.enhance_batch(wav, out)
        pipe.submit()

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pipe.finish()
    fence()
    if not args.no_profile:
        eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rows = pipe.finish()    # every gather of the timed steps has landed on rank 0
    fence()
    dt = time.perf_counter() - t0
    prof = eng.get_profile() if not args.no_profile else None
    stages = eng.get_stage_profile() if not args.no_profile else None
    eng.set_profiling(False)

    if use_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        assert bool(torch.isfinite(out).all()), "non-finite output"
        if use_pg:          # the gathered rows of the last step: rank r's rows are its own clips' enhancement
            assert len(rows) == world and all(bool(torch.isfinite(r).all()) for r in rows), "non-finite gathered rows"
        utts = world * B * args.steps
        value = utts / dt
        res = {
            "metric": "utterances/sec, 16 kHz 4 s clips, DCCRN decode (STFT->network->iSTFT)",
            "value": round(value, 2), "unit": "utt/s",
            "rtf": round(dt / (utts * CLIP_SECONDS), 8), "x_realtime": round(value * CLIP_SECONDS, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded speech-like clips; random-init DCCRN weights)",
            "config": {"workload": "BASELINE configs[2]: DCCRN complex-mask, compressed input (0.5/2.0), "
                                   "16 kHz x 4 s clips, batch %d per MI355X, utterance-sharded" % B,
                       "batch_per_gpu": B, "global_batch": B * world, "clip_samples": CLIP_SAMPLES,
                       "parallelism": "utterance shard x%d + RCCL gather to rank 0" % world},
        }
        if prof and prof['gemm_ms'] > 0:
            ach = prof['gemm_flops'] / (prof['gemm_ms'] * 1e-3) / 1e12
            pmc = pmc_traffic()
            rd_b, wr_b = dccrn_conv_bytes(B)
            res["roofline"] = {
                "bound": "mfma", "kernel": "se::gc_kernel<BM,BN,..> (f32-MFMA tap-table implicit-GEMM conv)",
                "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": pmc['traffic_GB_per_launch'] if pmc and B == 256 else None,
                "traffic_unit": "GB of HBM traffic per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 "
                                "--pmc passes of this command at batch 256: %s)" % (pmc['source'] if pmc else 'profiles/'),
                "algorithmic_GB_per_launch": round((rd_b + wr_b) / 1e9 / max(prof['gemm_launches'], 1), 3),
                "launches_per_step": prof['gemm_launches'],
                "algorithmic_gflop_per_step": round(prof['gemm_flops'] / 1e9, 1),
                "kernel_ms_per_step": round(prof['gemm_ms'], 3),
                "avg_launch_us": round(1e3 * prof['gemm_ms'] / max(prof['gemm_launches'], 1), 2),
                "share_of_step": round(prof['gemm_ms'] / (1e3 * dt / args.steps), 4),
            }
            res["roofline_whole_path"] = {
                "achieved": round(value / world * DCCRN_GFLOP_PER_UTT / 1e3, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s per GPU (utt/s x 53.4 GFLOP/utt, SURVEY 8(d))",
                "frac": round(value / world * DCCRN_GFLOP_PER_UTT / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)}
        if stages:
            # the HBM-bound front / back-end kernels of the same (last timed) step: algorithmic bytes over
            # their HIP-event time, against the HBM peak
            names = {'rms': 'se::rms_partial_kernel + rms_finish_kernel (c = sqrt(L / sum x^2))',
                     'stft': 'se::stft_kernel<512> (frame, reflect pad, Hann, LDS Stockham FFT, x*c, |X|^p)',
                     'mask': 'se::dccrn_mask_kernel (E mask + decompress)',
                     'istft': 'se::istft_ola_kernel<512> (inverse FFT + overlap-add + /c, frames stay in LDS)'}
            res["roofline_stages"] = [
                {"stage": k, "kernel": names[k], "bound": "hbm", "ms_per_step": round(v['ms'], 4),
                 "algorithmic_GB_per_step": round(v['bytes'] / 1e9, 4),
                 "achieved": round(v['bytes'] / 1e9 / max(v['ms'] * 1e-3, 1e-12), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(v['bytes'] / 1e9 / max(v['ms'] * 1e-3, 1e-12) / HBM_PEAK_GBS, 4),
                 "share_of_step": round(v['ms'] / (1e3 * dt / args.steps), 5)}
                for k, v in stages.items() if v['launches'] > 0]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(seed, p_in, p_out)
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
