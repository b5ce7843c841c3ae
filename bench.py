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
    counts (complex = 2 x).  The layers that run as three real products add x_r + x_i planes and k1..k3 tensors
    that are written and read back: implementation traffic, part of `traffic` (PMC), not of this figure."""
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
    # the counters describe the kernels they were collected on: the summary carries a hash of the kernel sources the DCCRN
    # decode runs through (tools/pmc_summary.py: csrc_sha16); when the tree's differ, the figure is STALE and the line says so
    # instead of quoting it (there is no .git on the GPU box to ask for an ancestry check)
    d['stale'] = d.get('csrc_sha16') != dccrn_csrc_sha16()
    return d


DCCRN_KERNEL_SOURCES = ['gemmconv.hip', 'gemmconv.h', 'gauss.h', 'model_dccrn.hip', 'layers.hip', 'layers.h', 'k_lstm.hip',
                        'k_misc.hip', 'k_stft2.hip', 'rnn.h', 'kernels.h', 'model.h', 'fastmath.h']


def dccrn_csrc_sha16():
    """sha256[:16] over the kernel sources of the DCCRN decode path (the files a PMC pass of this bench exercises)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, [d for d in os.listdir(ROOT) if d.endswith('_amd')][0], 'csrc')
    for fn in DCCRN_KERNEL_SOURCES:
        with open(os.path.join(csrc, fn), 'rb') as f:
            h.update(fn.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def _host_cpu():
    """(model string, physical core count, logical CPU count) of this host."""
    ncpu = os.cpu_count() or 1
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
    return cpu_model, (len(cores) if cores else max(1, ncpu // 2)), ncpu


def _cpu_worker(spec):
    """`bench.py --cpu-worker seed,p_in,p_out,P`: times oracle/dccrn_cpu.cpp (the C++ / OpenMP restatement of the DCCRN
    decode, pinned by tests/test_dccrn_cpu.py to the reference-generated fixtures) on this host and prints one JSON object.
    Runs in its own process so that OMP_PROC_BIND / OMP_PLACES take effect before the OpenMP runtime starts."""
    seed, p_in, p_out, P = spec.split(',')
    seed, p_in, p_out, P = int(seed), float(p_in), float(p_out), int(P)
    import se_amd  # noqa: F401
    from se_amd import synth, schemas
    from oracle.dccrn_cpu import DccrnCpu
    net = DccrnCpu(synth.synth_state_dict(schemas.dccrn_schema(), seed))
    clips = np.stack([synth.synth_clip(n, 'speech', CLIP_SAMPLES) for n in range(8)])
    net.enhance(clips[:1], p_in, p_out, threads=1, mode=0)                      # warm-up (page-in, first touch)
    out = {}
    t0 = time.perf_counter()
    net.enhance(clips[:3], p_in, p_out, threads=1, mode=0)
    out['one_thread_utt_s'] = 3 / (time.perf_counter() - t0)
    # the reference's shape: one clip after the other (`for file_id in file_list`) with 8 threads inside each layer - the
    # thread count of SURVEY 8(d)'s anchor for the reference's own modules (PyTorch CPU, 8 threads, batch 1: 3 - 24 utt/s)
    t8 = min(8, P)
    net.enhance(clips[:1], p_in, p_out, threads=t8, mode=0)
    t0 = time.perf_counter()
    net.enhance(clips[:8], p_in, p_out, threads=t8, mode=0)
    out['batch1_loop_8_threads_utt_s'] = 8 / (time.perf_counter() - t0)
    # how a host would be filled: P clips in flight, one core each
    big = np.tile(clips, ((2 * P + 7) // 8, 1))[:2 * P]
    net.enhance(big[:P], p_in, p_out, threads=P, mode=1)
    t0 = time.perf_counter()
    net.enhance(big, p_in, p_out, threads=P, mode=1)
    out['utterance_parallel_utt_s'] = 2 * P / (time.perf_counter() - t0)
    out['clips_utterance_parallel'] = 2 * P
    return out


def cpu_baseline(seed, p_in, p_out):
    """The CPU path next to the GPU number: a compiled C++ / OpenMP restatement of the same DCCRN decode loop
    (oracle/dccrn_cpu.cpp; kind "port" - the Python reference cannot travel to the GPU box), pinned by the same
    reference-generated fixtures as the numpy oracle, timed on this host at 1 thread and at all physical cores
    (SURVEY 8(d)): a bounded sample of the same workload, 4 s clips decoded whole-path."""
    import subprocess
    cpu_model, P, ncpu = _host_cpu()
    env = dict(os.environ, OMP_PROC_BIND='spread', OMP_PLACES='cores', OMP_NUM_THREADS=str(P), HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='')
    t0 = time.time()
    base = {"unit": "utt/s", "kind": "port", "language": "c++ (g++ -O3, OpenMP, AVX-512 / AVX2 clones of the conv micro-kernel)",
            "source": "oracle/dccrn_cpu.cpp (pinned by tests/test_dccrn_cpu.py to tests/golden/dccrn.npz)",
            "cpu_model": cpu_model, "logical_cpus": ncpu, "physical_cores": P}
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', f'{seed},{p_in},{p_out},{P}'], env=env,
                           capture_output=True, text=True, timeout=240)
        r = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as ex:
        return dict(base, value=None, cores=0, sample=f"CPU worker failed: {type(ex).__name__}")
    return dict(base, value=round(r['utterance_parallel_utt_s'], 2), cores=P,
                sample=f"{r['clips_utterance_parallel']} x 4 s clips, {P} in flight (one physical core each, OMP_PLACES=cores); "
                       f"also 3 clips on 1 thread and 8 clips one after the other with 8 threads inside each layer; "
                       f"{time.time() - t0:.1f} s with start-up",
                one_thread_utt_s=round(r['one_thread_utt_s'], 3),
                one_thread_gflops=round(r['one_thread_utt_s'] * DCCRN_GFLOP_PER_UTT, 1),
                batch1_loop_8_threads_utt_s=round(r['batch1_loop_8_threads_utt_s'], 2),
                all_cores_gflops=round(r['utterance_parallel_utt_s'] * DCCRN_GFLOP_PER_UTT, 1))


# BASELINE.json's other configurations on the same clock (a few unprofiled steps each, after the headline's timed region):
# (config, host class key, batch per GPU, SURVEY 8(d) GFLOP per 4 s utterance, constructor exponents)
OTHER_CONFIGS = [
    ("configs[0]: LSTM magnitude-mask, one 4 s clip (the reference's CPU plumbing case, here on the GPU)", 'lstm', 1, 17.5, {}),
    ("configs[1]: CRN real-spectrum mask, batch 64 x 4 s clips", 'crn', 64, 17.3, {}),
    ("configs[3]: FullSubNet full-band + sub-band LSTM, per-GPU utterance shard of 128 clips", 'fullsubnet', 128, 238.5, {}),
    ("configs[4]: Uformer dual-path complex conformer, per-GPU utterance shard of 256 clips", 'uformer', 256, 27.5, {}),
]


# The rest of the zoo (the north star's ten model families + the three `*_new` directories) on the same clock, batch
# 256 x 4 s clips: `roofline.zoo` (VERDICT r3 #4 - every network's figure driver-timed, not only the five BASELINE configs).  The
# `_new` variants do the arithmetic of their bases (cumulative LayerNorm in place of InstanceNorm): same GFLOP per utterance.
ZOO_CONFIGS = [
    ("GCRN (GCRN/GCRN_noncprs.py), batch 256", 'gcrn', 256, 13.2, {}),
    ("DPCRN (DPCRN/DPCRN.py), batch 256", 'dpcrn', 256, 4.8, {}),
    ("CTSNet (CTSNet/Step1_network.py + Step2_network.py), batch 256", 'ctsnet', 256, 25.6, {}),
    ("G2Net (G2Net_VB/gaf_net_320.py), batch 256", 'g2net', 256, 10.7, {}),
    ("TaylorSENet (TaylorSENet/TaylorSENet.py), batch 256", 'taylorsenet', 256, 31.3, {}),
    ("CTSNet_new (cLN), batch 256", 'ctsnet_new', 256, 25.6, {}),
    ("G2Net_new (cLN), batch 256", 'g2net_new', 256, 10.7, {}),
    ("TaylorSENet_new (cLN), batch 256", 'taylorsenet_new', 256, 31.3, {}),
]


def _build_model(name, local_rank, B, kw):
    from se_amd import models, models_new
    args = dict(device=local_rank, max_batch=B, max_samples=CLIP_SAMPLES, **kw)
    if name == 'ctsnet':
        return models.CTSNet(**args).load_synthetic(17, 18)
    if name == 'ctsnet_new':
        return models_new.CTSNet(**args).load_synthetic(17, 18)
    return models.MODEL_CLASSES[name](**args).load_synthetic(1)


def run_other_configs(torch, local_rank, steps, configs=None):
    from se_amd import synth
    rows = []
    base = synth.synth_batch(16, 'speech', CLIP_SAMPLES, seed0=300)
    for cfg, name, B, gflop, kw in (configs if configs is not None else OTHER_CONFIGS):
        m = _build_model(name, local_rank, B, kw)
        eng = m.engine
        wav = torch.from_numpy(np.tile(base, ((B + 15) // 16, 1))[:B].copy()).cuda()
        out = torch.empty((B, eng.output_samples(CLIP_SAMPLES)), dtype=torch.float32, device=wav.device)
        # untimed warm-up: kernel attributes, lazily grown scratch - and the clocks, which fall back while the host packs the next
        # model's weights: at least three calls and one second of work (round 6: G2Net_new's first timed pass read 47.0 ms against
        # 40.6 for the second after two warm-up calls; with three calls / 0.25 s TaylorSENet - 105 ms a call, the longest weight
        # packing of the zoo in front of it - still read 112.9 then 106.9 ms)
        t_w, n_w = time.perf_counter(), 0
        while n_w < 3 or (time.perf_counter() - t_w < 1.0 and n_w < 400):
            eng.enhance_batch(wav, out)
            torch.cuda.synchronize()
            n_w += 1
        k = max(steps, 20) if B == 1 else steps
        passes = []          # two timed passes of k steps; their mean is reported, both are in the row (a pass of a fresh engine
        for _ in range(2):   # now and then catches a clock dip: G2Net 4 374 vs 5 375 utt/s in one of round 4's runs)
            t0 = time.perf_counter()
            for _ in range(k):
                eng.enhance_batch(wav, out)
            torch.cuda.synchronize()
            passes.append((time.perf_counter() - t0) / k)
        dt = sum(passes) / len(passes)          # the MEAN of the passes is the figure (a min-of-N is a selection)
        assert bool(torch.isfinite(out).all()), name
        ups = B / dt
        rows.append({"config": cfg, "model": name, "batch": B, "steps": k, "utt_s": round(ups, 1), "ms_per_step": round(dt * 1e3, 3),
                     "ms_per_step_passes": [round(v * 1e3, 3) for v in passes], "utt_s_best_pass": round(B / min(passes), 1),
                     "x_realtime": round(ups * CLIP_SECONDS, 0), "gflop_per_utt": gflop,
                     "achieved": round(ups * gflop / 1e3, 2), "frac": round(ups * gflop / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)})
        del m, eng, wav, out
        torch.cuda.empty_cache()
    return rows


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
    ap.add_argument('--no-configs', action='store_true', help="skip BASELINE's other configurations (roofline.configs)")
    ap.add_argument('--no-zoo', action='store_true', help='skip the other eight networks of the zoo (roofline.zoo)')
    ap.add_argument('--cpu-worker', type=str, default='', help='internal: "seed,p_in,p_out,P" - time the C++ CPU restatement, print JSON')
    args = ap.parse_args()
    if args.cpu_worker:
        print(json.dumps(_cpu_worker(args.cpu_worker)), flush=True)
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
    distinct_devices = 1
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank) if args.backend == 'nccl' else None)
        # one rank = one GPU: every rank names the device it sits on (uuid / PCI bus id where torch exposes them); over
        # nccl (= RCCL) the ranks must sit on `world` DISTINCT devices - the line then carries `rccl_ranks` so that a scaling
        # record shows how many ranks the collective library really saw (VERDICT r5 next #7c)
        props = torch.cuda.get_device_properties(local_rank)
        me = (os.uname().nodename, local_rank, str(getattr(props, 'uuid', '')), str(getattr(props, 'pci_bus_id', '')))
        devs = [None] * world
        dist.all_gather_object(devs, me)
        distinct_devices = len(set(devs))
        if args.backend == 'nccl':
            assert distinct_devices == world, f"RCCL group of {world} ranks on {distinct_devices} distinct device(s): {devs}"

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
            "rccl_ranks": world if (use_pg and args.backend == 'nccl') else 0,
            "collective": ({"backend": args.backend, "ranks": world, "distinct_devices": distinct_devices} if use_pg else None),
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
                "bound": "mfma", "kernel": "se::gc_kernel<BM,BN,..> (f32-MFMA tap-table implicit-GEMM conv; the 128- / 256-channel "
                                           "complex layers as three real products + sum / combine passes, counted in the family)",
                "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": pmc['traffic_GB_per_launch'] if pmc and B == 256 and not pmc['stale'] else None,
                "traffic_stale": bool(pmc and pmc['stale']),
                "traffic_unit": "GB of HBM traffic per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 "
                                "--pmc passes of this command at batch 256: %s)" % (pmc['source'] if pmc else 'profiles/'),
                "algorithmic_GB_per_launch": round((rd_b + wr_b) / 1e9 / max(prof['gemm_launches'], 1), 3),
                "launches_per_step": prof['gemm_launches'],
                "algorithmic_gflop_per_step": round(prof['gemm_flops'] / 1e9, 1),
                "algorithmic_gflop_note": "counted on the clips' own frame count (T = 501), not on the 504 frames the engine "
                                          "runs to keep rows whole 16 B groups",
                "algorithmic_flops_note": "SURVEY 8(d) counts four real products per complex product; encoder layers 4-6 and "
                                          "decoder layers 1-2 execute three (Gauss, DESIGN 3.6), so the matrix cores execute "
                                          "fewer flops than `achieved` prices: executed_mfma_tflop_per_step is the PMC count "
                                          "(SQ_INSTS_VALU_MFMA_MOPS_F32 x 512, padded tile rows included)",
                "executed_mfma_tflop_per_step": pmc.get('executed_mfma_tflop_per_step') if pmc and B == 256 and not pmc['stale'] else None,
                # like-for-like with rounds 1-3: the flops the matrix cores EXECUTE over the family's time
                "executed_frac": (round(pmc['executed_mfma_tflop_per_step'] / (prof['gemm_ms'] * 1e-3) / F32_MFMA_PEAK_TFLOPS, 4)
                                  if pmc and B == 256 and not pmc['stale'] and pmc.get('executed_mfma_tflop_per_step') else None),
                "profiler_events_in_timed_region": True,
                "traffic_source_commit": pmc.get('commit') if pmc else None,
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
                     'stft': 'se::stft2_kernel<512> (frame, reflect pad, Hann, register-resident radix-8 FFT, x*c, |X|^p, 128 B runs)',
                     'mask': 'se::dccrn_mask_kernel (E mask + decompress)',
                     'istft': 'se::istft2_kernel<512> (register-resident inverse FFT + overlap-add + /c, frames stay in LDS)'}
            res["roofline_stages"] = [
                {"stage": k, "kernel": names[k], "bound": "hbm", "ms_per_step": round(v['ms'], 4),
                 "algorithmic_GB_per_step": round(v['bytes'] / 1e9, 4),
                 "achieved": round(v['bytes'] / 1e9 / max(v['ms'] * 1e-3, 1e-12), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(v['bytes'] / 1e9 / max(v['ms'] * 1e-3, 1e-12) / HBM_PEAK_GBS, 4),
                 "share_of_step": round(v['ms'] / (1e3 * dt / args.steps), 5)}
                for k, v in stages.items() if v['launches'] > 0]
        if world == 1 and not args.no_configs and "roofline" in res:
            # the headline engine's 24 GB go back first (FullSubNet's shard needs most of the HBM)
            del model, eng, pipe, out, wav
            torch.cuda.empty_cache()
            head = {"config": "configs[2]: DCCRN complex-mask, compressed input, batch %d (the headline above)" % B,
                    "model": "dccrn", "batch": B, "steps": args.steps, "utt_s": round(value, 1),
                    "ms_per_step": round(1e3 * dt / args.steps, 3), "x_realtime": round(value * CLIP_SECONDS, 0),
                    "gflop_per_utt": DCCRN_GFLOP_PER_UTT, "achieved": res["roofline_whole_path"]["achieved"],
                    "frac": res["roofline_whole_path"]["frac"]}
            rows = run_other_configs(torch, local_rank, max(3, min(args.steps, 10)))
            res["roofline"]["configs"] = rows[:2] + [head] + rows[2:]
            if not args.no_zoo:
                res["roofline"]["zoo"] = run_other_configs(torch, local_rank, 5, ZOO_CONFIGS)
            res["roofline"]["configs_note"] = ("whole decode path per config: utt/s x SURVEY 8(d) GFLOP per utterance against the "
                                              "f32 MFMA peak; unprofiled steps timed by the host clock around a device sync; two passes per config, "
                                              "their MEAN reported (utt_s, ms_per_step), both in ms_per_step_passes, the faster one in utt_s_best_pass")
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(seed, p_in, p_out)
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
