"""`enhance(args)` drivers: the engine-side counterpart of every reference `*_decode_vb.py`.

Same argparse surface (`--mix_file_path`, `--esti_clean_file_path` or `--esti_file_path`, `--fs`), same iteration
order (`os.listdir`), same output file names and PCM_16 WAV output; the per-utterance arithmetic
(normalise -> STFT -> network -> iSTFT -> de-normalise) runs batched in the HIP engine.  Utterances of equal length
are decoded together (results are batch-invariant: every utterance is an independent sequence); utterances of
different lengths share a call through se_enhance_ragged (each one still gets exactly its batch-1 result).
"""
import argparse
import os

import numpy as np

from . import wavio, resample
from .models import MODEL_CLASSES

# checkpoint name(s) checked in at each reference decode script
DEFAULT_CKPT = {
    'lstm': './BEST_MODEL/vb_lstm_noncprs_model.pth',                       # LSTM/lstm_decode_vb.py:19
    'crn': './BEST_MODEL/vb_crn_noncprs_model.pth',                         # CRN/crn_decode_vb.py:19
    'gcrn': './BEST_MODEL/vb_gcrn_cprs_model.pth',                          # GCRN/gcrn_decode_vb.py:20
    'dpcrn': './BEST_MODEL/vb_dpcrn_noncprs_model.pth',                     # DPCRN/dpcrn_decode_vb.py:20
    'dccrn': './BEST_MODEL/vb_dccrn_noncprs_model.pth',                     # DCCRN/dccrn_decode_vb.py:12
    'fullsubnet': './BEST_MODEL/vb_fullsubnet_noncprs_model_512_256.pth',   # FullSubNet/fullsubnet_sa_decode_vb.py:25
    'ctsnet': ('./BEST_MODEL/step1_vb_cts_noncprs_model_final.pth',         # CTSNet/two_stage_com_decode_vb.py:15-16
               './BEST_MODEL/step2_vb_cts_noncprs_model.pth'),
    'g2net': './BEST_MODEL/vb_gaf_noncprs_model.pth',                       # G2Net_VB/com_decode.py:103 (--Model_path)
    'taylorsenet': './BEST_MODEL/vb_taylor_noncprs_model.pth',              # TaylorSENet/taylorsenet_decode_vb.py:14
    'uformer': './BEST_MODEL/vb_uformer_cprs_model.pth',                    # Uformer/uformer_decode_vb.py:20
    'ctsnet_new': ('./BEST_MODEL/step1_vb_cts_cprs_model_final.pth',        # CTSNet_new/two_stage_com_decode_vb.py:15-16
                   './BEST_MODEL/step2_vb_cts_cprs_model.pth'),
    'taylorsenet_new': './BEST_MODEL/vb_taylor_cprs_model.pth',             # TaylorSENet_new/taylorsenet_decode_vb.py:14
    'g2net_new': './BEST_MODEL/vb_gaf_cprs_model.pth',                      # G2Net_new/com_decode.py:104
}
MODELS = sorted(DEFAULT_CKPT)


def load_checkpoint(path):
    """`torch.load(path)` of a flat state dict, or an .npz with the same keys."""
    if path.endswith('.npz'):
        return dict(np.load(path))
    import torch
    return torch.load(path, map_location='cpu')


def _build(model, checkpoint, state_dict, **kw):
    """Construct the host class of `model` and load its weights (CTSNet: two files / two state dicts)."""
    from . import models, models_new
    if model in ('ctsnet', 'ctsnet_new'):
        net = (models_new if model.endswith('_new') else models).CTSNet(**kw)
        sds = state_dict if state_dict is not None else [load_checkpoint(c) for c in (checkpoint or DEFAULT_CKPT[model])]
        return net.load_state_dicts(*sds)
    net = MODEL_CLASSES[model](**kw)
    sd = state_dict if state_dict is not None else load_checkpoint(checkpoint or DEFAULT_CKPT[model])
    if isinstance(sd, dict) and 'model_state_dict' in sd:        # TaylorSENet/taylorsenet_decode_vb.py:14-15 wraps it
        sd = sd['model_state_dict']
    net.load_state_dict(sd)
    return net


RAGGED_MODELS = frozenset(MODELS)        # every model takes clips of different lengths in one call


def plan_batches(lengths, max_batch, batch_samples, ragged, max_pad=0.15):
    """Group clip indices into engine calls.  ragged: clips sorted by length, consecutive runs of up to `max_batch` clips
    whose padded size (count x longest) stays within `batch_samples` AND whose padding (1 - sum of lengths / padded size)
    stays within `max_pad` - a run is closed as soon as the next, longer clip would push the rows already in it past that
    share of wasted frames (without the cap a run of 256 clips of a VoiceBank+DEMAND-like corpus spans 1.2 - 9.8 s and
    carries 42 % padding).  Not ragged: only clips of exactly equal length share a call."""
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    batches, cur, tot = [], [], 0
    for i in order:
        n = lengths[i]
        same = not cur or lengths[cur[0]] == n
        fits = len(cur) < max_batch and (len(cur) + 1) * n <= max(batch_samples, n)
        tight = (len(cur) + 1) * n * (1.0 - max_pad) <= tot + n            # padding of the run with this clip in it
        if cur and not (fits and ((ragged and tight) or same)):
            batches.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += n
    if cur:
        batches.append(cur)
    return batches


def shard_clips(lengths, rank, world):
    """Clips -> ranks: the length-sorted clip list dealt round-robin (clip k of the sorted order goes to rank k % world), so
    every rank gets the same number of clips (+-1), the same length distribution and therefore the same number of frames
    to decode and the same padding in its calls - a contiguous shard of a length-sorted plan would hand rank 0 all the
    short clips and the last rank all the long ones.  Deterministic: every rank derives its own share from the
    same file list, nothing is exchanged.  -> this rank's clip indices (ascending length)."""
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    return order[rank::world]


def _rank_world(rank, world):
    import torch.distributed as dist
    if rank is not None and world is not None:
        return int(rank), int(world)
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


class _Slot:
    """One stage of the pipeline's ring: pinned host + device staging for a call's input and output.  The buffers are flat;
    a call of nb rows x n samples uses the first nb * n elements as a dense [nb, n] matrix, so every host <-> device
    copy is one contiguous pinned transfer (a strided sub-view makes torch stage the copy through pageable memory and
    block the issuing thread until the decode in front of it has finished)."""

    def __init__(self, torch, rows, n_in, n16, n_out, device, raw16):
        dev = torch.device('cuda', device)
        self.h_in = torch.empty(rows * n_in, dtype=torch.int16 if raw16 else torch.float32).pin_memory()
        self.h_in_np = self.h_in.numpy()
        self.d_in = torch.empty(rows * n_in, dtype=self.h_in.dtype, device=dev)
        self.d_nat = torch.empty(rows * n_in, dtype=torch.float32, device=dev) if raw16 else self.d_in
        self.wav = torch.empty(rows * n16, dtype=torch.float32, device=dev)
        self.out = torch.empty(rows * n_out, dtype=torch.float32, device=dev)
        self.d_q = torch.empty(rows * n_out, dtype=torch.int16, device=dev)
        self.h_q = torch.empty(rows * n_out, dtype=torch.int16).pin_memory()
        self.h_q_np = self.h_q.numpy()
        self.ready = torch.cuda.Event()     # input resident on the device
        self.done = torch.cuda.Event()      # quantised output resident in h_q


def enhance(args, model='dccrn', checkpoint=None, p_in=None, p_out=None, max_batch=64, state_dict=None,
            batch_samples=None, max_pad=0.15, rank=None, world=None, verbose=True, stats=None, readers=4):
    """The file -> file decode of a directory (`enhance(args)` of every `*_decode_vb.py`), as a pipeline:

      plan     lengths come from the WAV headers only; the length-sorted clip list is dealt round-robin to the ranks and every
               rank reads, decodes and writes ONLY its own clips (one process per GPU, no collective on the data path);
               clips of different lengths share calls (se_enhance_ragged: each clip gets exactly its batch-1 result)
               within a padding cap;
      reader   threads copy the raw PCM_16 samples of the next call into pinned memory, a side stream uploads them, turns
               them into floats (se_pcm16_decode) and resamples to 16 kHz where the corpus is not (se_resample,
               librosa.resample(..., 16000) of e.g. DCCRN/dccrn_decode_vb.py:26) - while the current call decodes;
      decode   the caller's stream: se_enhance_ragged / se_enhance_batch, then PCM_16 quantisation on the device
               (se_pcm16_encode) and one asynchronous D2H of 2-byte samples;
      writer   a thread writes each clip's `<out>/<same file name>` as soon as its call's samples have landed.
    p_in / p_out None -> the exponents checked in at the model's decode script (host class defaults).  `stats` (a dict)
    receives the timing of the run (tools/corpus_bench.py)."""
    import queue
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    import ctypes as C
    import torch
    from . import _lib
    t_begin = time.perf_counter()
    mix, out_dir = args.mix_file_path, getattr(args, 'esti_clean_file_path', None) or args.esti_file_path
    if getattr(args, 'noise_type', None):
        # WSJ0-SI84 grid drivers (`*_decode.py`, e.g. CRN/crn_decode.py:28-32): one (noise, seen/unseen, SNR) cell
        mix = os.path.join(mix, args.noise_type, args.seen, str(args.snr))
        out_dir = os.path.join(out_dir, args.noise_type, args.seen, str(args.snr))
    os.makedirs(out_dir, exist_ok=True)
    rank, world = _rank_world(rank, world)
    device = torch.cuda.current_device()
    files = sorted(os.listdir(mix)) if world > 1 else os.listdir(mix)     # ranks must agree on the order
    if not files:
        return 0
    # ---- plan from the headers
    info = [wavio.wav_info(os.path.join(mix, f)) for f in files]
    for f, (_, _, ch, tag, bits, _) in zip(files, info):
        if ch != 1:
            raise ValueError(f'{f}: {ch} channels - the decode scripts read mono clips')
    native = [i[0] for i in info]
    rates = [i[1] for i in info]
    lengths = [n if fs == 16000 else resample.resample_samples(n, fs, 16000) for n, fs in zip(native, rates)]
    raw16 = all(i[3] == 1 and i[4] == 16 for i in info)       # PCM_16 corpus: raw samples to the device, floats made there
    if batch_samples is None:
        batch_samples = max_batch * 64000                     # the padded size of a call of max_batch 4 s clips
    own = shard_clips(lengths, rank, world)
    mine = [[own[k] for k in b] for b in plan_batches([lengths[i] for i in own], max_batch, batch_samples,
                                                       model in RAGGED_MODELS, max_pad)]
    if stats is not None:
        pad, use = sum(len(b) * max(lengths[i] for i in b) for b in mine), sum(lengths[i] for i in own)
        stats.update(files=len(files), files_rank=len(own), calls_rank=len(mine), world=world,
                     pad_frac=round(1.0 - use / max(pad, 1), 4), pad_over_audio=round(pad / max(use, 1) - 1.0, 4),
                     audio_s_rank=round(use / 16000.0, 2))      # pad_frac: the planner's definition (share of padded rows' samples
                                                                # that is padding, capped by max_pad); pad_over_audio: extra work / audio
    if not mine:
        return 0
    # ---- engine: the workspace is sized for the calls this rank actually makes, not for max_batch x the longest clip
    eng_batch = max(len(b) for b in mine)
    eng_len = max(lengths[i] for b in mine for i in b)
    nat_len = max(native[i] for b in mine for i in b)
    net = _build(model, checkpoint, state_dict, device=device, max_batch=eng_batch, max_samples=max(eng_len, 512),
                 p_in=p_in, p_out=p_out)
    eng = net.engine
    lib = _lib.load()
    n_out_max = eng.output_samples(eng_len)
    NS = 3
    slots = [_Slot(torch, eng_batch, nat_len, eng_len, n_out_max, device, raw16) for _ in range(NS)]
    free = queue.Queue()
    for sl in slots:
        free.put(sl)
    staged, finished = queue.Queue(), queue.Queue()
    side = torch.cuda.Stream(device)
    main = torch.cuda.current_stream(device)
    errors = []
    busy = {'read': 0.0, 'stage': 0.0, 'write': 0.0, 'wait_in': 0.0, 'issue': 0.0}      # seconds per stage (stats)
    t_ready = time.perf_counter()

    def _check(rc):
        if rc:
            raise RuntimeError(lib.se_last_error(None).decode())

    def load_row(sl, r, i, nmax):
        path = os.path.join(mix, files[i])
        row = sl.h_in_np[r * nmax:r * nmax + native[i]]
        if raw16:
            fd = os.open(path, os.O_RDONLY)
            try:
                got = os.preadv(fd, [memoryview(row).cast('B')], info[i][5])
            finally:
                os.close(fd)
            if got != 2 * native[i]:
                raise ValueError(f'{path}: short read ({got} of {2 * native[i]} bytes)')
        else:
            row[:] = wavio.read_wav(path)[0]

    def reader():
        try:
            torch.cuda.set_device(device)
            p = lambda t: C.c_void_p(t.data_ptr())
            with ThreadPoolExecutor(max(1, readers)) as pool:
                for b in mine:
                    sl = free.get()
                    if errors:
                        break
                    t0 = time.perf_counter()
                    nb, nmax, lmax = len(b), max(native[i] for i in b), max(lengths[i] for i in b)
                    list(pool.map(lambda ri: load_row(sl, ri[0], ri[1], nmax), enumerate(b)))
                    t1 = time.perf_counter()
                    busy['read'] += t1 - t0
                    with torch.cuda.stream(side):
                        st = C.c_void_p(side.cuda_stream)
                        sl.d_in[:nb * nmax].copy_(sl.h_in[:nb * nmax], non_blocking=True)
                        if raw16:
                            _check(lib.se_pcm16_decode(p(sl.d_in), nmax, nb, nmax, p(sl.d_nat), nmax, st))
                        if all(rates[i] == 16000 for i in b):
                            wav = sl.d_nat                     # already at 16 kHz: rows of pitch nmax = lmax
                        else:
                            wav = sl.wav
                            esz = 4
                            for r, i in enumerate(b):
                                src = C.c_void_p(sl.d_nat.data_ptr() + esz * r * nmax)
                                dst = C.c_void_p(sl.wav.data_ptr() + esz * r * lmax)
                                if rates[i] == 16000:     # librosa.resample returns its input when the rates agree
                                    sl.wav[r * lmax:r * lmax + native[i]].copy_(sl.d_nat[r * nmax:r * nmax + native[i]],
                                                                                non_blocking=True)
                                else:
                                    # librosa.resample(feat_wav, orig_fs, 16000, fix=True, scale=False), dccrn_decode_vb.py:26
                                    _check(lib.se_resample(src, native[i], 1, native[i], rates[i], 16000, dst, lengths[i], st))
                        sl.ready.record(side)
                    busy['stage'] += time.perf_counter() - t1
                    staged.put((b, sl, wav))
        except Exception as ex:                    # surface reader failures in the caller's thread
            errors.append(ex)
        finally:
            staged.put(None)

    cnt = [0]

    def writer():
        try:
            while True:
                item = finished.get()
                if item is None:
                    return
                b, sl, n_out = item
                sl.done.synchronize()
                t0 = time.perf_counter()
                for r, i in enumerate(b):
                    n = eng.output_samples(lengths[i])
                    with open(os.path.join(out_dir, files[i]), 'wb', buffering=0) as f:
                        f.write(wavio.wav_header_pcm16(2 * n, args.fs) + sl.h_q_np[r * n_out:r * n_out + n].tobytes())
                    cnt[0] += 1
                    if verbose:
                        print(' The %d utterance has been decoded!' % cnt[0])
                busy['write'] += time.perf_counter() - t0
                free.put(sl)
        except Exception as ex:
            errors.append(ex)
            while True:                            # keep the ring moving so the other stages can finish
                item = finished.get()
                if item is None:
                    return
                free.put(item[1])

    tr, tw = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
    tr.start()
    tw.start()
    # the decode loop may raise (an EngineError, a refused shape): the sentinels below must go out whatever happens, or the
    # writer would block on `finished` and the reader on `free` for ever with pinned / device slots in their hands
    aborted = False
    try:
        while True:
            t0 = time.perf_counter()
            item = staged.get()
            if item is None or errors:
                break
            t1 = time.perf_counter()
            busy['wait_in'] += t1 - t0
            b, sl, wav_flat = item
            lens = [lengths[i] for i in b]
            nb, lmax = len(b), max(lens)
            main.wait_event(sl.ready)
            n_out = eng.output_samples(lmax)
            wav = wav_flat[:nb * lmax].view(nb, lmax)
            out = sl.out[:nb * n_out].view(nb, n_out)
            if min(lens) == lmax:
                eng.enhance_batch(wav, out)
            else:
                eng.enhance_ragged(wav, lens, out)
            _check(lib.se_pcm16_encode(C.c_void_p(out.data_ptr()), n_out, nb, n_out, C.c_void_p(sl.d_q.data_ptr()), n_out,
                                       C.c_void_p(main.cuda_stream)))
            sl.h_q[:nb * n_out].copy_(sl.d_q[:nb * n_out], non_blocking=True)
            sl.done.record(main)
            busy['issue'] += time.perf_counter() - t1
            finished.put((b, sl, n_out))
    except BaseException:
        aborted = True
        raise
    finally:
        finished.put(None)
        if aborted or errors:
            # un-block a reader waiting for a slot and drop what it staged: it stops at its next `free.get()` / end of list
            if aborted and not errors:
                errors.append(RuntimeError('decode aborted'))       # tells the reader to stop
            for sl in slots:
                free.put(sl)
        tw.join(timeout=30.0)
        tr.join(timeout=30.0 if not (aborted or errors) else 5.0)
    if errors and not aborted:
        raise errors[0]
    if stats is not None:
        t_end = time.perf_counter()
        stats.update(decoded=cnt[0], setup_s=round(t_ready - t_begin, 3), pipeline_s=round(t_end - t_ready, 3),
                     total_s=round(t_end - t_begin, 3), clips_per_s=round(cnt[0] / max(t_end - t_ready, 1e-9), 1),
                     raw_pcm16=bool(raw16), stage_busy_s={k: round(v, 3) for k, v in busy.items()})
    return cnt[0]


def main():
    parser = argparse.ArgumentParser('Recovering audio')
    parser.add_argument('--mix_file_path', type=str, required=True)
    parser.add_argument('--esti_clean_file_path', '--esti_file_path', dest='esti_clean_file_path', type=str, required=True)
    parser.add_argument('--fs', type=int, default=16000)
    parser.add_argument('--model', type=str, default='dccrn', choices=MODELS)
    parser.add_argument('--checkpoint', '--Model_path', dest='checkpoint', type=str, nargs='+', default=None,
                        help='state dict file (two for ctsnet); --Model_path is G2Net\'s name for it (G2Net_VB/com_decode.py:103)')
    parser.add_argument('--max_batch', type=int, default=64, help='most clips per engine call')
    parser.add_argument('--cprs', action='store_true', help='compressed-spectrum variant: exponents 0.5 / 2.0')
    parser.add_argument('--noncprs', action='store_true', help='uncompressed variant: exponents 1.0 / 1.0')
    # WSJ0-SI84 grid drivers (`*_decode.py`): decode <mix>/<noise_type>/<seen>/<snr>/
    parser.add_argument('--noise_type', type=str, default=None)
    parser.add_argument('--seen', type=str, default='unseen')
    parser.add_argument('--snr', type=str, default='-5')
    args = parser.parse_args()
    p_in, p_out = (0.5, 2.0) if args.cprs else ((1.0, 1.0) if args.noncprs else (None, None))
    ck = args.checkpoint
    if ck is not None and not args.model.startswith('ctsnet'):
        ck = ck[0]
    # one process per GPU: under a launcher (torch.distributed.run) the file list is dealt to the ranks by RANK / WORLD_SIZE;
    # no process group is needed - there is no collective on this path, every rank writes its own output files
    import torch
    if 'LOCAL_RANK' in os.environ:
        torch.cuda.set_device(int(os.environ['LOCAL_RANK']) % max(torch.cuda.device_count(), 1))
    enhance(args, args.model, ck, p_in, p_out, max_batch=args.max_batch)


if __name__ == '__main__':
    main()
