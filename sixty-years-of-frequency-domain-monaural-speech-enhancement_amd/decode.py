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


def plan_batches(lengths, max_batch, batch_samples, ragged):
    """Group clip indices into engine calls.  ragged: clips sorted by length, consecutive runs of up to `max_batch` clips
    whose padded size (count x longest) stays within `batch_samples` - the padding a call carries is the spread of
    lengths inside one run, a few percent on a corpus like VoiceBank+DEMAND (824 clips, ~700 distinct lengths).
    Not ragged: only clips of exactly equal length share a call, as in round 1."""
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    batches, cur = [], []
    for i in order:
        same = not cur or lengths[cur[0]] == lengths[i]
        fits = len(cur) < max_batch and (len(cur) + 1) * lengths[i] <= max(batch_samples, lengths[i])
        if cur and not (fits and (ragged or same)):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def enhance(args, model='dccrn', checkpoint=None, p_in=None, p_out=None, max_batch=64, state_dict=None,
            batch_samples=64 * 64000):
    """p_in / p_out None -> the exponents checked in at the model's decode script (host class defaults).
    Clips of different lengths are decoded together through se_enhance_ragged (each clip gets exactly its batch-1
    result); under torch.distributed the list of engine calls is sharded contiguously over the ranks (one GPU each),
    every rank writing its own output files - the reference's `for file_id in file_list` split across GPUs."""
    import torch
    import torch.distributed as dist
    from . import shard
    mix, out_dir = args.mix_file_path, getattr(args, 'esti_clean_file_path', None) or args.esti_file_path
    if getattr(args, 'noise_type', None):
        # WSJ0-SI84 grid drivers (`*_decode.py`, e.g. CRN/crn_decode.py:28-32): one (noise, seen/unseen, SNR) cell
        mix = os.path.join(mix, args.noise_type, args.seen, str(args.snr))
        out_dir = os.path.join(out_dir, args.noise_type, args.seen, str(args.snr))
    os.makedirs(out_dir, exist_ok=True)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    device = torch.cuda.current_device()
    files = sorted(os.listdir(mix)) if world > 1 else os.listdir(mix)     # ranks must agree on the order
    clips = []
    for name in files:
        x, fs = wavio.read_wav(os.path.join(mix, name))
        if fs != 16000:
            # librosa.resample(feat_wav, orig_fs, 16000, fix=True, scale=False), e.g. DCCRN/dccrn_decode_vb.py:26
            x = resample.resample(torch.from_numpy(x.astype(np.float32)).cuda(), fs, 16000).cpu().numpy()
        clips.append((name, x.astype(np.float32)))
    if not clips:
        return 0
    lengths = [len(x) for _, x in clips]
    ragged = model in RAGGED_MODELS
    batches = plan_batches(lengths, max_batch, batch_samples, ragged)
    lo, hi = shard.shard_range(len(batches), rank, world)
    batches = batches[lo:hi]
    if not batches:
        return 0
    # the workspace is sized for the calls this rank actually makes, not for max_batch x the longest clip
    eng_batch = max(len(b) for b in batches)
    eng_len = max(lengths[i] for b in batches for i in b)
    net = _build(model, checkpoint, state_dict, device=device, max_batch=eng_batch, max_samples=max(eng_len, 512),
                 p_in=p_in, p_out=p_out)
    cnt = 0
    for b in batches:
        lens = [lengths[i] for i in b]
        wav = np.zeros((len(b), max(lens)), np.float32)
        for r, i in enumerate(b):
            wav[r, :lens[r]] = clips[i][1]
        wt = torch.from_numpy(wav).cuda()
        y = (net.enhance_batch(wt) if min(lens) == max(lens) else net.enhance_ragged(wt, lens)).cpu().numpy()
        for r, i in enumerate(b):
            n = net.engine.output_samples(lens[r])
            wavio.write_wav_pcm16(os.path.join(out_dir, clips[i][0]), y[r, :n], args.fs)
            cnt += 1
            print(' The %d utterance has been decoded!' % cnt)
    return cnt


def main():
    parser = argparse.ArgumentParser('Recovering audio')
    parser.add_argument('--mix_file_path', type=str, required=True)
    parser.add_argument('--esti_clean_file_path', '--esti_file_path', dest='esti_clean_file_path', type=str, required=True)
    parser.add_argument('--fs', type=int, default=16000)
    parser.add_argument('--model', type=str, default='dccrn', choices=MODELS)
    parser.add_argument('--checkpoint', type=str, nargs='+', default=None, help='state dict file (two for ctsnet)')
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
    enhance(args, args.model, ck, p_in, p_out)


if __name__ == '__main__':
    main()
