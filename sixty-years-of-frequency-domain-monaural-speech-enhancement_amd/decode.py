"""`enhance(args)` drivers: the engine-side counterpart of every reference `*_decode_vb.py`.

Same argparse surface (`--mix_file_path`, `--esti_clean_file_path` or `--esti_file_path`, `--fs`), same iteration
order (`os.listdir`), same output file names and PCM_16 WAV output; the per-utterance arithmetic
(normalise -> STFT -> network -> iSTFT -> de-normalise) runs batched in the HIP engine.  Utterances of equal length
are decoded together (results are batch-invariant: every utterance is an independent sequence).
"""
import argparse
import os

import numpy as np

from . import wavio
from .models import MODEL_CLASSES

# checkpoint name checked in at each reference decode script
DEFAULT_CKPT = {
    'lstm': './BEST_MODEL/vb_lstm_noncprs_model.pth',       # LSTM/lstm_decode_vb.py:19
    'crn': './BEST_MODEL/vb_crn_noncprs_model.pth',         # CRN/crn_decode_vb.py:19
    'dpcrn': './BEST_MODEL/vb_dpcrn_noncprs_model.pth',     # DPCRN/dpcrn_decode_vb.py:20
    'dccrn': './BEST_MODEL/vb_dccrn_noncprs_model.pth',     # DCCRN/dccrn_decode_vb.py:12
}


def load_checkpoint(path):
    """`torch.load(path)` of a flat state dict, or an .npz with the same keys."""
    if path.endswith('.npz'):
        return dict(np.load(path))
    import torch
    return torch.load(path, map_location='cpu')


def enhance(args, model='dccrn', checkpoint=None, p_in=1.0, p_out=1.0, max_batch=64, state_dict=None):
    import torch
    mix, out_dir = args.mix_file_path, getattr(args, 'esti_clean_file_path', None) or args.esti_file_path
    os.makedirs(out_dir, exist_ok=True)
    files = os.listdir(mix)
    clips = {}
    for name in files:
        x, fs = wavio.read_wav(os.path.join(mix, name))
        if fs != 16000:
            raise NotImplementedError(f'{name}: {fs} Hz input needs the resampler (librosa.resample in the reference, '
                                      'SURVEY 8(f) rank 1: next)')
        clips.setdefault(len(x), []).append((name, x.astype(np.float32)))
    max_len = max(clips) if clips else 0
    net = MODEL_CLASSES[model](max_batch=max_batch, max_samples=max(max_len, 512), p_in=p_in, p_out=p_out)
    net.load_state_dict(state_dict if state_dict is not None else load_checkpoint(checkpoint or DEFAULT_CKPT[model]))
    net.eval()
    cnt = 0
    for length, items in clips.items():
        for i in range(0, len(items), max_batch):
            chunk = items[i:i + max_batch]
            wav = torch.from_numpy(np.stack([x for _, x in chunk])).cuda()
            y = net.enhance_batch(wav).cpu().numpy()
            for (name, _), yy in zip(chunk, y):
                wavio.write_wav_pcm16(os.path.join(out_dir, name), yy, args.fs)
                cnt += 1
                print(' The %d utterance has been decoded!' % cnt)
    return cnt


def main():
    parser = argparse.ArgumentParser('Recovering audio')
    parser.add_argument('--mix_file_path', type=str, required=True)
    parser.add_argument('--esti_clean_file_path', '--esti_file_path', dest='esti_clean_file_path', type=str, required=True)
    parser.add_argument('--fs', type=int, default=16000)
    parser.add_argument('--model', type=str, default='dccrn', choices=sorted(MODEL_CLASSES))
    parser.add_argument('--checkpoint', type=str, default=None)
    parser.add_argument('--cprs', action='store_true', help='compressed-spectrum variant: exponents 0.5 / 2.0')
    args = parser.parse_args()
    p_in, p_out = (0.5, 2.0) if args.cprs else (1.0, 1.0)
    enhance(args, args.model, args.checkpoint, p_in, p_out)


if __name__ == '__main__':
    main()
