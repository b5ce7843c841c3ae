"""Adopt the upstream `complexnn.py` (huyanxin/DeepComplexCRN) as the pin of DCCRN's operators.

TEST INFRASTRUCTURE (build container only: imports /root/reference).

`DCCRN/DCCRN_cprs.py:6` imports ComplexConv2d, ComplexConvTranspose2d, NavieComplexLSTM, complex_cat and ComplexBatchNorm
from a third-party `complexnn.py` that is absent from the reference and unversioned (used at :66-72, :84-90, :108-115,
:182, :197).  Until that file is supplied the fixtures of DCCRN are generated on top of oracle/_complexnn_recall.py, a
restatement, and DCCRN parity is "unpinned at the complexnn boundary".  This script is the one-command adoption:

    python -m oracle.adopt_complexnn /path/to/complexnn.py            # report only
    python -m oracle.adopt_complexnn /path/to/complexnn.py --write    # + regenerate tests/golden/dccrn*.npz from it

It imports the supplied file in place of the recall (nothing else changes: the reference's own DCCRN class runs on top of
it), and reports
  1. the state-dict key schema of the reference's DCCRN built on the supplied operators against the schema the engine's
     loader expects (se_amd.schemas.dccrn_schema) - key names come from complexnn.py, so a difference here means the
     loader needs a key mapping;
  2. for every combination of the two conventions DCCRN_cprs.py itself does not determine (SE_CFG_DCCRN_BIAS_PER_PART,
     SE_CFG_DCCRN_PLAIN_CAT - include/se_engine.h), whether the numpy oracle (oracle/models.py:dccrn_forward(variant=))
     reproduces the reference's forward and decode with the supplied operators: the matching combination is the value of
     `se_config.flags` (and of `variant`) that makes the engine follow the real file;
  3. whether the supplied file and the recall give the same output (i.e. whether the committed fixtures change at all).
With --write the DCCRN fixtures (dccrn.npz, schema_dccrn.json, long10/15_dccrn.npz) are regenerated from the supplied
file and a marker tests/golden/dccrn_pin.json records its sha256 - tests/test_oracle_golden.py then reports DCCRN as pinned.
Exit status 0 = some flag combination matches, 1 = none does (the operators differ beyond the two flags: read the report).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_golden as GG  # noqa: E402
from oracle import models as M  # noqa: E402
from oracle import decode as D  # noqa: E402
import se_amd  # noqa: E402,F401
from se_amd import synth, schemas  # noqa: E402

FLAG_NAMES = {0: '0 (two-real-conv bias combination, real/imag-wise complex_cat)', 2: 'SE_CFG_DCCRN_BIAS_PER_PART',
              4: 'SE_CFG_DCCRN_PLAIN_CAT', 6: 'SE_CFG_DCCRN_BIAS_PER_PART | SE_CFG_DCCRN_PLAIN_CAT'}


def _rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))


def _build(path):
    """The reference's DCCRN (decode script's constructor, dccrn_decode_vb.py:11) on the operators of `path`."""
    GG.COMPLEXNN_PATH = path
    sys.modules.pop('complexnn', None)
    sys.modules.pop('DCCRN_cprs', None)
    mod = GG.import_ref('DCCRN', 'DCCRN_cprs')
    return mod.DCCRN(rnn_units=256, masking_mode='E', use_clstm=True, kernel_num=[32, 64, 128, 256, 256, 256])


def adopt(path, write=False, gold=None, out=print):
    path = None if path is None else os.path.abspath(path)
    model = _build(path)
    report = {'file': path or 'oracle/_complexnn_recall.py',
              'sha256': hashlib.sha256(open(path or os.path.join(HERE, '_complexnn_recall.py'), 'rb').read()).hexdigest()}
    # 1. key schema
    got = synth.schema_of(model.state_dict())
    want = schemas.dccrn_schema()
    missing = [k for k in want if k not in got]
    extra = [k for k in got if k not in want]
    shape_diff = [k for k in want if k in got and tuple(got[k][0]) != tuple(want[k][0])]
    report['schema'] = {'keys': len(got), 'missing_from_supplied': missing, 'unknown_to_engine': extra, 'shape_differs': shape_diff}
    out(f"[1] state dict: {len(got)} keys; engine expects {len(want)}; missing {len(missing)}, unknown {len(extra)}, "
        f"mis-shaped {len(shape_diff)}")
    for k in (missing[:5] + extra[:5] + shape_diff[:5]):
        out(f"      {k}")
    if missing or extra or shape_diff:
        out("    -> the loader's key schema (se_amd/schemas.py:dccrn_schema, csrc/model_dccrn.hip finalize) needs a mapping")
        report['flags'] = None
        return report, 1
    # 2. which convention flags reproduce it
    sd = synth.synth_state_dict(got, 14)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    x = np.random.default_rng(8).standard_normal((2, 2, 257, 7)).astype(np.float32)
    wav = synth.synth_clip(6, 'speech', 4000)
    try:
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()
        enh_c = GG._enhance_dccrn(model, wav, 0.5, 2.0)[0]
    except Exception as ex:          # e.g. non-causal time padding breaks DCCRN_cprs.py:197-199's shape algebra
        out(f"[2] the reference's DCCRN.forward does not run on the supplied operators: {type(ex).__name__}: {ex}")
        report['flags'] = []
        report['forward_error'] = f'{type(ex).__name__}: {ex}'
        return report, 1
    match = []
    for v in (0, 2, 4, 6):
        yo = M.dccrn_forward(sd, x, variant=v)
        e = _rms(yo - y) / max(_rms(y), 1e-12)
        ok = e < 1e-5
        out(f"[2] flags = {v}: oracle forward vs reference-on-supplied-operators: relative rms err {e:.2e}  "
            f"{'MATCH' if ok else 'differs'}   ({FLAG_NAMES[v]})")
        if ok:
            match.append(v)
    report['flags'] = match
    if len(match) == 1 and match[0] == 0:
        eo = _rms(D.enhance_dccrn(sd, wav, 0.5, 2.0) - enh_c)
        out(f"    decode of a 4 000-sample clip, oracle vs reference: rms err {eo:.2e}")
        report['decode_rms_err'] = eo
    # 3. does anything change against the committed fixtures?
    gdir = gold or GG.GOLD
    fx = os.path.join(GG.GOLD, 'dccrn.npz')
    if os.path.exists(fx):
        G = np.load(fx)
        same = _rms(G['y'] - y) < 1e-6 * max(_rms(y), 1e-12) and _rms(G['enh_cprs'] - enh_c) < 1e-6 * max(_rms(enh_c), 1e-12)
        report['fixtures_unchanged'] = bool(same)
        out(f"[3] committed tests/golden/dccrn.npz {'is reproduced by' if same else 'DIFFERS from'} the supplied operators")
    if match:
        out(f"==> engine configuration that follows the supplied file: se_config.flags |= {match[0]}"
            + ("  (the default)" if match[0] == 0 else f"  = {FLAG_NAMES[match[0]]}"))
    else:
        out("==> no combination of the two convention flags reproduces the supplied operators: they differ from the recall in "
            "more than bias combination / concat order (time padding side, LSTM cross terms, projection) - diff the file "
            "against oracle/_complexnn_recall.py")
    if write:
        old = GG.GOLD
        GG.GOLD = gdir
        try:
            GG.FULL_ONLY = GG.LONG_ONLY = False
            GG.gen_dccrn()
            GG.FULL_ONLY = GG.LONG_ONLY = True
            GG.gen_dccrn()
        finally:
            GG.FULL_ONLY = GG.LONG_ONLY = False
            GG.GOLD = old
        with open(os.path.join(gdir, 'dccrn_pin.json'), 'w') as f:
            json.dump({'complexnn': report['file'] if path else 'recall', 'sha256': report['sha256'], 'flags': match,
                       'pinned': path is not None}, f, indent=1)
        out(f"wrote DCCRN fixtures + dccrn_pin.json under {gdir}")
    return report, (0 if match else 1)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    if len(args) != 1:
        print(__doc__)
        sys.exit(2)
    torch.set_num_threads(8)
    _, rc = adopt(args[0], write='--write' in sys.argv)
    sys.exit(rc)
