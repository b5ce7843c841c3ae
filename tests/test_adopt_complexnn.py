"""CPU (build container only - needs /root/reference): oracle/adopt_complexnn.py, the one-command adoption of an upstream
complexnn.py as DCCRN's pin (DCCRN/DCCRN_cprs.py:6, :66-72, :84-90, :108-115, :182, :197 are the only touch points).
Exercised with oracle/_complexnn_recall.py as the stand-in input, and with two mutated copies of it whose conventions
differ exactly by the two engine flags - the script must name the flag combination that follows each file."""
import os

import pytest

REF = '/root/reference/DCCRN/DCCRN_cprs.py'
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason='needs the reference checkout (build container)')
HERE = os.path.dirname(os.path.abspath(__file__))
RECALL = os.path.join(os.path.dirname(HERE), 'oracle', '_complexnn_recall.py')


def _adopt(path):
    from oracle import adopt_complexnn as A
    from oracle import gen_golden as GG
    lines = []
    try:
        return A.adopt(path, write=False, out=lines.append) + (lines,)
    finally:
        GG.COMPLEXNN_PATH = None


def test_recall_as_stand_in_matches_the_default_flags():
    rep, rc, lines = _adopt(RECALL)
    assert rc == 0 and rep['flags'] == [0], lines
    assert rep['schema']['keys'] == 134 and not rep['schema']['missing_from_supplied']
    assert rep['fixtures_unchanged'] is True                 # committed dccrn.npz was generated from this very file
    assert rep['decode_rms_err'] < 1e-6


def test_mutated_conventions_are_identified(tmp_path):
    src = open(RECALL).read()
    # (a) complex_cat as a plain channel concat  ->  SE_CFG_DCCRN_PLAIN_CAT
    a = src.replace("return torch.cat([torch.cat(real, axis), torch.cat(imag, axis)], axis)", "return torch.cat(inputs, axis)")
    assert a != src
    pa = tmp_path / 'complexnn_plain_cat.py'
    pa.write_text(a)
    rep, rc, lines = _adopt(str(pa))
    assert rc == 0 and rep['flags'] == [4], lines
    assert rep['fixtures_unchanged'] is False
    # (b) each part adds only its own conv's bias  ->  SE_CFG_DCCRN_BIAS_PER_PART
    b = src.replace("real_out = self.real_conv(real) - self.imag_conv(imag)\n        imag_out = self.imag_conv(real) + self.real_conv(imag)",
                    "real_out = self.real_conv(real) - (self.imag_conv(imag) - self.imag_conv.bias.view(1, -1, 1, 1))\n"
                    "        imag_out = self.imag_conv(real) + (self.real_conv(imag) - self.real_conv.bias.view(1, -1, 1, 1))")
    assert b.count('.bias.view') == 4
    pb = tmp_path / 'complexnn_bias_per_part.py'
    pb.write_text(b)
    rep, rc, lines = _adopt(str(pb))
    assert rc == 0 and rep['flags'] == [2], lines
    # (c) operators that differ beyond the two flags (sign of an LSTM cross term) -> no match, exit status 1
    c = src.replace("real_out = r2r - i2i", "real_out = r2r + i2i")
    assert c != src
    pc = tmp_path / 'complexnn_lstm_sign.py'
    pc.write_text(c)
    rep, rc, lines = _adopt(str(pc))
    assert rc == 1 and rep['flags'] == [], lines
    # (d) operators the reference's forward cannot even run on (symmetric time padding: DCCRN_cprs.py:197 cat fails)
    d = src.replace("if self.padding[1] != 0 and self.causal:", "if False:")
    pd = tmp_path / 'complexnn_noncausal.py'
    pd.write_text(d)
    rep, rc, lines = _adopt(str(pd))
    assert rc == 1 and rep['flags'] == [] and 'forward_error' in rep, lines
