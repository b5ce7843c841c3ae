"""CPU oracle for the STFT -> network -> iSTFT decode path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain-numpy restatement of the reference's algorithm for the hot path
(`*/..._decode_vb.py` loops and the `nn.Module.forward` of each model), written
from the reference's behaviour with every function citing the reference
file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker.  The
product path (`se_amd`, the HIP engine behind `include/se_engine.h`) never
imports, links or calls anything in here, and fails loudly when the HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * front/back end (STFT / iSTFT / compress / polar)  - pinned against
    `torch.stft` / `torch.istft` run in the build container (fixtures in
    tests/golden, generator oracle/gen_golden.py).
  * LSTM, CRN, DPCRN (incl. the real `vb_dpcrn_noncprs` checkpoint) -
    pinned against the reference `nn.Module`s imported from /root/reference in
    the build container (fixtures + generator committed).
  * DCCRN - the reference's own `DCCRN_cprs.py` is imported, but its operator
    library `complexnn.py` (third-party, huyanxin/DeepComplexCRN, no version
    pinned by the reference, absent from /root/reference) is restated from
    the published upstream in `oracle/_complexnn_recall.py`.  PARITY UNPINNED
    at the complexnn boundary; pinned above it.
  * librosa.stft/istft (un-vendored, unversioned third party) are restated by
    the torch.stft convention; PARITY UNPINNED at the librosa boundary.
"""
