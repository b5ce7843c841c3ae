#!/usr/bin/env python
"""Command-line stand-in for the reference's `*_decode_vb.py` / `*_decode.py` scripts (se_amd/decode.py:main): same
argument names (`--mix_file_path`, `--esti_clean_file_path` / `--esti_file_path`, `--fs`, G2Net's `--Model_path`), plus
`--model`.  One process per GPU:

    python tools/decode_vb.py --model dccrn --mix_file_path noisy/ --esti_clean_file_path enh/ --checkpoint m.pth
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/decode_vb.py --model dccrn ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se_amd  # noqa: E402,F401
from se_amd.decode import main  # noqa: E402

if __name__ == '__main__':
    main()
