#!/usr/bin/env python
"""Instruction counts of the kernels of one csrc file (hipcc -S for gfx950): the footprint DESIGN.md 3.1 tracks.

  python tools/code_size.py gemmconv.hip [min_instructions]  ->  one line per kernel: instructions, v_mfma, scratch ops
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    csrc = [d for d in os.listdir(ROOT) if d.endswith('_amd')][0]
    path = os.path.join(ROOT, csrc, 'csrc', src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-S', '--offload-arch=gfx950', '--cuda-device-only', '-O3', '-std=c++17', path, '-o', out],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for m in re.finditer(r'^(_Z\S+):\s*;\s*@', txt, re.M):
        j = txt.index('s_endpgm', m.end())
        body = [l for l in txt[m.end():j].split('\n') if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
        rows.append((len(body), sum('v_mfma' in l for l in body), sum('scratch_' in l for l in body), name))
    for n, mf, sc, name in sorted(rows, reverse=True):
        if n >= lo:
            print('%7d instructions  %4d mfma  %3d scratch ops  %s' % (n, mf, sc, name.replace('void ', '')))


if __name__ == '__main__':
    main()
