"""Minimal RIFF/WAVE reader and writer (numpy only).

The reference reads with `soundfile.read` (float64 in [-1, 1)) and writes with `soundfile.write(path, y, fs)`, whose
default subtype for .wav is PCM_16 (e.g. DCCRN/dccrn_decode_vb.py:25,64).  soundfile is not available here, so the
decode driver carries its own I/O: PCM 16/24/32-bit and IEEE float 32/64 in, PCM_16 out (libsndfile's 0x7FFF scale, round-to-nearest-even, clipped).
"""
import os
import struct

import numpy as np


def read_wav(path):
    """-> (float64 array [n] or [n, ch], sample_rate)."""
    with open(path, 'rb') as f:
        data = f.read()
    if data[:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError(f'{path}: not a RIFF/WAVE file')
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack('<I', data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            tag, ch, fs, _, _, bits = struct.unpack('<HHIIHH', body[:16])
            if tag == 0xFFFE and len(body) >= 26:      # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                tag = struct.unpack('<H', body[24:26])[0]
            fmt = (tag, ch, fs, bits)
        elif cid == b'data':
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f'{path}: missing fmt/data chunk')
    tag, ch, fs, bits = fmt
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(raw, dtype='<i2').astype(np.float64) / 32768.0
        elif bits == 32:
            x = np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float64) / 8388608.0
        elif bits == 8:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
        else:
            raise ValueError(f'{path}: unsupported PCM width {bits}')
    elif tag == 3:
        x = np.frombuffer(raw, dtype='<f4' if bits == 32 else '<f8').astype(np.float64)
    else:
        raise ValueError(f'{path}: unsupported WAVE format tag {tag}')
    if ch > 1:
        x = x.reshape(-1, ch)
    return x, fs


def wav_info(path):
    """-> (frames, sample_rate, channels, format tag, bits, byte offset of the samples) from the header only: the decode
    driver plans its engine calls from the clip lengths before it reads any audio."""
    with open(path, 'rb') as f:
        head = f.read(12)
        if head[:4] != b'RIFF' or head[8:12] != b'WAVE':
            raise ValueError(f'{path}: not a RIFF/WAVE file')
        fmt = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError(f'{path}: missing fmt/data chunk')
            cid, size = hdr[:4], struct.unpack('<I', hdr[4:])[0]
            if cid == b'fmt ':
                body = f.read(size + (size & 1))
                tag, ch, fs, _, _, bits = struct.unpack('<HHIIHH', body[:16])
                if tag == 0xFFFE and len(body) >= 26:
                    tag = struct.unpack('<H', body[24:26])[0]
                fmt = (tag, ch, fs, bits)
            elif cid == b'data':
                if fmt is None:
                    raise ValueError(f'{path}: data chunk before fmt chunk')
                tag, ch, fs, bits = fmt
                # a streamed / truncated file may announce more than it holds (0xFFFFFFFF, or a size written before the
                # recording ended): the plan is made from what is really there
                off = f.tell()
                size = max(0, min(size, os.fstat(f.fileno()).st_size - off))
                return size // (ch * (bits // 8)), fs, ch, tag, bits, off
            else:
                f.seek(size + (size & 1), 1)


def read_pcm16_into(path, offset, frames, dst):
    """Raw little-endian PCM_16 mono samples of `path` straight into dst[:frames] (an int16 view of a pinned staging row):
    no float conversion on the host - x / 32768 is exact in float32 and is done on the GPU."""
    with open(path, 'rb') as f:
        f.seek(offset)
        n = f.readinto(memoryview(dst[:frames]).cast('B'))
    if n != 2 * frames:
        raise ValueError(f'{path}: short read ({n} of {2 * frames} bytes)')


def pcm16_bytes(y):
    """float samples -> PCM_16 little-endian array as soundfile.write's default subtype makes them: libsndfile scales a
    normalised float by 0x7FFF and rounds to nearest even (pcm.c f2s_array; restated from the published source - soundfile is
    not in this image, unpinned).  Out-of-range samples are clipped here, where libsndfile without SFC_SET_CLIPPING wraps."""
    return np.clip(np.rint(np.asarray(y, dtype=np.float64) * 32767.0), -32768, 32767).astype('<i2')


def wav_header_pcm16(n_bytes, fs, ch=1):
    return b'RIFF' + struct.pack('<I', 36 + n_bytes) + b'WAVE' + b'fmt ' + struct.pack(
        '<IHHIIHH', 16, 1, ch, fs, fs * ch * 2, ch * 2, 16) + b'data' + struct.pack('<I', n_bytes)


def write_wav_pcm16(path, y, fs):
    """soundfile.write(path, y, fs) default for .wav: PCM_16 (see pcm16_bytes for the scale and the clipping)."""
    q = pcm16_bytes(y)
    ch = 1 if q.ndim == 1 else q.shape[1]
    raw = q.tobytes()
    hdr = b'RIFF' + struct.pack('<I', 36 + len(raw)) + b'WAVE' + b'fmt ' + struct.pack(
        '<IHHIIHH', 16, 1, ch, fs, fs * ch * 2, ch * 2, 16) + b'data' + struct.pack('<I', len(raw))
    with open(path, 'wb') as f:
        f.write(hdr + raw)
