"""Test helper: an independent PNG encoder/decoder (numpy + zlib) for checking chan_vese_amd/host/png_io.hpp.
Encoder: any colour type / bit depth, per-row filter types 0-4.  Decoder: 8-bit gray/RGB, any filters."""
import struct
import zlib

import numpy as np


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def encode(rows_bytes, w, h, depth, ctype, filters=None, palette=None, idat_split=1):
    """rows_bytes: list of h bytes objects (packed scanlines, big-endian for 16 bit)."""
    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, samples * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows_bytes[0]))
    for y, row in enumerate(rows_bytes):
        ft = 0 if filters is None else filters[y % len(filters)]
        out = bytearray(len(row))
        for x in range(len(row)):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][ft]
            out[x] = (row[x] - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = row
    comp = zlib.compress(bytes(raw), 6)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        png += _chunk(b"PLTE", bytes(np.asarray(palette, dtype=np.uint8).ravel()))
    step = (len(comp) + idat_split - 1) // idat_split
    for k in range(0, len(comp), step):
        png += _chunk(b"IDAT", comp[k:k + step])
    return png + _chunk(b"tEXt", b"Comment\x00test") + _chunk(b"IEND", b"")


def pack_samples(arr, depth):
    """arr: (h, w*samples) integer samples -> list of packed scanlines."""
    arr = np.asarray(arr)
    rows = []
    for r in arr:
        if depth == 8:
            rows.append(bytes(r.astype(np.uint8)))
        elif depth == 16:
            rows.append(r.astype(">u2").tobytes())
        else:
            bits = np.zeros(((len(r) * depth + 7) // 8) * 8, dtype=np.uint8)
            for k in range(depth):
                bits[k:len(r) * depth:depth] = (r >> (depth - 1 - k)) & 1
            rows.append(bytes(np.packbits(bits)))
    return rows


def decode8(data):
    """8-bit gray (ctype 0) or RGB (ctype 2) PNG -> ndarray (h, w) or (h, w, 3)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        d = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + d) & 0xFFFFFFFF == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", d)
        elif t == b"IDAT":
            idat += d
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert depth == 8 and ctype in (0, 2)
    c = 1 if ctype == 0 else 3
    raw = zlib.decompress(idat)
    stride = w * c
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int64)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)], dtype=np.uint8).astype(np.int64)
        cur = np.zeros(stride, dtype=np.int64)
        for x in range(stride):
            a = cur[x - c] if x >= c else 0
            b = prev[x]
            cc = prev[x - c] if x >= c else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, cc)][ft]
            cur[x] = (line[x] + pred) & 0xFF
        out[y] = cur
        prev = cur
    return out.reshape(h, w) if c == 1 else out.reshape(h, w, 3)
