"""Feature file readers of the data path (reference msmctts/utils/utils.py:20-116 and the ``parse_*`` methods of
msmctts/datasets/base_dataset.py:120-229): every reader returns ``[time, dim]``-shaped data from ``start`` for ``length``
frames (``length <= 0``: to the end), or only the shape.

A path may name a member of a zip archive as ``archive.zip:member``.  ``.npy`` files are read through the header and
ONE ranged read of the payload (C order: a contiguous byte range; Fortran order: one range per column), never the
whole file -- a training item is a 0.5 s window of a multi-minute recording.
"""
import array
import io
import os
import wave
import zipfile

import numpy as np

_ZIPS = {}


def open_source(path):
    """(file object, close?) for a plain path or an ``archive.zip:member`` name"""
    if not os.path.isfile(path) and ':' in path:
        archive, member = path.split(':', 1)
        z = _ZIPS.get(archive)
        if z is None:
            z = _ZIPS[archive] = zipfile.ZipFile(archive, 'r')
        return io.BytesIO(z.read(member)), True
    return open(path, 'rb'), True


def _window(n, start, length):
    stop = n if length <= 0 else min(n, start + length)
    return start, max(start, stop)


def read_npy(src, start=0, length=-1, shape_only=False):
    fid, owned = (src, False) if hasattr(src, 'read') else open_source(src)
    try:
        version = np.lib.format.read_magic(fid)
        header = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
        shape, fortran, dtype = header(fid)
        if shape_only:
            return tuple(shape)
        if dtype.hasobject:
            raise ValueError('object arrays are not feature files')
        if len(shape) == 0:
            return np.frombuffer(fid.read(dtype.itemsize), dtype=dtype).reshape(())
        if not start < shape[0]:
            raise ValueError('window start %d beyond %d frames' % (start, shape[0]))
        a, b = _window(shape[0], start, length)
        inner = int(np.prod(shape[1:], dtype=np.int64)) if len(shape) > 1 else 1
        base = fid.tell()
        if not fortran:
            fid.seek(base + a * inner * dtype.itemsize)
            data = np.frombuffer(fid.read((b - a) * inner * dtype.itemsize), dtype=dtype)
            tail = tuple(shape[1:]) if len(shape) > 1 else ((1,) if length > 0 else ())   # (a windowed vector reads as [n, 1])
            return data.reshape((b - a,) + tail).copy()
        if len(shape) != 2:
            raise RuntimeError('Fortran-ordered feature files must be matrices')
        out = np.empty((b - a, shape[1]), dtype=dtype)
        for col in range(shape[1]):
            fid.seek(base + (col * shape[0] + a) * dtype.itemsize)
            out[:, col] = np.frombuffer(fid.read((b - a) * dtype.itemsize), dtype=dtype)
        return out
    finally:
        if owned:
            fid.close()


def read_raw_float32(src, dimension=None, start=0, length=-1, shape_only=False):
    """headerless float32 frames (``.dat`` / ``.mgc`` / ``.ap``)"""
    fid, owned = (src, False) if hasattr(src, 'read') else open_source(src)
    try:
        buf = array.array('f')
        buf.frombytes(fid.read())
    finally:
        if owned:
            fid.close()
    data = np.frombuffer(buf, dtype=np.float32).reshape(-1, dimension or 1)
    return data.shape if shape_only else data


def read_wav(src, start=0, length=-1, shape_only=False):
    """PCM WAV -> float32 [frames, 1] in [-1, 1) (first channel), sample rate; 8 / 16 / 24 / 32-bit integer files through
    the standard library, IEEE-float files through scipy."""
    fid, owned = (src, False) if hasattr(src, 'read') else open_source(src)
    try:
        try:
            with wave.open(fid, 'rb') as w:
                n, ch, width, rate = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
                if shape_only:
                    return (n, ch)
                a, b = _window(n, start, length)
                w.setpos(a)
                raw = w.readframes(b - a)
            if width == 1:
                x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
            elif width == 2:
                x = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
            elif width == 3:
                b3 = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
                v = b3[:, 0] | (b3[:, 1] << 8) | (b3[:, 2] << 16)
                x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
            else:
                x = np.frombuffer(raw, dtype='<i4').astype(np.float32) / 2147483648.0
            x = x.reshape(-1, ch)
        except wave.Error:                     # not integer PCM
            from scipy.io import wavfile
            fid.seek(0)
            rate, x = wavfile.read(fid)
            x = np.asarray(x, dtype=np.float32).reshape(len(x), -1)
            if shape_only:
                return x.shape
            a, b = _window(len(x), start, length)
            x = x[a:b]
        return x[:, :1].astype(np.float32), rate
    finally:
        if owned:
            fid.close()


def read_torch(src, dimension=None, start=0, length=-1, shape_only=False):
    import torch
    data = torch.load(src, map_location='cpu').squeeze(0).numpy()
    if dimension is not None and data.shape[0] == dimension:
        data = data.T
    if shape_only:
        return data.shape
    a, b = _window(data.shape[0], start, length)
    return data[a:b]
