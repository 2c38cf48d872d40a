"""LD ingestion without magenpy / zarr / numcodecs: a zarr-v2 *directory store* reader and a blosc-1 frame
codec, feeding the device-side plan builders directly from the compact on-disk form.

What the reference does (outside its own tree): ``bin/viprs_fit:49`` looks for ``.zgroup`` files under
``--ld-dir``, magenpy opens each as an ``LDMatrix`` and ``VIPRS.__init__`` calls
``ld_mat.load(return_symmetric=not low_memory, dtype=...)`` (VIPRS.py:153-172) to obtain ``ld_data / ld_indptr /
leftmost_idx``; with ``dequantize_on_the_fly`` the integer data goes to the kernel as stored and is scaled by
``1 / iinfo(dtype).max`` (VIPRS.py:203-207).  The published panels are int8-quantised, upper-triangular
(no diagonal), CSR-like (``docs/download_ld.md:6-10``).

Layout read here (magenpy's ``LDMatrix`` zarr hierarchy, inferred -- magenpy is not in the reference tree and no
sample store ships with it, so this reader is **parity unpinned**; it is pinned by round trips against the encoder
below and by the zarr-v2 / blosc-1 format specifications):

    <store>/.zgroup                      {"zarr_format": 2}
    <store>/.zattrs                      {"Chromosome": ..., "Sample size": ..., "LD estimator": ..., ...}
    <store>/matrix/data/.zarray + chunks (nnz,)   int8 / int16 / float32 / float64: row j = correlations with
                                                  SNPs j+1 .. j+len_j (upper triangle, diagonal absent)
    <store>/matrix/indptr/.zarray + ...  (m+1,)   int32 / int64 row pointers
    <store>/metadata/<name>/...          per-SNP annotations (snps, bp, a1, a2, maf, ...), read on demand

zarr v2 essentials: ``.zarray`` = JSON {shape, chunks, dtype, compressor, fill_value, order, filters,
dimension_separator}; chunk ``i`` of a 1-d array is the file ``<array>/<i>``; a missing chunk file means
``fill_value``; every stored chunk holds a FULL chunk's worth of elements (the tail is padding).

blosc-1 frame (c-blosc 1.x ``blosc.h`` / ``blosc.c``): 16-byte header {version, versionlz, flags, typesize,
nbytes u32, blocksize u32, cbytes u32}; flags: 0x1 byte-shuffle, 0x2 memcpy'ed, 0x4 bit-shuffle, 0x10 don't
split, bits 5-7 codec (0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd); then one int32 offset per block;
a block is ``nsplits`` streams (``typesize`` streams when splitting applies, else 1), each
{int32 compressed size, bytes} -- a stream whose compressed size equals its uncompressed size is stored raw.
lz4 and zstd come from the system libraries (``liblz4.so.1`` / ``libzstd.so.1``) through ctypes, zlib from the
Python standard library; blosclz and snappy frames are reported as unsupported.
"""
import ctypes
import ctypes.util
import json
import os
import struct
import zlib

import numpy as np

BLOSC_MIN_BUFFERSIZE = 128
BLOSC_MAX_SPLITS = 16
_CODEC_NAME = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_CODEC_CODE = {"lz4": 1, "lz4hc": 1, "zlib": 3, "zstd": 4}


class _Codecs:
    """ctypes bindings of liblz4 / libzstd (loaded on first use)."""
    _lz4 = _zstd = None

    @classmethod
    def lz4(cls):
        if cls._lz4 is None:
            lib = ctypes.CDLL(ctypes.util.find_library("lz4") or "liblz4.so.1")
            lib.LZ4_decompress_safe.restype = ctypes.c_int
            lib.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
            lib.LZ4_compress_default.restype = ctypes.c_int
            lib.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
            lib.LZ4_compressBound.restype = ctypes.c_int
            lib.LZ4_compressBound.argtypes = [ctypes.c_int]
            cls._lz4 = lib
        return cls._lz4

    @classmethod
    def zstd(cls):
        if cls._zstd is None:
            lib = ctypes.CDLL(ctypes.util.find_library("zstd") or "libzstd.so.1")
            lib.ZSTD_decompress.restype = ctypes.c_size_t
            lib.ZSTD_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            lib.ZSTD_compress.restype = ctypes.c_size_t
            lib.ZSTD_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
            lib.ZSTD_compressBound.restype = ctypes.c_size_t
            lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            lib.ZSTD_isError.restype = ctypes.c_uint
            lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
            cls._zstd = lib
        return cls._zstd


def _inflate(codec, src, n_out):
    if codec == 1:
        dst = ctypes.create_string_buffer(n_out)
        n = _Codecs.lz4().LZ4_decompress_safe(src, dst, len(src), n_out)
        if n != n_out:
            raise ValueError(f"blosc: lz4 stream inflated to {n} bytes, expected {n_out}")
        return dst.raw
    if codec == 4:
        dst = ctypes.create_string_buffer(n_out)
        lib = _Codecs.zstd()
        n = lib.ZSTD_decompress(dst, n_out, src, len(src))
        if lib.ZSTD_isError(n) or n != n_out:
            raise ValueError(f"blosc: zstd stream did not inflate to {n_out} bytes")
        return dst.raw
    if codec == 3:
        out = zlib.decompress(src)
        if len(out) != n_out:
            raise ValueError(f"blosc: zlib stream inflated to {len(out)} bytes, expected {n_out}")
        return out
    raise NotImplementedError(f"blosc: codec '{_CODEC_NAME.get(codec, codec)}' is not supported (lz4, zstd, zlib are)")


def _deflate(codec, src, clevel):
    if codec == 1:
        lib = _Codecs.lz4()
        cap = lib.LZ4_compressBound(len(src))
        dst = ctypes.create_string_buffer(cap)
        n = lib.LZ4_compress_default(src, dst, len(src), cap)
        return dst.raw[:n] if n > 0 else None
    if codec == 4:
        lib = _Codecs.zstd()
        cap = lib.ZSTD_compressBound(len(src))
        dst = ctypes.create_string_buffer(cap)
        n = lib.ZSTD_compress(dst, cap, src, len(src), int(clevel))
        return None if lib.ZSTD_isError(n) else dst.raw[:n]
    if codec == 3:
        return zlib.compress(src, int(clevel))
    raise NotImplementedError(f"blosc: cannot encode with codec {codec}")


# ---- shuffle filters (c-blosc shuffle-generic.h / bitshuffle-generic.h) --------------------------------------
def _unshuffle_bytes(block, typesize):
    n = len(block) // typesize
    a = np.frombuffer(block, dtype=np.uint8)
    out = np.empty(len(block), dtype=np.uint8)
    out[:n * typesize] = a[:n * typesize].reshape(typesize, n).T.ravel()
    out[n * typesize:] = a[n * typesize:]                          # leftover bytes are copied as they are
    return out.tobytes()


def _shuffle_bytes(block, typesize):
    n = len(block) // typesize
    a = np.frombuffer(block, dtype=np.uint8)
    out = np.empty(len(block), dtype=np.uint8)
    out[:n * typesize] = a[:n * typesize].reshape(n, typesize).T.ravel()
    out[n * typesize:] = a[n * typesize:]
    return out.tobytes()


def _unshuffle_bits(block, typesize):
    # bit b of element i sits at bit (i % 8) of byte b * (n / 8) + i / 8; only whole groups of 8 elements are
    # transposed, the rest is copied (bitshuffle-generic.c: bshuf_untrans_bit_elem)
    n = (len(block) // typesize) // 8 * 8
    a = np.frombuffer(block, dtype=np.uint8)
    out = np.empty(len(block), dtype=np.uint8)
    if n:
        bits = np.unpackbits(a[:n * typesize].reshape(8 * typesize, n // 8), axis=1, bitorder="little")   # [bit][elem]
        out[:n * typesize] = np.packbits(bits.T, axis=1, bitorder="little").ravel()                        # [elem][bit]
    out[n * typesize:] = a[n * typesize:]
    return out.tobytes()


def _shuffle_bits(block, typesize):
    n = (len(block) // typesize) // 8 * 8
    a = np.frombuffer(block, dtype=np.uint8)
    out = np.empty(len(block), dtype=np.uint8)
    if n:
        bits = np.unpackbits(a[:n * typesize].reshape(n, typesize), axis=1, bitorder="little")            # [elem][bit]
        out[:n * typesize] = np.packbits(bits.T, axis=1, bitorder="little").ravel()                        # [bit][elem]
    out[n * typesize:] = a[n * typesize:]
    return out.tobytes()


# ---- blosc-1 frames ---------------------------------------------------------------------------------------------
def blosc_decompress(buf):
    """One blosc-1 frame -> bytes."""
    buf = bytes(buf)
    if len(buf) < 16:
        raise ValueError("blosc: frame shorter than its 16-byte header")
    version, _, flags, typesize, nbytes, blocksize, cbytes = struct.unpack_from("<BBBBIII", buf, 0)
    if version != 2:
        raise ValueError(f"blosc: unsupported frame format version {version}")
    if cbytes > len(buf):
        raise ValueError("blosc: frame is truncated")
    if nbytes == 0:
        return b""
    if flags & 0x2:                                                 # memcpy'ed: raw bytes behind the header
        return buf[16:16 + nbytes]
    codec = flags >> 5
    dont_split = bool(flags & 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", buf, 16)
    out = []
    for bi in range(nblocks):
        bsize = min(blocksize, nbytes - bi * blocksize)
        leftover = bsize != blocksize
        split = (not dont_split and typesize <= BLOSC_MAX_SPLITS and blocksize // typesize >= BLOSC_MIN_BUFFERSIZE
                 and not leftover)
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        pos = bstarts[bi]
        parts = []
        for _ in range(nsplits):
            (cb,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            if cb < 0 or pos + cb > len(buf):
                raise ValueError("blosc: corrupt stream length")
            parts.append(buf[pos:pos + cb] if cb == neblock else _inflate(codec, buf[pos:pos + cb], neblock))
            pos += cb
        block = b"".join(parts)
        if flags & 0x1:
            block = _unshuffle_bytes(block, typesize)
        elif flags & 0x4:
            block = _unshuffle_bits(block, typesize)
        out.append(block)
    return b"".join(out)


def blosc_compress(data, typesize=1, cname="lz4", clevel=5, shuffle=1, blocksize=None, split=True):
    """bytes -> one blosc-1 frame (the encoder the round-trip tests and `write_ld_store` use).
    shuffle: 0 none, 1 byte shuffle, 2 bit shuffle (numcodecs.Blosc's NOSHUFFLE / SHUFFLE / BITSHUFFLE);
    split=False sets the don't-split flag (what c-blosc >= 1.15 does for lz4 / zstd): one stream per block."""
    data = bytes(data)
    codec = _CODEC_CODE[cname]
    nbytes = len(data)
    if blocksize is None:
        blocksize = 32 * 1024
    blocksize = max(typesize, min(blocksize, max(nbytes, 1)) // typesize * typesize)
    flags = (codec << 5) | {0: 0, 1: 0x1, 2: 0x4}[shuffle] | (0 if split else 0x10)
    if typesize == 1:
        flags &= ~0x1                                               # c-blosc skips the byte shuffle for 1-byte types
    if nbytes < BLOSC_MIN_BUFFERSIZE:                               # tiny buffers are stored raw
        return struct.pack("<BBBBIII", 2, 1, flags | 0x2, typesize, nbytes, blocksize, 16 + nbytes) + data
    nblocks = (nbytes + blocksize - 1) // blocksize
    body, bstarts, pos = [], [], 16 + 4 * nblocks
    for bi in range(nblocks):
        block = data[bi * blocksize:(bi + 1) * blocksize]
        leftover = len(block) != blocksize
        if flags & 0x1:
            block = _shuffle_bytes(block, typesize)
        elif flags & 0x4:
            block = _shuffle_bits(block, typesize)
        do_split = (split and typesize <= BLOSC_MAX_SPLITS and blocksize // typesize >= BLOSC_MIN_BUFFERSIZE
                    and not leftover)
        nsplits = typesize if do_split else 1
        neblock = len(block) // nsplits
        bstarts.append(pos)
        for k in range(nsplits):
            raw = block[k * neblock:(k + 1) * neblock]
            comp = _deflate(codec, raw, clevel)
            if comp is None or len(comp) >= len(raw):
                comp = raw                                          # incompressible: stored raw (length == neblock)
            body.append(struct.pack("<i", len(comp)) + comp)
            pos += 4 + len(comp)
    head = struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, pos)
    return head + struct.pack(f"<{nblocks}i", *bstarts) + b"".join(body)


# ---- zarr v2 arrays in a directory store ------------------------------------------------------------------------
class ZarrArray:
    """Read-only 1-d (or C-ordered n-d, read whole) zarr-v2 array stored as a directory of chunk files."""

    def __init__(self, path):
        self.path = path
        meta_file = os.path.join(path, ".zarray")
        if not os.path.exists(meta_file):
            raise FileNotFoundError(f"{path}: not a zarr v2 array (no .zarray)")
        with open(meta_file) as f:
            meta = json.load(f)
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path}: zarr_format {meta.get('zarr_format')} is not supported (only 2)")
        if meta.get("filters"):
            raise NotImplementedError(f"{path}: zarr filters are not supported")
        if meta.get("order", "C") != "C" and len(meta["shape"]) > 1:
            raise NotImplementedError(f"{path}: Fortran-ordered chunks are not supported")
        self.shape = tuple(int(s) for s in meta["shape"])
        self.chunks = tuple(int(c) for c in meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill_value = meta.get("fill_value")
        self.sep = meta.get("dimension_separator", ".")
        self.compressor = meta.get("compressor")
        if self.compressor is not None and self.compressor.get("id") not in ("blosc", "zlib", "zstd", "lz4"):
            raise NotImplementedError(f"{path}: compressor '{self.compressor.get('id')}' is not supported")

    def _decode(self, raw):
        c = self.compressor
        if c is None:
            return raw
        if c["id"] == "blosc":
            return blosc_decompress(raw)
        if c["id"] == "zlib":
            return zlib.decompress(raw)
        n_out = int(np.prod(self.chunks)) * self.dtype.itemsize
        if c["id"] == "zstd":
            return _inflate(4, raw, n_out)
        # numcodecs.LZ4 prefixes the stream with the uncompressed size (int32 LE)
        return _inflate(1, raw[4:], struct.unpack_from("<i", raw, 0)[0])

    def _chunk(self, idx):
        """Chunk `idx` (tuple) as a flat array of prod(chunks) elements."""
        name = self.sep.join(str(i) for i in idx)
        f = os.path.join(self.path, *name.split("/")) if self.sep == "/" else os.path.join(self.path, name)
        n = int(np.prod(self.chunks))
        if not os.path.exists(f):
            fill = 0 if self.fill_value in (None, "NaN") else self.fill_value
            return np.full(n, fill, dtype=self.dtype)
        with open(f, "rb") as fh:
            raw = self._decode(fh.read())
        a = np.frombuffer(raw, dtype=self.dtype)
        if a.size != n:
            raise ValueError(f"{f}: chunk holds {a.size} elements, expected {n}")
        return a

    def read(self, start=0, stop=None):
        """Elements [start, stop) of a 1-d array (only the chunks that overlap the range are read);
        n-d arrays are read whole."""
        if len(self.shape) != 1:
            return self._read_nd()
        n = self.shape[0]
        stop = n if stop is None else min(int(stop), n)
        start = max(0, int(start))
        out = np.empty(max(0, stop - start), dtype=self.dtype)
        if out.size == 0:
            return out
        c = self.chunks[0]
        for ci in range(start // c, (stop - 1) // c + 1):
            lo, hi = max(start, ci * c), min(stop, (ci + 1) * c)
            out[lo - start:hi - start] = self._chunk((ci,))[lo - ci * c:hi - ci * c]
        return out

    def _read_nd(self):
        out = np.empty(self.shape, dtype=self.dtype)
        grid = [range((s + c - 1) // c) for s, c in zip(self.shape, self.chunks)]
        for idx in np.ndindex(*[len(g) for g in grid]):
            blk = self._chunk(idx).reshape(self.chunks)
            sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, self.chunks, self.shape))
            out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def __len__(self):
        return self.shape[0]


def write_zarr_array(path, array, chunks=None, cname="zstd", clevel=5, shuffle=1, blocksize=None):
    """Write a 1-d array as a zarr-v2 directory array with blosc-compressed chunks."""
    a = np.ascontiguousarray(array)
    if a.ndim != 1:
        raise ValueError("write_zarr_array: 1-d arrays only")
    os.makedirs(path, exist_ok=True)
    c = int(chunks or max(1, min(a.shape[0], 1 << 20)))
    meta = {"zarr_format": 2, "shape": [int(a.shape[0])], "chunks": [c], "dtype": a.dtype.str, "fill_value": 0,
            "order": "C", "filters": None, "dimension_separator": ".",
            "compressor": {"id": "blosc", "cname": cname, "clevel": int(clevel), "shuffle": int(shuffle),
                           "blocksize": int(blocksize or 0)}}
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f)
    for ci in range((a.shape[0] + c - 1) // c):
        part = a[ci * c:(ci + 1) * c]
        if part.shape[0] < c:                                       # stored chunks are always full-size
            part = np.concatenate([part, np.zeros(c - part.shape[0], dtype=a.dtype)])
        with open(os.path.join(path, str(ci)), "wb") as f:
            f.write(blosc_compress(part.tobytes(), a.dtype.itemsize, cname, clevel, shuffle, blocksize))


# ---- the LD matrix of one chromosome ----------------------------------------------------------------------------------
class ZarrLDMatrix:
    """Duck-typed on what ``VIPRS.__init__`` asks of a magenpy ``LDMatrix`` (VIPRS.py:153-191): ``stored_dtype``,
    ``load(return_symmetric=..., dtype=...)`` -> ``ld_data / ld_indptr / leftmost_idx``, ``get_lambda_min()``.

    Only the stored (upper-triangular) form is handed out: asking for ``return_symmetric=True`` raises ValueError,
    which makes ``VIPRS(low_memory=False)`` build the symmetric windows ON THE DEVICE from the compact store
    (``viprs_plan_create_expanded``) instead of in host memory."""

    def __init__(self, path):
        self.path = path
        if not os.path.exists(os.path.join(path, ".zgroup")):
            raise FileNotFoundError(f"{path}: not a zarr v2 group (no .zgroup)")
        self.attrs = {}
        za = os.path.join(path, ".zattrs")
        if os.path.exists(za):
            with open(za) as f:
                self.attrs = json.load(f)
        self._data = ZarrArray(os.path.join(path, "matrix", "data"))
        self._indptr = ZarrArray(os.path.join(path, "matrix", "indptr"))
        self.stored_dtype = self._data.dtype
        self.n_snps = len(self._indptr) - 1
        self._ip = None

    @property
    def chromosome(self):
        return self.attrs.get("Chromosome")

    @property
    def sample_size(self):
        return self.attrs.get("Sample size")

    @property
    def ld_estimator(self):
        return self.attrs.get("LD estimator")

    @property
    def dq_scale(self):
        return 1.0 / np.iinfo(self.stored_dtype).max if np.issubdtype(self.stored_dtype, np.integer) else 1.0

    def indptr(self):
        if self._ip is None:
            self._ip = self._indptr.read().astype(np.int64)
        return self._ip

    def metadata(self, name):
        return ZarrArray(os.path.join(self.path, "metadata", name)).read()

    class _Loaded:
        def __init__(self, lb, ip, data):
            self.leftmost_idx, self.ld_indptr, self.ld_data = lb, ip, data

    def _cast(self, data, dtype):
        if dtype is None or np.dtype(dtype) == data.dtype:
            return data
        if np.issubdtype(data.dtype, np.integer) and np.issubdtype(np.dtype(dtype), np.floating):
            # dequantise at load time, as magenpy does when a float dtype is asked for: a DIVISION by the integer
            # type's maximum (multiplying by the rounded reciprocal differs in the last bit for some values)
            return (data.astype(dtype) / np.dtype(dtype).type(np.iinfo(data.dtype).max)).astype(dtype)
        return data.astype(dtype)

    def load(self, return_symmetric=False, dtype=None):
        if return_symmetric:
            raise ValueError("ZarrLDMatrix hands out the stored upper-triangular form only; the symmetric form is "
                             "built on the device (VIPRS(low_memory=False) does that by itself)")
        ip = self.indptr()
        data = self._cast(self._data.read(0, int(ip[-1])), dtype)
        return ZarrLDMatrix._Loaded(np.arange(1, self.n_snps + 1, dtype=np.int32), ip, data)

    def load_rows(self, start, stop, dtype=None):
        """Rows [start, stop) only (LD blocks of one rank): (leftmost_idx, indptr re-based to 0, data); reads
        just the chunks that hold those rows."""
        ip = self.indptr()
        data = self._cast(self._data.read(int(ip[start]), int(ip[stop])), dtype)
        return np.arange(start + 1, stop + 1, dtype=np.int32), ip[start:stop + 1] - ip[start], data

    # Which of the two candidate formulas get_lambda_min uses when the store carries extremal eigenvalues and a
    # min_max_ratio r > 0 is asked for.  None (the default) REFUSES: magenpy (not in the reference tree, not in this
    # image) defines the formula; `tools/check_store.py` decides it against a real magenpy installation.
    #   "one_plus_r"  : max((r lambda_max - lambda_min) / (1 + r), 0)   (as recalled from magenpy 0.1.x)
    #   "one_minus_r" : max((r lambda_max - lambda_min) / (1 - r), 0)   (the algebraic solution of
    #                   (lambda_min + x) = r (lambda_max + x))
    lambda_min_formula = None

    def get_lambda_min(self, min_max_ratio=0.0, formula=None):
        """The ridge penalty `VIPRS(lambda_min='infer')` asks for (`ld_mat.get_lambda_min(min_max_ratio=1e-3)`,
        VIPRS.py:186-191), from the extremal eigenvalues magenpy stores under 'Spectral properties' -> 'Extremal'.

        * no spectral attributes in the store: 0.0 -- what the reference documents (VIPRS.py:189-190: "If this is not
          available, we set the minimum eigenvalue to 0");
        * r = min_max_ratio == 0: |min(lambda_min, 0)| (no formula involved);
        * r > 0: PARITY UNPINNED -- the two candidates above differ by (1 - r) / (1 + r) (0.2 % at VIPRS's r = 1e-3) and
          nothing in the reference tree or this image decides between them.  Rather than hand `fit()` an unverified
          ridge, this RAISES unless a formula is chosen explicitly (`formula=` here, or the class / instance attribute
          `lambda_min_formula`); passing a numeric `lambda_min` to VIPRS avoids the question altogether."""
        sp = self.attrs.get("Spectral properties") or {}
        ext = sp.get("Extremal") or sp.get("extremal") or sp

        def pick(d, *names):
            for n in names:
                if isinstance(d, dict) and d.get(n) is not None:
                    return float(d[n])
            return None

        lam_min, lam_max = pick(ext, "min", "Min"), pick(ext, "max", "Max")
        if lam_min is None:
            return 0.0
        r = float(min_max_ratio or 0.0)
        if r > 0.0 and lam_max is not None:
            formula = formula or self.lambda_min_formula
            if formula not in ("one_plus_r", "one_minus_r"):
                raise UnpinnedLambdaMinError(
                    f"{self.path}: the store carries extremal eigenvalues (min {lam_min}, max {lam_max}) but the formula "
                    "magenpy's LDMatrix.get_lambda_min(min_max_ratio) applies to them could not be verified (magenpy is "
                    "not available where this reader was written).  Pass a numeric lambda_min to VIPRS(...), or choose "
                    "ZarrLDMatrix.lambda_min_formula = 'one_plus_r' | 'one_minus_r' (tools/check_store.py tells which one "
                    "a magenpy installation agrees with).")
            den = 1.0 + r if formula == "one_plus_r" else 1.0 - r
            return max((r * lam_max - lam_min) / den, 0.0)
        return abs(min(lam_min, 0.0))


class UnpinnedLambdaMinError(NotImplementedError):
    """`lambda_min='infer'` on a store with spectral attributes: the formula is unverified (ZarrLDMatrix.get_lambda_min)."""


def write_ld_store(path, ld_indptr, ld_data, attrs=None, chunks=None, cname="zstd", clevel=5, shuffle=1,
                   metadata=None):
    """Write the compact upper-triangular LD of one chromosome as a zarr-v2 group in the layout above."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, ".zgroup"), "w") as f:
        json.dump({"zarr_format": 2}, f)
    with open(os.path.join(path, ".zattrs"), "w") as f:
        json.dump(dict(attrs or {}), f)
    for sub in ("matrix", "metadata"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
        with open(os.path.join(path, sub, ".zgroup"), "w") as f:
            json.dump({"zarr_format": 2}, f)
    write_zarr_array(os.path.join(path, "matrix", "data"), ld_data, chunks, cname, clevel, shuffle)
    write_zarr_array(os.path.join(path, "matrix", "indptr"), ld_indptr, None, cname, clevel, shuffle)
    for name, arr in (metadata or {}).items():
        write_zarr_array(os.path.join(path, "metadata", name), arr, None, cname, clevel, shuffle)


def find_ld_stores(ld_dir):
    """Directories under `ld_dir` that hold an LD matrix (a `.zgroup` with a `matrix/data` array), as
    bin/viprs_fit:49 collects them; keyed by the store's 'Chromosome' attribute when it has one."""
    found = {}
    for root, dirs, files in os.walk(ld_dir):
        if ".zgroup" in files and os.path.exists(os.path.join(root, "matrix", "data", ".zarray")):
            m = ZarrLDMatrix(root)
            key = m.chromosome if m.chromosome is not None else os.path.basename(root)
            try:
                key = int(key)
            except (TypeError, ValueError):
                pass
            found[key] = m
            dirs[:] = []
    return found
