"""Test scaffolding: raw DEFLATE streams made by libdeflate (the compressor htslib - what pysam, samtools and the aligners'
BAM writers sit on - uses for BGZF blocks when it is built with it), through ctypes.  Another compressor than zlib: its own
block splitting, its own length-limited Huffman codes, near-optimal parsing at levels 10-12."""
import ctypes as C
import ctypes.util

_lib = None


def available():
    return _load() is not None


def _load():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library('deflate')
        if not name:
            _lib = False
            return None
        try:
            lib = C.CDLL(name)
        except OSError:
            _lib = False
            return None
        lib.libdeflate_alloc_compressor.restype = C.c_void_p
        lib.libdeflate_alloc_compressor.argtypes = [C.c_int]
        lib.libdeflate_deflate_compress.restype = C.c_size_t
        lib.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        lib.libdeflate_deflate_compress_bound.restype = C.c_size_t
        lib.libdeflate_deflate_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
        lib.libdeflate_free_compressor.argtypes = [C.c_void_p]
        _lib = lib
    return _lib or None


def deflate(raw, level):
    """raw -> a raw DEFLATE stream (no zlib / gzip wrapper), libdeflate level 0-12."""
    lib = _load()
    c = lib.libdeflate_alloc_compressor(int(level))
    try:
        cap = lib.libdeflate_deflate_compress_bound(c, len(raw))
        out = C.create_string_buffer(cap)
        n = lib.libdeflate_deflate_compress(c, bytes(raw), len(raw), out, cap)
        assert n > 0
        return out.raw[:n]
    finally:
        lib.libdeflate_free_compressor(c)
