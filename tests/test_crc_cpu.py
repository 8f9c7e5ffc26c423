"""The arithmetic of svx_crc.hip (svx_bgzf_crc32) restated in plain Python against zlib.crc32: 64 interleaved sub-sequences of
the block right-aligned in a 16,384-dword buffer, register <- register * x^2048 + dword per step, one multiply by
x^(32 (64 - lane)) per lane, xor of the 64 registers; initial value as the complement of the first four bytes.  The GPU test
(tests/test_gpu_inflate.py) checks the kernel itself against zlib on every block the inflate tests decode."""
import random
import zlib

POLY = 0xEDB88320


def _times_x(r, bits):
    for _ in range(bits):
        r = (r >> 1) ^ (POLY if r & 1 else 0)
    return r


def _gf_mul(a, b):
    r = 0
    for _ in range(32):
        if a & 0x80000000:
            r ^= b
        a = (a << 1) & 0xFFFFFFFF
        b = _times_x(b, 1)
    return r


_STRIDE = [[_times_x(b << (8 * k), 2048) for b in range(256)] for k in range(4)]
_TAIL = [_times_x(0x80000000, 32 * (64 - lane)) for lane in range(64)]


def crc_like_the_kernel(data):
    n = len(data)
    if n < 4:
        r = 0xFFFFFFFF
        for x in data:
            r = _times_x(r ^ x, 8)
        return r ^ 0xFFFFFFFF
    pad = 65536 - n
    virt = bytearray(pad) + bytearray(data)
    for k in range(4):
        virt[pad + k] ^= 0xFF
    acc = 0
    g0 = (pad >> 2) // 64
    for lane in range(64):
        reg = 0
        for g in range(g0, 256):
            v = 4 * (64 * g + lane)
            d = int.from_bytes(virt[v:v + 4], "little")
            reg = _STRIDE[0][reg & 255] ^ _STRIDE[1][(reg >> 8) & 255] ^ _STRIDE[2][(reg >> 16) & 255] ^ _STRIDE[3][reg >> 24] ^ d
        acc ^= _gf_mul(reg, _TAIL[lane])
    return acc ^ 0xFFFFFFFF


def test_interleaved_crc_equals_zlib():
    rng = random.Random(3)
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 63, 64, 255, 256, 257, 1000, 4097, 65279, 65280, 65535, 65536):
        data = bytes(rng.getrandbits(8) for _ in range(n))
        assert crc_like_the_kernel(data) == zlib.crc32(data), n
