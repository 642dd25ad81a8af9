"""The GF(2) identities k_assemble's CRCs stand on (flake_amd/csrc/k4_assemble.hip), checked on the CPU against the
oracle's crc16 / crc8 (crc.c:24-94 restated in oracle/flake_oracle.c): the kernel never reads a frame back -- a lane
carries the CRC-16 of its own 16-byte quads, moves it to the frame's end by constant products, the lanes' values are
XORed, and the zero bytes that fill the last quad are taken off again by x^-8; the header's CRC-8 is the XOR of its
fields' CRCs moved to the header's end."""
import numpy as np
import pytest

P16 = 0x18005        # x^16 + x^15 + x^2 + 1
P8 = 0x107           # x^8 + x^2 + x + 1


def mulmod(a, b, poly=P16, deg=16):
    r = 0
    for i in range(deg - 1, -1, -1):
        r <<= 1
        if r >> deg:
            r ^= poly
        if (b >> i) & 1:
            r ^= a
    return r


def xpow(e, poly=P16, deg=16):
    r, b = 1, 2
    while e:
        if e & 1:
            r = mulmod(r, b, poly, deg)
        b = mulmod(b, b, poly, deg)
        e >>= 1
    return r


@pytest.fixture(scope="module")
def oracle():
    import oraclelib
    return oraclelib.Oracle()


def test_crc16_of_a_concatenation(oracle):
    r = np.random.RandomState(5)
    for _ in range(50):
        a = r.randint(0, 256, r.randint(0, 300)).astype(np.uint8)
        b = r.randint(0, 256, r.randint(0, 300)).astype(np.uint8)
        whole = oracle.crc16(np.concatenate([a, b]))
        assert whole == mulmod(oracle.crc16(a), xpow(8 * len(b))) ^ oracle.crc16(b)


def test_crc16_x_has_order_32767_and_padding_comes_off(oracle):
    assert xpow(32767) == 1
    r = np.random.RandomState(6)
    for _ in range(50):
        body = r.randint(0, 256, r.randint(1, 500)).astype(np.uint8)
        pad = (-len(body)) % 16
        padded = oracle.crc16(np.concatenate([body, np.zeros(pad, np.uint8)]))
        assert oracle.crc16(body) == mulmod(padded, xpow(32767 - 8 * pad))


def test_crc16_from_strided_quads(oracle):
    """What the kernel's lanes do: lane t owns the quads last - t - 64 i; Horner over its own quads with x^(8 * 16 * 64),
    then x^(8 * 16 * t), XOR over the lanes, x^-8 per padding byte."""
    r = np.random.RandomState(7)
    for nbytes in (1, 15, 16, 17, 1000, 1024, 4097, 20000):
        body = r.randint(0, 256, nbytes).astype(np.uint8)
        nq = (nbytes + 15) // 16
        padded = np.concatenate([body, np.zeros(16 * nq - nbytes, np.uint8)])
        step = xpow(8 * 16 * 64)
        total = 0
        for t in range(64):
            last = nq - 1 - t
            if last < 0:
                continue
            acc = 0
            for q in range(last % 64, last + 1, 64):
                acc = mulmod(acc, step) ^ oracle.crc16(padded[16 * q:16 * q + 16])
            total ^= mulmod(acc, xpow(8 * 16 * t))
        assert mulmod(total, xpow(32767 - 8 * (16 * nq - nbytes))) == oracle.crc16(body)


def test_crc8_of_header_fields(oracle):
    """The frame header's CRC-8 as the XOR of its fields' CRCs, each followed by the zero bytes behind it."""
    r = np.random.RandomState(8)
    for _ in range(50):
        hdr = r.randint(0, 256, r.randint(5, 16)).astype(np.uint8)
        cuts = sorted(set([0, 4] + list(r.randint(5, len(hdr) + 1, 3)) + [len(hdr)]))
        total = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            field = np.concatenate([hdr[a:b], np.zeros(len(hdr) - b, np.uint8)])
            total ^= oracle.crc8(field)
        assert total == oracle.crc8(hdr)
