"""The arithmetic of the in-kernel read staging (round 6: rowbowt_amd/csrc/rbg_runs_device.hpp stage_chunk / stage_read / staged_bits, the register tables of
capi/upload_runs.ipp) restated in numpy and checked against plain 2-bit packing -- on the CPU, so that an edit of the constants (the v_perm_b32 tables, the
0x40100401 pack multiplier, the nibble masks, the funnel shift) is caught without a GPU.  What it stages are the symbols RowBowt::find_range consumes right to left
(rowbowt.hpp:121-131): symbol q[m - 1 - t] at bits [2t, 2t + 2) of the read's code words."""
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M32 = 0xFFFFFFFF


def tables(major_bytes):
    """upload_runs.ipp: a shift under which the four major bytes hash to four different three-bit values; code / expected-byte tables of eight places"""
    for sh in range(6):
        code, byte, used = [0] * 8, [0] * 8, [False] * 8
        ok = True
        for m, b in enumerate(major_bytes):
            t = (b >> sh) & 7
            ok = ok and not used[t]
            used[t], code[t], byte[t] = True, m, b
        if not ok:
            continue
        for t in range(8):
            if not used[t]:
                byte[t] = ((t ^ 1) & 7) << sh
        return sh, code, byte
    return None


def perm(table8, idx_word):
    """v_perm_b32 with selectors 0..7: byte i of the result = table8[byte i of idx_word]"""
    return sum(table8[(idx_word >> (8 * i)) & 7] << (8 * i) for i in range(4))


def stage_chunk(chunk16, sh, code, byte):
    """-> (32-bit code word of the chunk's 16 bytes in consumption order: byte 15 first; 4 diff words)"""
    c8, diff = [], []
    for t in range(4):
        x = int.from_bytes(chunk16[4 * t:4 * t + 4], "little")
        idx = (x >> sh) & 0x07070707
        diff.append(x ^ perm(byte, idx))
        c8.append(((perm(code, idx) * 0x40100401) & M32) >> 24)
    return c8[3] | (c8[2] << 8) | (c8[1] << 16) | (c8[0] << 24), diff


def nibble_bytes(nib):
    return ((((nib & 15) * 0x00204081) & 0x01010101) * 0xFF) & M32


def stage_read(buf, beg, end, sh, code, byte):
    """-> (code words, bad) as the device function makes them"""
    if end <= beg:
        return [], False
    ci, ci_lo = (end - 1) >> 4, beg >> 4
    hi_b, lo_b = (end - 1) & 15, beg & 15
    skip2 = 2 * (15 - hi_b)
    nwords = (end - beg + 15) >> 4
    prev, d = stage_chunk(buf[16 * ci:16 * ci + 16], sh, code, byte)
    vm = (2 << hi_b) - 1
    if ci == ci_lo:
        vm &= ~((1 << lo_b) - 1)
    bad = (d[0] & nibble_bytes(vm)) | (d[1] & nibble_bytes(vm >> 4)) | (d[2] & nibble_bytes(vm >> 8)) | (d[3] & nibble_bytes(vm >> 12))
    words = []
    while nwords:
        nxt = 0
        if ci > ci_lo:
            ci -= 1
            nxt, d = stage_chunk(buf[16 * ci:16 * ci + 16], sh, code, byte)
            if ci == ci_lo:
                vm = ~((1 << lo_b) - 1)
                bad |= (d[0] & nibble_bytes(vm)) | (d[1] & nibble_bytes(vm >> 4)) | (d[2] & nibble_bytes(vm >> 8)) | (d[3] & nibble_bytes(vm >> 12))
            else:
                bad |= d[0] | d[1] | d[2] | d[3]
        words.append((((nxt << 32) | prev) >> skip2) & M32)       # v_alignbit_b32(next, prev, skip2)
        prev = nxt
        nwords -= 1
    return words, bad != 0


def staged_bits(words, t, nsym):
    w0 = words[t >> 4] if (t >> 4) < len(words) else 0
    w1 = words[(t >> 4) + 1] if (t >> 4) + 1 < len(words) else 0
    return (((w1 << 32) | w0) >> ((t & 15) * 2)) & ((1 << (2 * nsym)) - 1)


def test_tables_for_usual_alphabets_and_an_impossible_one():
    for alpha, want_shift in ((b"ACGT", 0), (b"acgt", 0), (b"ACGN", 0), (b"\x41\x49\x51\x59", 2), (b"\x10\x20\x30\x40", 3)):
        sh, code, byte = tables(list(alpha))
        assert sh == want_shift
        for m, b in enumerate(alpha):
            assert code[(b >> sh) & 7] == m and byte[(b >> sh) & 7] == b
        for t in range(8):     # a place no symbol owns never equals a byte that hashes to it
            if byte[t] not in alpha:
                assert (byte[t] >> sh) & 7 != t
    assert tables([0x02, 0x03, 0x82, 0x83]) is None      # bits 0 and 7 never share a three-bit window: the byte walk


def test_staged_codes_equal_plain_packing_for_every_alignment():
    rng = np.random.default_rng(3)
    alpha = b"ACGT"
    sh, code, byte = tables(list(alpha))
    for trial in range(400):
        m = int(rng.integers(1, 257))
        beg = int(rng.integers(0, 48))
        buf = bytearray(rng.integers(0, 256, size=beg + m + 64, dtype=np.uint8).tobytes())     # (garbage around the read: it must not matter)
        read = bytes(alpha[i] for i in rng.integers(0, 4, size=m))
        buf[beg:beg + m] = read
        bad_at = None
        if trial % 5 == 0:                                  # one byte outside the alphabet, anywhere in the read
            bad_at = int(rng.integers(0, m))
            buf[beg + bad_at] = int(rng.choice([ord("N"), ord("a"), 0, 255, ord("A") ^ 0x80, ord("C") | 0x08]))
        words, bad = stage_read(bytes(buf), beg, beg + m, sh, code, byte)
        assert bad == (bad_at is not None), (trial, m, beg, bad_at)
        if bad:
            continue
        assert len(words) == (m + 15) // 16
        want = [alpha.index(read[m - 1 - t]) for t in range(m)]
        for t in range(m):
            assert staged_bits(words, t, 1) == want[t], (trial, m, beg, t)
        for _ in range(20):                                 # a k-mer step's table index / the ftab word: up to 16 symbols from any consumption index
            t = int(rng.integers(0, m))
            k = int(rng.integers(1, min(16, m - t) + 1))
            assert staged_bits(words, t, k) == sum(want[t + j] << (2 * j) for j in range(k))


def test_the_constants_in_the_source_are_the_ones_modelled_here():
    src = open(os.path.join(ROOT, "rowbowt_amd", "csrc", "rbg_runs_device.hpp")).read()
    assert "0x40100401u" in src and "0x00204081u" in src and "0x07070707u" in src and re.search(r"kStageCap = 256u", src)
    up = open(os.path.join(ROOT, "rowbowt_amd", "csrc", "capi", "upload_runs.ipp")).read()
    assert "((t ^ 1u) & 7u) << sh" in up and "sh <= 5" in up
