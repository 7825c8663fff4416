"""A numpy reading of the variant shard's file format (colormipsearch_amd/csrc/cds_variants.hip): the checker of the host-side packer
in the CPU tests and of the device decoders in the GPU tests.  Test infrastructure, not shipped."""
import struct

import numpy as np


def read_shard(path):
    """-> width, height, [(flags, rgb u8[H,W,3], gradient u16[H,W] | None, zgap u8[H,W,3] | None) per target]"""
    b = open(path, "rb").read()
    magic, W, H, n, index_off, file_bytes = struct.unpack_from("<8sIIQQQ", b, 0)
    assert magic == b"CDSVAR01" and file_bytes == len(b)
    index = np.frombuffer(b, dtype=np.dtype([("off", "<u8"), ("bytes", "<u4"), ("flags", "<u4")]), count=n, offset=index_off)

    def rle4(sec):
        rows, runs = sec[:H + 1], sec[H + 1:]
        img = np.zeros((H, W, 3), np.uint8)
        for y in range(H):
            r = runs[rows[y]:rows[y + 1]]
            length = (r >> 24).astype(np.int64)
            assert length.sum() == W and (length > 0).all()
            colour = np.stack([r & 255, (r >> 8) & 255, (r >> 16) & 255], 1).astype(np.uint8)
            img[y] = np.repeat(colour, length, axis=0)
        return img

    def spans(sec):   # per row the spans `x0 | len << 16` of its non-black pixels; their bytes follow, 3 per pixel
        row_span, row_pixel = sec[:H + 1], sec[H + 1:2 * H + 2]
        sp = sec[2 * H + 2:2 * H + 2 + row_span[H]]
        lit = sec[2 * H + 2 + row_span[H]:].view(np.uint8)[:3 * int(row_pixel[H])].reshape(-1, 3)
        img = np.zeros((H, W, 3), np.uint8)
        for y in range(H):
            at = int(row_pixel[y])
            for s_ in sp[row_span[y]:row_span[y + 1]]:
                x0, n = int(s_ & 0xffff), int(s_ >> 16)
                assert n > 0 and x0 + n <= W and (lit[at:at + n] != 0).any(axis=1).all()
                img[y, x0:x0 + n] = lit[at:at + n]
                at += n
            assert at == row_pixel[y + 1]
        return img

    out = []
    for e in index:
        assert e["off"] % 16 == 0
        blob = np.frombuffer(b, dtype="<u4", count=int(e["bytes"]) // 4, offset=int(e["off"]))
        flags, rb, gb, zb = (int(v) for v in blob[:4])
        assert flags == e["flags"] and 16 + rb + gb + zb == e["bytes"]
        rgb = (spans if flags & 16 else rle4)(blob[4:4 + rb // 4])
        grad = None
        if flags & 1:
            sec = blob[4 + rb // 4:4 + (rb + gb) // 4]
            if flags & 8:   # the non-zero second mixed differences, per row `x | e << 16`: the image is their 2-D prefix sum (mod 2^16)
                rows, ex = sec[:H + 1], sec[H + 1:]
                e = np.zeros((H, W), np.int64)
                for y in range(H):
                    r = ex[rows[y]:rows[y + 1]]
                    assert ((r & 0xffff) < W).all() and ((r >> 16) != 0).all() and (np.diff((r & 0xffff).astype(np.int64)) > 0).all()
                    e[y, r & 0xffff] = r >> 16
                grad = (np.cumsum(np.cumsum(e, axis=1), axis=0) & 0xffff).astype(np.uint16)
            elif flags & 2:   # two bits per pixel: 0 same, 1 up, 2 down, relative to the pixel on its left; word 0 of a row = its first value
                sec = sec.reshape(H, 1 + (W + 15) // 16)
                codes = ((sec[:, 1:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(H, -1)[:, :W].astype(np.int64)
                grad = ((sec[:, 0:1] & 0xffff).astype(np.int64) + np.cumsum((codes == 1).astype(np.int64) - (codes == 2), axis=1)).astype(np.uint16)
            else:
                sec = sec.reshape(H, (W + 1) // 2)
                grad = np.stack([sec & 0xffff, sec >> 16], 2).reshape(H, -1)[:, :W].astype(np.uint16)
        zgap = (spans if flags & 32 else rle4)(blob[4 + (rb + gb) // 4:]) if flags & 4 else None
        out.append((flags, rgb, grad, zgap))
    return W, H, out
