"""Convert a 24-bit BMP (as rt_bmp_save_rgba writes it) to PNG with the standard library only; for eyeballing renders."""
import struct
import sys
import zlib


def convert(src, dst):
    b = open(src, "rb").read()
    off, = struct.unpack_from("<I", b, 10)
    w, h = struct.unpack_from("<ii", b, 18)
    stride = (w * 3 + 3) & ~3
    rows = []
    for y in range(h - 1, -1, -1):
        r = b[off + y * stride: off + y * stride + w * 3]
        rgb = bytearray(r)
        rgb[0::3], rgb[2::3] = r[2::3], r[0::3]
        rows.append(b"\0" + bytes(rgb))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")
    open(dst, "wb").write(png)


if __name__ == "__main__":
    convert(sys.argv[1], sys.argv[2])
