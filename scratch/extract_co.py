"""Extract the gfx950 code object embedded in libsmcp_amd.so (clang offload bundle in .hip_fatbin): python3 scratch/extract_co.py out.co"""
import struct, sys
data = open("smcp_amd/libsmcp_amd.so", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(magic)
n, = struct.unpack_from("<Q", data, pos + len(magic))
off = pos + len(magic) + 8
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off)
    triple = data[off + 24: off + 24 + tl].decode()
    off += 24 + tl
    if "gfx950" in triple:
        open(sys.argv[1], "wb").write(data[pos + o: pos + o + sz])
        print(triple, sz)
