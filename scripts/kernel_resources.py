"""Register / LDS budget of every kernel in the built library, read from the code objects embedded in
anakin_amd/libsaber_mi355x.so (clang offload bundles -> gfx950 ELF -> NT_AMDGPU_METADATA msgpack note).
`.vgpr_count` there is the TOTAL (architectural + accumulation) registers a wave allocates, which is what sets the
occupancy; rocprofv3's kernel-trace table only carries the architectural count.

usage: python scripts/kernel_resources.py [lib.so]      -> one line per kernel
       from kernel_resources import load; load(path) -> {mangled name: dict(vgpr, sgpr, lds, wg)}"""
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _notes(elf):
    # ELF64 little endian: section headers -> SHT_NOTE
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        off = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, off + 4)
        if sh_type != 7:
            continue
        o, size = struct.unpack_from("<QQ", elf, off + 0x18)
        end = o + size
        while o + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, o)
            o += 12
            name = elf[o:o + namesz]
            o += (namesz + 3) & ~3
            desc = elf[o:o + descsz]
            o += (descsz + 3) & ~3
            yield name.rstrip(b"\0"), ntype, desc


def load(path=None):
    import msgpack
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anakin_amd", "libsaber_mi355x.so")
    blob = open(path, "rb").read()
    out = {}
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        o = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tsize].decode()
            o += 24 + tsize
            if "gfx950" not in triple or esize == 0:
                continue
            elf = blob[pos + eoff:pos + eoff + esize]
            for name, ntype, desc in _notes(elf):
                if name == b"AMDGPU" and ntype == 32:
                    md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in md.get("amdhsa.kernels", []):
                        out[k[".name"]] = dict(vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"],
                                               lds=k[".group_segment_fixed_size"], wg=k[".max_flat_workgroup_size"],
                                               spill=k.get(".vgpr_spill_count", 0))
        pos += len(MAGIC)
    return out


def workgroups_per_cu(r, wgsz=None, lds=None):
    """MI355X: 4 SIMDs x 512 registers per lane, 8 waves per SIMD, 160 KB LDS, 8-register granules."""
    total = (r["vgpr"] + 7) // 8 * 8
    waves = min(8, 512 // max(total, 8))
    wg_waves = max((wgsz or r["wg"]) // 64, 1)
    lds = r["lds"] if lds is None else lds
    return max(min(waves * 4 // wg_waves, (160 * 1024) // lds if lds else 32, 32), 0)


if __name__ == "__main__":
    res = load(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%5s %5s %7s %5s %6s  %s" % ("vgpr", "sgpr", "lds", "wg", "wg/CU", "kernel"))
    for k in sorted(res):
        r = res[k]
        print("%5d %5d %7d %5d %6d  %s" % (r["vgpr"], r["sgpr"], r["lds"], r["wg"], workgroups_per_cu(r), k))
