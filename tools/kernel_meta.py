#!/usr/bin/env python3
"""Per-kernel resource metadata of libishap_hip.so without a GPU: scratch bytes per lane, VGPRs, SGPRs, LDS, kernarg size.

Reads the gfx950 code objects out of the shared library's `.hip_fatbin` section (clang offload bundles, one per
translation unit) and their NT_AMDGPU_METADATA notes (msgpack).  Why it exists (round 6): removing three unused ints from
`IgemmArgs` moved {alpha, out_mode, stat_out} onto a 16-byte boundary; the compiler then kept those four dwords in a PRIVATE
copy of the argument block -- an s_load + s_waitcnt + scratch_store at kernel entry and a scratch-enabled dispatch: every
LDS-DMA convolution launch got 0.5-1.1 us slower (+3 % per edit, profiles/round6_ab_prune_scratch.txt) with no warning
anywhere.  tests/test_host_cpu.py holds the hot kernels to zero scratch with this module.

    python tools/kernel_meta.py [path/to/libishap_hip.so] [name substring]
"""
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob: bytes):
    """every amdgcn ELF image inside the concatenated offload bundles of `blob`"""
    pos = 0
    while True:
        b = blob.find(MAGIC, pos)
        if b < 0:
            return
        n, = struct.unpack_from("<Q", blob, b + len(MAGIC))
        p = b + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size > 0:
                yield triple, blob[b + off:b + off + size]
        pos = p


def _notes(elf: bytes):
    """(name, type, desc) of every note in the ELF64 image (section headers)"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "ELF64 image expected"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:                 # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0").decode()
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(lib_path: str):
    """{kernel symbol: metadata dict} over every gfx950 code object of the library"""
    import msgpack
    blob = open(lib_path, "rb").read()
    out = {}
    for triple, elf in _code_objects(blob):
        if "gfx950" not in triple:
            continue
        for name, ntype, desc in _notes(elf):
            if name == "AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out


def demangle(names):
    import subprocess
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(root, "ishapediting_amd", "libishap_hip.so")
    pat = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    ks = kernels(lib)
    dm = demangle(sorted(ks))
    print(f"{len(ks)} kernels in {lib} ({os.path.getsize(lib)} bytes)")
    print(f"{'scratch':>8s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'kernarg':>8s}  kernel")
    for n in sorted(ks, key=lambda n: dm[n]):
        k = ks[n]
        short = dm[n].split("(")[0].replace("void ", "")
        if pat in short:
            print(f"{k.get('.private_segment_fixed_size', 0):8d} {k.get('.vgpr_count', 0):5d} {k.get('.agpr_count', 0):5d} {k.get('.sgpr_count', 0):5d} "
                  f"{k.get('.group_segment_fixed_size', 0):7d} {k.get('.kernarg_segment_size', 0):8d}  {short}")


if __name__ == "__main__":
    main()
