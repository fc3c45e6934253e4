"""Resource usage of every kernel in the built library, read from the gfx950 code objects inside ``libpqdet_hip.so``.

Used by ``tests/test_codeobj.py`` (no GPU needed): no ``pq::`` kernel may use scratch (private segment) memory or spill
vector registers.  The fat binary section of the linked library is a concatenation of clang offload bundles (one per
object file); each bundle is parsed by hand (magic, entry table), the device ELF is handed to ``llvm-readelf --notes``
and the AMDGPU metadata it prints is read line by line.
"""
import os
import re
import struct
import subprocess
import tempfile

LLVM_BIN = os.environ.get('PQ_LLVM_BIN', '/opt/rocm/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _section(path, name):
    """Raw bytes of ELF section `name` of the shared object at `path` (64-bit little-endian ELF)."""
    with open(path, 'rb') as f:
        data = f.read()
    assert data[:4] == b'\x7fELF' and data[4] == 2, 'not a 64-bit ELF: %s' % path
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    def sh(i):
        return struct.unpack_from('<IIQQQQIIQQ', data, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        s = sh(i)
        nm = data[stroff + s[0]:data.index(b'\0', stroff + s[0])].decode()
        if nm == name:
            return data[s[4]:s[4] + s[5]]
    raise KeyError(name)


def device_elfs(lib_path, arch='gfx950'):
    """Device code objects (bytes) for `arch` of every bundle in the library's .hip_fatbin section."""
    fat = _section(lib_path, '.hip_fatbin')
    out = []
    pos = fat.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from('<Q', fat, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', fat, q)
            triple = fat[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos = fat.find(MAGIC, pos + len(MAGIC))
    return out


_KEYS = ('.name', '.private_segment_fixed_size', '.vgpr_spill_count', '.sgpr_spill_count', '.vgpr_count', '.agpr_count',
         '.sgpr_count', '.group_segment_fixed_size', '.uses_dynamic_stack')


def kernels(lib_path, arch='gfx950'):
    """[{name, private_segment_fixed_size, vgpr_spill_count, ...}] of every kernel in the library."""
    readelf = os.path.join(LLVM_BIN, 'llvm-readelf')
    if not os.path.exists(readelf):
        raise FileNotFoundError(readelf)
    res = []
    for elf in device_elfs(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([readelf, '--notes', f.name], check=True, capture_output=True, text=True).stdout
        cur = None
        in_kernels = False
        for line in txt.splitlines():
            if line.startswith('amdhsa.kernels:'):
                in_kernels = True
                continue
            if in_kernels and re.match(r'^[A-Za-z]', line):          # next top-level key
                in_kernels = False
            if not in_kernels:
                continue
            m = re.match(r'^  - (\.[a-z_]+): +(.*)$', line)            # first key of a kernel entry
            if m:
                cur = {}
                res.append(cur)
            else:
                m = re.match(r'^    (\.[a-z_]+): +(.*)$', line)
            if m and cur is not None and m.group(1) in _KEYS:
                v = m.group(2).strip().strip("'\"")
                cur[m.group(1)[1:]] = v if m.group(1) == '.name' else (v == 'true' if v in ('true', 'false') else int(v))
    return [k for k in res if 'name' in k]


def demangle(names):
    filt = os.path.join(LLVM_BIN, 'llvm-cxxfilt')
    if not os.path.exists(filt):
        return list(names)
    out = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


if __name__ == '__main__':
    import sys
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libpqdet_hip.so')
    ks = kernels(lib)
    names = demangle([k['name'] for k in ks])
    print('%d kernels' % len(ks))
    for k, n in sorted(zip(ks, names), key=lambda kn: -(kn[0].get('vgpr_count', 0) + kn[0].get('agpr_count', 0))):
        flag = '  <-- SCRATCH' if k.get('private_segment_fixed_size', 0) or k.get('vgpr_spill_count', 0) else ''
        print('%4d v %3d a %3d s  lds %6d  scratch %4d  spill v%d s%d  %s%s' % (
            k.get('vgpr_count', 0), k.get('agpr_count', 0), k.get('sgpr_count', 0), k.get('group_segment_fixed_size', 0),
            k.get('private_segment_fixed_size', 0), k.get('vgpr_spill_count', 0), k.get('sgpr_spill_count', 0), n[:150], flag))
