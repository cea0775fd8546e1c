#!/usr/bin/env python
"""Disassemble the gfx950 code objects inside libconvnet_hip.so and count, per kernel, the
instructions the design relies on (MFMA, LDS-DMA, transpose reads, non-temporal accesses).

Used as a build guard (tests/test_build_isa.py, __graft_entry__.build): round 1 shipped a binary in
which a run-time select had folded the non-temporal BatchNorm loads back into plain loads without
anybody noticing.  No GPU needed.

    python tools/isa_check.py [path/to/libconvnet_hip.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    """Yield (triple, bytes) for every device code object bundled in `path`."""
    blob = open(path, 'rb').read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx' in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def disassemble(path):
    """{kernel symbol: [instruction lines]} over all gfx950 code objects of the library."""
    out = {}
    for triple, data in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(data)
            name = f.name
        try:
            txt = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', name], capture_output=True, text=True,
                                 check=True).stdout
        finally:
            os.unlink(name)
        cur = None
        for line in txt.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
            if m:
                cur = m.group(1)
                out.setdefault(cur, [])
            elif cur is not None and line.strip():
                out[cur].append(line.split('//')[0].strip())
    return out


def demangle(names):
    try:
        r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True,
                           text=True, check=True).stdout.split('\n')
        return dict(zip(names, r))
    except Exception:
        return {n: n for n in names}


def counts(lines):
    c = {'mfma': 0, 'lds_dma': 0, 'tr_read': 0, 'nt_load16': 0, 'plain_load16': 0, 'nt_store16': 0,
         'plain_store16': 0}
    for l in lines:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('buffer_load') and l.rstrip().endswith('lds'):
            c['lds_dma'] += 1
        elif op.startswith('ds_read_b64_tr'):
            c['tr_read'] += 1
        elif op in ('global_load_dwordx4', 'flat_load_dwordx4'):
            c['nt_load16' if re.search(r'\bnt\b', l) else 'plain_load16'] += 1
        elif op in ('global_store_dwordx4', 'flat_store_dwordx4'):
            c['nt_store16' if re.search(r'\bnt\b', l) else 'plain_store16'] += 1
    return c


READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def kernel_resources(path):
    """{demangled kernel: {'lds': static LDS bytes, 'vgpr': VGPRs, 'agpr': AGPRs, 'sgpr': SGPRs, 'scratch': private bytes}}
    from the code objects' AMDGPU metadata notes: the footprint a resident workgroup takes from its CU - what a kernel of
    the other stream has to fit beside (profiles/README.md, round-4 weight-gradient A/Bs)."""
    res = {}
    for triple, data in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(data)
            name = f.name
        try:
            txt = subprocess.run([READELF, '--notes', name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(name)
        for blk in txt.split('- .agpr_count')[1:]:
            blk = '.agpr_count' + blk

            def field(key, b=blk):
                m = re.search(r'\.%s:\s+(\S+)' % key, b)
                return m.group(1) if m else None
            sym = field('name')
            if sym is None:
                continue
            res[sym] = {'lds': int(field('group_segment_fixed_size') or 0), 'vgpr': int(field('vgpr_count') or 0),
                        'agpr': int(field('agpr_count') or 0), 'sgpr': int(field('sgpr_count') or 0),
                        'scratch': int(field('private_segment_fixed_size') or 0)}
    dm = demangle(list(res))
    return {dm[k]: v for k, v in res.items()}


def kernel_table(path):
    dis = disassemble(path)
    dm = demangle(list(dis))
    return {dm[k]: counts(v) for k, v in dis.items()}


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = args[0] if args else os.path.join(here, 'convnet.pytorch_amd', 'libconvnet_hip.so')
    if '--resources' in sys.argv:
        res = kernel_resources(path)
        print('%-100s %8s %5s %5s %5s %7s' % ('kernel', 'LDS B', 'VGPR', 'AGPR', 'SGPR', 'scratch'))
        for name in sorted(res):
            r = res[name]
            print('%-100s %8d %5d %5d %5d %7d' % (name[:100], r['lds'], r['vgpr'], r['agpr'], r['sgpr'], r['scratch']))
        return
    tab = kernel_table(path)
    for name in sorted(tab):
        c = tab[name]
        if any(c.values()):
            print('%-110s %s' % (name[:110], ' '.join('%s=%d' % kv for kv in c.items() if kv[1])))


if __name__ == '__main__':
    main()
