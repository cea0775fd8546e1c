"""Build guard: the shipped gfx950 code objects contain the instructions the design claims
(tools/isa_check.py disassembles libconvnet_hip.so; no GPU needed).  Round 1 shipped a library whose
"non-temporal" BatchNorm loads had been folded into plain loads by a run-time select; this test pins
the cache policy, the MFMA / LDS-DMA / transpose-read core of the GEMM kernels and the absence of
anything but our own kernels."""
import os
import sys

import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIP_LIB = os.path.join(ROOT, 'convnet.pytorch_amd', 'libconvnet_hip.so')


@pytest.fixture(scope='module')
def table():
    import isa_check
    if not os.path.exists(HIP_LIB):
        import __graft_entry__ as g
        g.build()
    return isa_check.kernel_table(HIP_LIB)


def _find_any(table, *needles):
    return [(k, v) for k, v in table.items() if all(n in k for n in needles)]


def _find(table, *needles):
    hits = [(k, v) for k, v in table.items() if all(n in k for n in needles)]
    assert hits, 'no kernel matching %r in the library' % (needles,)
    return hits


def test_bn_streaming_passes_really_use_nontemporal_accesses(table):
    # NT = true instantiations: the activation loads are `nt`, the plain ones left are coefficient loads
    # (the reduction pass exists with cached loads only since round 4: non-temporal loads there measured slower)
    # (leading space: the BatchNorm kernels proper, not config 5's rangebn_* kernels, whose trailing template booleans
    # since round 6 say "operand stored as 8-bit levels")
    assert not _find_any(table, ' bn_bwd_reduce_kernel<', ', true>')
    for kern, sites in ((' bn_apply_kernel', 'nt_load16'), (' bn_bwd_apply_kernel', 'nt_load16'),
                        (' bn_bwd_apply_kernel', 'nt_store16')):
        # (bn_apply_kernel<T, NT, DUAL>: the cache policy is its second parameter; the others end with it)
        on, off = ((', true, ', ', false, ') if kern == ' bn_apply_kernel' else (', true>', ', false>'))
        for name, c in _find(table, kern + '<', on):
            assert c[sites] > 0, (name, c)
        for name, c in _find(table, kern + '<', off):
            assert c['nt_load16'] == 0 and c['nt_store16'] == 0, (name, c)
    # z is stored with the default policy in both instantiations (its consumer follows at once)
    for name, c in _find(table, ' bn_apply_kernel'):
        assert c['nt_store16'] == 0 and c['plain_store16'] > 0, (name, c)


def test_gemm_kernels_are_mfma_lds_dma_and_transpose_reads(table):
    ig = _find(table, 'igemm_kernel', 'bf16_t')
    assert all(c['mfma'] >= 8 for _, c in ig), [(n, c['mfma']) for n, c in ig]
    assert any(c['lds_dma'] > 0 for _, c in ig)
    wg = _find(table, 'wgrad', 'kernel')
    assert any(c['tr_read'] > 0 and c['mfma'] > 0 for _, c in wg)
    assert any(c['lds_dma'] > 0 for _, c in wg)


def test_round3_kernels_are_what_they_claim(table):
    """The band weight-gradient kernel is MFMA + transpose reads + LDS-DMA (dy) with no scratch spill path visible as
    plain 16-byte global stores only in its epilogue; the interleaved-issue GEMM instantiations carry their LDS-DMA
    instructions; the lazy-dy weight-gradient instantiation exists next to the plain one."""
    for name, c in _find(table, 'wgrad3x3_kernel', 'bf16_t'):
        assert c['mfma'] >= 9 and c['tr_read'] >= 20 and c['lds_dma'] > 0, (name, c)
    ilv = _find(table, 'igemm_kernel<bf16_t', ', true>(IgemmParams)')       # last template argument: ILV
    assert all(c['lds_dma'] >= 8 and c['mfma'] >= 16 for _, c in ilv), ilv
    lazy = _find(table, 'wgrad_kernel<bf16_t', ', 1>(WgradParams)') + _find(table, 'wgrad_kernel<bf16_t', ', 2>(WgradParams)')
    plain = _find(table, 'wgrad_kernel<bf16_t', ', 0>(WgradParams)')      # last template argument: LAZY (0 / 1 / 2)
    assert lazy and plain and all(c['mfma'] >= 4 and c['tr_read'] >= 4 for _, c in lazy + plain)
    # round 3, second half: the streaming junction kernels and the halo kernels are MFMA kernels whose filter operand
    # never passes through LDS-DMA, with 16-byte global stores; the junction pair and the stem weight gradient use the
    # LDS transpose read
    for kern in ('jdgrad_kernel<bf16_t', 'stem_fwd_kernel<bf16_t', 'conv3x3_c64_kernel<bf16_t'):
        for name, c in _find(table, kern):
            assert c['mfma'] >= 4 and c['lds_dma'] == 0 and c['plain_store16'] > 0, (name, c)
    for kern in ('jbwd_kernel<bf16_t', 'stem_wgrad_kernel<bf16_t'):
        for name, c in _find(table, kern):
            assert c['mfma'] >= 2 and c['tr_read'] >= 4, (name, c)


def test_workgroup_footprints_of_the_two_stream_schedule():
    """Round 4: what the backward chain pays for is the FOOTPRINT a weight-gradient workgroup takes from its CU (LDS,
    registers), not the side kernel's duration (profiles/README.md, round-4 A/Bs).  The figures the dispatch rules and
    DESIGN.md quote, read from the code objects' metadata: the register-staged 64 x 128 weight-gradient tile is the
    small one (32 KB; the LDS-DMA tile it replaced for <= 64 output channels held 48 KB), the 128-wide LDS-DMA kernel
    holds 64 KB (two per CU), no kernel exceeds the CU's 160 KB, and no bf16 kernel of the headline step spills more
    than a few set-up registers."""
    import isa_check
    res = isa_check.kernel_resources(HIP_LIB)

    def one(*needles):
        hits = [(k, v) for k, v in res.items() if all(n in k for n in needles)]
        assert len(hits) == 1, (needles, [k for k, _ in hits])
        return hits[0][1]
    assert one('wgrad_kernel<bf16_t, 64, 128, 0>')['lds'] <= 32 * 1024
    assert one('wgrad_kernel<bf16_t, 64, 128, 0>')['vgpr'] + one('wgrad_kernel<bf16_t, 64, 128, 0>')['agpr'] <= 136
    assert one('wgrad_dma_kernel<128>')['lds'] == 64 * 1024
    assert one('wgrad3x3_kernel<bf16_t, 128>')['lds'] <= 80 * 1024       # two per CU
    assert one('jbwd_kernel<bf16_t')['lds'] <= 160 * 1024
    assert not any('wgrad_dma_kernel<64>' in k for k in res)             # removed from the dispatch and from the library
    assert all(v['lds'] <= 160 * 1024 for v in res.values())
    for k, v in res.items():
        if 'bf16_t' in k or 'wgrad_dma' in k:
            assert v['scratch'] <= 16, (k, v)                            # bytes per lane; 0 for all but two set-up spills
