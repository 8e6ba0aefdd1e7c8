"""Properties of the instruction streams hipcc emits for the two kernels that are two thirds of a step (CPU only: hipcc cross-compiles
gfx950 here).  Round 3 found the fused halo conv's GroupNorm + SiLU arithmetic in one 70-150-instruction lump per tap with the matrix
pipe idle, its fragment reads behind the MFMAs they were meant to precede, a conservative vmcnt behind a run-time branch, and the first
global load of both kernels hundreds of instructions into the set-up code -- none of it visible in the source, all of it the compiler's
placement of register-only code.  These checks fail when a source edit or a compiler update brings such a stream back."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'rq-vae-transformer_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import isa_summary  # noqa: E402


def _asm(src, tmp_path_factory, extra=()):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    out = str(tmp_path_factory.mktemp('isa') / (src + '.s'))
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', CSRC, *extra, '-S', '--cuda-device-only',
                           os.path.join(CSRC, src), '-o', out], stderr=subprocess.DEVNULL)
    return isa_summary.kernels(out)


@pytest.fixture(scope='module')
def conv_kernels(tmp_path_factory):
    return _asm('conv_halo.hip', tmp_path_factory)


@pytest.fixture(scope='module')
def conv_kernels_f16(tmp_path_factory):
    """the same source in the fp16 build of the library (-DRQ_F16=1: librqamd_f16.so, the opt-in fp16 RQ-VAE engine)"""
    return _asm('conv_halo.hip', tmp_path_factory, ('-DRQ_F16=1',))


@pytest.fixture(scope='module')
def gemm_kernels(tmp_path_factory):
    return _asm('gemm.hip', tmp_path_factory)


def _taps(instrs):
    """barrier intervals that are taps of the 3x3 stencil: exactly 16 MFMAs"""
    return [iv for iv in isa_summary.intervals(instrs) if iv.count('M') == 16]


@pytest.mark.parametrize('name', ['conv3x3_halo_kernel<1, 0, 0>', 'conv3x3_halo_kernel<1, 0, 1>'])
def test_fused_halo_conv_schedule(conv_kernels, name):
    ins = conv_kernels[name]
    taps = _taps(ins)
    assert len(taps) >= 14, len(taps)                      # 9 taps of the chunk loop + 9 of the last chunk (a few share an interval with loop glue)
    lumps = []
    for iv in taps:
        # the longest run of VALU / transcendental / SALU instructions with no MFMA in it, counted from the tap's first MFMA to its last
        body = iv[iv.index('M'):iv.rindex('M') + 1]
        lumps.append(max(len(r) for r in re.split('M', body)))
        # fragment reads precede the MFMAs of the previous k-step: never eight reads back to back followed by a wait
        assert 'rrrrrrrr_' not in iv, iv
    # a piece-carrying tap deals ~4 VALU + 1 transcendental out per MFMA (longest gap seen: 12); before the fix 70-150 sat in one gap
    # (round 5: the gap that carries the weight unit's two LDS-DMAs holds their M0 / base-address scalar instructions as well: 25)
    assert max(lumps) <= 32, lumps
    # some taps do carry GroupNorm work (transcendentals between MFMAs): the fusion is still there
    assert sum('t' in iv[iv.index('M'):iv.rindex('M')] for iv in taps) >= 5
    # the first global load comes early (it was instruction ~310: the second half of the workgroup reached it ~6000 cycles late)
    first = next(i for i, t in enumerate(ins) if t.startswith('global_load'))
    assert first < 200, first


def _tap_memory_ops(ins):
    """per barrier interval with exactly 16 MFMAs (a tap): the vector-memory loads and vmcnt waits in issue order --
    'D' LDS-DMA, 'G' ordinary load, ('w', N) s_waitcnt vmcnt(N)"""
    taps, cur, mf = [], [], 0
    for t in ins:
        op = t.split()[0]
        if op.startswith('global_load_lds'):
            cur.append('D')
        elif op.startswith('global_load'):
            cur.append('G')
        elif op.startswith('v_mfma'):
            mf += 1
            if mf == 1:
                cur.append('M')                      # (where the tap's first MFMA stands)
        elif op.startswith('s_waitcnt') and 'vmcnt' in t:
            cur.append(('w', int(re.search(r'vmcnt\((\d+)\)', t).group(1))))
        elif op.startswith('s_barrier'):
            taps.append(cur if mf == 16 else None)
            cur, mf = [], 0
    return taps


@pytest.mark.parametrize('name', ['conv3x3_halo_kernel<1, 0, 0>', 'conv3x3_halo_kernel<1, 0, 1>', 'conv3x3_halo_kernel<0, 0, 0>',
                                  'conv3x3_halo_kernel<0, 0, 1>', 'conv3x3_halo_kernel<0, 1, 0>'])
def test_halo_conv_weight_dma_waits_are_exact(conv_kernels, name):
    """The per-tile halo conv's weight units arrive by LDS-DMA (round 5) behind hand-counted waits, in a kernel whose patch / residual /
    GroupNorm loads the compiler counts on its own -- without knowing about the DMAs.  The counter retires in issue order, so the wait
    before the barrier that ends a tap must leave in flight exactly: this tap's two DMAs + every ordinary load issued since LAST tap's
    DMAs.  One too many and a wavefront passes the barrier with its part of the next unit still on the way (silent garbage on the GPU;
    the host emulator counts DMAs only); fewer and the wait also stands on loads that come from HBM.  Checked here on the instruction
    stream itself: a source edit that moves a load across a DMA, or a compiler that merges / splits / moves one, fails this."""
    _check_weight_dma_waits(conv_kernels, name)


@pytest.mark.parametrize('name', ['conv3x3_halo_kernel<1, 0, 0>', 'conv3x3_halo_kernel<1, 0, 1>', 'conv3x3_halo_kernel<0, 0, 0>',
                                  'conv3x3_halo_kernel<0, 0, 1>', 'conv3x3_halo_kernel<0, 1, 0>'])
def test_halo_conv_weight_dma_waits_are_exact_in_the_fp16_build(conv_kernels_f16, name):
    """The same property of the instruction stream hipcc emits for the fp16 build (other conversion instructions around the same loads:
    the hand-counted waits must fit that stream too)."""
    _check_weight_dma_waits(conv_kernels_f16, name)


def _check_weight_dma_waits(conv_kernels, name):
    taps = _tap_memory_ops(conv_kernels[name])
    checked = 0
    for i, ops in enumerate(taps):
        if ops is None:
            continue
        dmas = [k for k, o in enumerate(ops) if o == 'D']
        assert len(dmas) in (0, 2), (i, ops)
        if dmas:
            assert ops.index('M') < dmas[0], (i, ops)               # requested behind the tap's first k-step, not in front of its MFMAs
        waits = [o for o in ops if isinstance(o, tuple)]
        prev = taps[i - 1] if i > 0 else None
        prev_dma_expected = prev is not None and 'D' in prev
        if not prev_dma_expected:
            continue                                               # (first tap of the loop body / after the last request: nothing to wait for, or checked below)
        # ordinary loads younger than the previous tap's DMAs: behind them in that tap, and all of this tap's
        prev_after = sum(1 for o in prev[len(prev) - prev[::-1].index('D'):] if o == 'G')
        mine = sum(1 for o in ops if o == 'G')
        assert waits, (i, ops)
        assert waits[-1][1] == len(dmas) + prev_after + mine, (name, i, prev, ops)
        checked += 1
    assert checked >= 12, checked
    assert 'conv3x3_halo_kernel<1, 0, 0>' in conv_kernels and 'conv3x3_halo_kernel<0, 0, 1>' in conv_kernels


def test_halo_conv_register_budget():
    """The per-tile halo conv after the LDS-DMA weight ring: no scratch, and the fused forms well under the 256 registers at which the round-3
    attempts at deeper pipelining spilled (242 before the ring, 212 with it).  A change that brings the staging registers back shows up here."""
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    err = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', CSRC, '-c', os.path.join(CSRC, 'conv_halo.hip'), '-o', os.devnull,
                          '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
    cur, seen = None, {}
    for line in err.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            seen[cur] = {}
            continue
        for key in ('VGPRs', 'ScratchSize \\[bytes/lane\\]'):
            m = re.search(r'remark:\s+' + key + r': (\d+)', line)
            if m and cur:
                seen[cur][key[:7]] = int(m.group(1))
    halo = {k: v for k, v in seen.items() if 'conv3x3_halo_kernel' in k}
    assert len(halo) >= 5, list(seen)
    for k, v in halo.items():
        assert v.get('Scratch', 0) == 0, (k, v)
        assert v['VGPRs'] <= 224, (k, v)


def test_gemm_p8_schedule(gemm_kernels):
    ins = gemm_kernels['gemm_p8_kernel<1, 2, 0>']
    # K-tile 0 is requested ahead of the fragment addresses and the accumulator init (it was instruction ~500)
    first = next(i for i, t in enumerate(ins) if t.startswith('global_load_lds'))
    assert first < 420, first
    # the epilogue family is a compile-time constant: the bf16 kernel carries no GELU polynomial and is a fraction of the
    # 22 000-line run-time form
    assert len(ins) < 4000, len(ins)
    # main loop body (between the loop header and its back edge): branch-free, 32 MFMAs, the staging of the next K-tile inside
    text = '\n'.join(ins)
    # find the densest 32-MFMA window without a branch: the peeled loop body
    idx = [i for i, t in enumerate(ins) if t.startswith('v_mfma')]
    ok = False
    for a in range(0, len(idx) - 31):
        lo, hi = idx[a], idx[a + 31]
        window = ins[lo:hi + 1]
        if not any(t.startswith('s_cbranch') or t.startswith('s_branch') for t in window) and sum(t.startswith('global_load_lds') for t in window) >= 2:
            ok = True
            break
    assert ok, 'no branch-free K-tile body with its staging found'
    # every shipped family exists as its own kernel
    for ek in (0, 1, 4, 5, 6):
        assert f'gemm_p8_kernel<1, 2, {ek}>' in gemm_kernels, ek
    assert 'gemm_p8_kernel<0, 4, 3>' in gemm_kernels


def test_conv_out_reads_run_ahead_of_their_mfma(conv_kernels):
    """Decoder.conv_out: 36 MFMAs per chunk on one accumulator, both operands out of LDS.  Written as read, read, MFMA every MFMA waited for
    its own reads (`s_waitcnt lgkmcnt(0)` 72 times per tile); with the fragments of step s + 4 read ahead of the MFMA of step s the waits
    are counted ones that leave the younger reads in flight."""
    ins = conv_kernels['conv_out_halo_kernel<1>']
    idx = [i for i, t in enumerate(ins) if t.startswith('v_mfma')]
    assert len(idx) == 36, len(idx)                        # the chunk loop is not unrolled: 9 taps x 4 k-steps
    waits = [int(m.group(1)) for t in ins[idx[0]:idx[-1] + 1] for m in [re.match(r's_waitcnt lgkmcnt\((\d+)\)', t)] if m]
    assert len(waits) >= 30 and sorted(waits)[len(waits) // 2] >= 4, waits
