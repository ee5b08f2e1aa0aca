"""Static check of the device code in liblp_hip.so (no GPU needed): no packed fp32 instruction whose LOW half takes the HIGH dword of a source.

Measured on MI355X in round 5 (scripts/pk_forms_probe.py, profiles/r05_pk_fp32_opsel_hazard.txt): `v_pk_fma_f32 ... op_sel:[0,1,0]`,
`v_pk_mul_f32 ... op_sel:[0,1]` and `v_pk_add_f32 ... op_sel:[0,1]` -- the forms hipcc picks for "vector times a scalar that sits in the high
half of a register pair" -- return a wrong low half in lanes 48..63 while the LDS-DMA convolution kernels run beside them on another stream
(hundreds of thousands of wrong results per launch; alone, or beside other kernels, they are exact).  That is what made two data-parallel
replicas drift apart: the spectral-norm power iteration (a W^T u accumulation of exactly that shape) dropped single terms whenever it overlapped
a conv of the other branch.  spectral_norm.hip and mobilenet.hip are compiled without packed fp32 arithmetic; this test keeps every op_sel:[...]
form of the packed fp32 instructions out of the whole library, whatever a future compiler or source change would like to emit.  The op_sel_hi
forms (HIGH half from a LOW dword), which the library uses thousands of times, were measured exact in the same probe."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'latent_pose_reenactment_amd', 'liblp_hip.so')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def _device_disassembly(tmp_path):
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    lib = shutil.copy(LIB, str(tmp_path / 'liblp_hip.so'))
    r = subprocess.run([OBJDUMP, '--offloading', lib], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    objs = sorted(f for f in os.listdir(tmp_path) if 'hipv4-amdgcn-amd-amdhsa--gfx950' in f)
    assert len(objs) >= 8, objs
    text = []
    for f in objs:
        d = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', f], cwd=str(tmp_path), capture_output=True, text=True)
        assert d.returncode == 0, d.stderr[-2000:]
        text.append(d.stdout)
    return '\n'.join(text)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_no_packed_fp32_low_half_from_a_high_dword(tmp_path):
    dis = _device_disassembly(tmp_path)
    kernel, bad, packed, hi_forms = None, [], 0, 0
    for line in dis.split('\n'):
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
        if m:
            kernel = m.group(1)
            continue
        ins = line.strip()
        if not re.match(r'v_pk_(fma|mul|add)_f32\b', ins):
            continue
        packed += 1
        hi_forms += 'op_sel_hi:[' in ins
        sel = re.search(r'op_sel:\[([0-9,]+)\]', ins)
        if sel and '1' in sel.group(1):
            bad.append(f'{kernel}: {ins.split("//")[0].strip()}')
    assert packed > 1000, f'only {packed} packed fp32 instructions found: did the disassembly work?'
    assert not bad, f'{len(bad)} packed fp32 instructions take a high source dword into their low half (wrong beside the conv kernels on gfx950): ' + '; '.join(bad[:8])


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_power_iteration_and_pose_encoder_kernels_use_no_packed_fp32(tmp_path):
    """the two files where hipcc had emitted the bad forms are built without packed fp32 arithmetic altogether"""
    dis = _device_disassembly(tmp_path)
    kernel, per = None, {}
    for line in dis.split('\n'):
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
        if m:
            kernel = m.group(1)
            per.setdefault(kernel, 0)
        elif kernel and re.match(r'\s*v_pk_(fma|mul|add)_f32\b', line):
            per[kernel] += 1
    sn = {k: v for k, v in per.items() if re.search(r'sn_(wtu|vsum|v|wv|u|dot|grad_apply|embed)', k)}
    mb = {k: v for k, v in per.items() if re.search(r'(stem_conv_s2|dwconv3x3|affine_res|affine_relu6)', k)}
    assert len(sn) >= 5 and len(mb) >= 3, (sorted(sn), sorted(mb))
    assert not any(sn.values()) and not any(mb.values()), {k: v for k, v in {**sn, **mb}.items() if v}
