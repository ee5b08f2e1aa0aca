"""Static check of the tap-pipelined conv kernels (csrc/conv_pipe.hip).  Their fp16 / bf16 instantiations issue the MFMAs through inline asm
(accumulator tied in place: the register plan that lets two workgroups share a CU), and an asm statement is invisible to hipcc's hazard
recogniser (cdna_hip_programming.md 5.7): a VALU instruction that writes a register an MFMA reads as operand needs wait states between
the two that nobody would insert.  The test compiles the file to gfx950 assembly (no GPU needed) and walks every such kernel: no VALU /
v_accvgpr write to an A, B or C register of an asm MFMA within the two preceding instructions; no scratch (a spill reload is a VMEM
operation hipcc waits for with `vmcnt(0)`, which would drain the DMA pipeline); no compiler-inserted `s_waitcnt vmcnt(0)` inside the tap
loop beyond the one the kernel asks for itself at the last tap."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def test_conv_pipe_asm_mfma_operands_and_scratch(tmp_path):
    out = str(tmp_path / 'conv_pipe.s')
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-I',
                        os.path.join(ROOT, 'latent_pose_reenactment_amd', 'csrc'), '-S', '--cuda-device-only',
                        os.path.join(ROOT, 'latent_pose_reenactment_amd', 'csrc', 'conv_pipe.hip'), '-o', out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(out).read()
    kernels = re.findall(r'^(_Z16conv_pipe_kernel\w+):[^\n]*\n(.*?)\.Lfunc_end', text, flags=re.S | re.M)
    assert len(kernels) >= 12, len(kernels)
    rng = lambda a, b: set(range(int(a), int(b) + 1))
    checked = 0
    for name, body in kernels:
        meta = text[text.index(f'.amdhsa_kernel {name}'):]
        meta = meta[:meta.index('.end_amdhsa_kernel')]
        x3 = 'ELi1ELi' in name.split('Conv16Params')[0][-14:]          # template argument PREC = 1 (bf16x3): builtin MFMAs, not asm
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', '//'))]
        in_asm, prev, n_asm = False, [], 0
        for l in lines:
            if l.startswith(';'):
                if '#ASMSTART' in l:
                    in_asm = True
                elif '#ASMEND' in l:
                    in_asm = False
                continue
            if in_asm and l.startswith('v_mfma'):
                n_asm += 1
                regs = set()
                for a, b in re.findall(r'v\[(\d+):(\d+)\]', l):
                    regs |= rng(a, b)
                for p_ in prev[-2:]:
                    if p_.startswith(('v_', )) and not p_.startswith('v_mfma'):
                        dst = re.match(r'\S+\s+v\[(\d+):(\d+)\]|\S+\s+v(\d+)', p_)
                        if dst:
                            d = rng(dst.group(1), dst.group(2)) if dst.group(1) else {int(dst.group(3))}
                            assert not (d & regs), f'{name}: `{p_}` writes an operand of the asm MFMA `{l}` without wait states'
            if not in_asm or l.startswith('v_mfma'):
                prev.append(l)
        if not x3:
            assert n_asm >= 9 * 16, (name, n_asm)
            checked += 1
            priv = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', meta)
            assert priv and int(priv.group(1)) == 0, f'{name}: scratch in use ({priv.group(1) if priv else "?"} B): spill reloads would drain vmcnt'
            # between the first and the last MFMA: only the kernel's own vmcnt(0) of the final taps (one static occurrence)
            first = next(i for i, l in enumerate(lines) if l.startswith('v_mfma'))
            last = max(i for i, l in enumerate(lines) if l.startswith('v_mfma'))
            drains = [l for l in lines[first:last] if l.startswith('s_waitcnt') and 'vmcnt(0)' in l]
            assert len(drains) <= 1, (name, drains)
    assert checked >= 8, checked
