"""ISA check for the asm-issued transposed LDS reads of the GEMM kernels ("the SPREAD = 2 anomaly").

`ds_read_b64_tr_b16` is issued through inline asm (gemm.hip, g16p_read_tr): the compiler does not know that the destination registers are
written LATER, when the LDS returns. Any instruction that touches such a register before an `s_waitcnt lgkmcnt(0)` — a `v_mov` the register
allocator places to resolve a PHI at a branch or loop back-edge is the case that was found — reads (or is overwritten by) data that may not
have landed. This script walks the gfx950 assembly of every kernel and reports, per kernel:
  * any instruction that names a still-pending destination register of a transposed read, and
  * any branch or label crossed while such registers are pending (the copy hazard lives at control-flow joins).
Usage: python tools/check_async_reads.py [file.s]   (without an argument it compiles exorl_amd/csrc/gemm.hip to assembly first)
Exit code 1 when a hazard is found.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_kernel(name, lines):
    """lines: instruction / label lines of one function, in program order. Returns a list of findings."""
    findings, pending = [], {}
    n_tr = 0
    for ln, raw, in_asm in lines:
        s = raw.split(';')[0].strip()
        if not s:
            continue
        if s.endswith(':'):                      # a label: a join point
            if pending:
                findings.append((ln, f'label {s} crossed with pending tr-read registers {sorted(pending)}'))
            continue
        op = s.split()[0]
        if op == 's_waitcnt' and 'lgkmcnt(0)' in s:
            pending.clear()
            continue
        if in_asm and (op.startswith('ds_read_b64_tr_b16') or op.startswith('ds_read_tr16_b64')):      # the builtin's reads are the compiler's to count
            n_tr += 1
            ops = s[len(op):].split(',')
            dst, rest = regs_of(ops[0]), regs_of(','.join(ops[1:]))
            hit = (dst | rest) & set(pending)
            if hit:
                findings.append((ln, f'`{s}` touches pending registers {sorted(hit)}'))
            for r in dst:
                pending[r] = ln
            continue
        if op.startswith('s_cbranch') or op == 's_branch' or op == 's_endpgm' or op == 's_setpc_b64':
            if pending:
                findings.append((ln, f'`{s}` crossed with pending tr-read registers {sorted(pending)}'))
            continue
        hit = regs_of(s) & set(pending)
        if hit:
            findings.append((ln, f'`{s}` touches registers {sorted(hit)} of a transposed read issued at line {min(pending[r] for r in hit)} before lgkmcnt(0)'))
    return n_tr, findings


def split_functions(path):
    funcs, cur, name, in_asm = {}, None, None, False
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                name, cur = m.group(1), []
                funcs[name] = cur
                continue
            if cur is None:
                continue
            if line.startswith('.Lfunc_end'):
                cur = None
                continue
            t = line.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not t or t.startswith(';') or (t.startswith('.') and not t.endswith(':')):
                continue
            cur.append((ln, t, in_asm))
    return funcs


def compile_to_asm(src, out, defines=()):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip', '--cuda-device-only', '-S', '-o', str(out), str(src),
           *[f'-D{d}' for d in defines]]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def demangle(names):
    try:
        r = subprocess.run(['c++filt'], input='\n'.join(names), text=True, stdout=subprocess.PIPE, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def run(path):
    funcs = split_functions(path)
    dm = demangle(list(funcs))
    bad = 0
    checked = 0
    for name, lines in funcs.items():
        n_tr, findings = check_kernel(name, lines)
        if n_tr == 0:
            continue
        checked += 1
        if findings:
            bad += 1
            print(f'HAZARD {dm[name][:150]}: {len(findings)} finding(s), {n_tr} transposed reads')
            for ln, msg in findings[:6]:
                print(f'    {path}:{ln}: {msg}')
    print(f'{checked} kernels with asm-issued transposed reads checked, {bad} with hazards')
    return bad


if __name__ == '__main__':
    if len(sys.argv) > 1 and not sys.argv[1].startswith('-'):
        sys.exit(1 if run(sys.argv[1]) else 0)
    defines = [a[2:] for a in sys.argv[1:] if a.startswith('-D')]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'gemm.s'
        compile_to_asm(ROOT / 'exorl_amd' / 'csrc' / 'gemm.hip', out, defines)
        sys.exit(1 if run(out) else 0)
