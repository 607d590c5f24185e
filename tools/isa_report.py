# -*- coding: utf-8 -*-
"""Static report of the compiled kernels (no GPU needed): compiles every translation unit of
ssqueezepy_amd/csrc for gfx950 to assembly and prints, per kernel, registers, scratch (spills),
static LDS, code size and the instruction mix.
    python tools/isa_report.py [substring-filter] > profiles/rNN_isa_report.txt"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ssqueezepy_amd', 'csrc')
HIPCC = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'bin', 'hipcc')
sys.path.insert(0, ROOT)
from ssqueezepy_amd.build import SOURCES, COMMON       # noqa: E402


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.split('\n')


def main(flt=''):
    rows = []
    for src, extra in SOURCES:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, 'k.s')
            cmd = [HIPCC] + [c for c in COMMON if c != '-fPIC'] + extra + [
                '-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', asm]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-2000:]); raise SystemExit(1)
            text = open(asm).read()
        meta = {}
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
            name, body = m.group(1), m.group(2)
            get = lambda k: int(re.search(r'\.%s (\d+)' % k, body).group(1)) if re.search(r'\.%s (\d+)' % k, body) else 0
            meta[name] = dict(ldsb=get('amdhsa_group_segment_fixed_size'), scratch=get('amdhsa_private_segment_fixed_size'))
        for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', text):
            if m.group(1) in meta:
                meta[m.group(1)].update(sgpr=int(m.group(2)), vgpr=int(m.group(3)))
        for name in meta:
            b = re.search(r'^%s:[^\n]*\n(.*?)s_endpgm' % re.escape(name), text, re.S | re.M)
            ins = re.findall(r'^\s+([a-z][a-z0-9_]+)', b.group(1), re.M) if b else []
            mix = dict(valu=sum(i.startswith('v_') for i in ins), salu=sum(i.startswith('s_') for i in ins),
                       lds=sum(i.startswith('ds_') for i in ins),
                       vmem=sum(i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) for i in ins))
            meta[name].update(mix, n=len(ins))
            rows.append((src, name, meta[name]))
    names = demangle([r[1] for r in rows])
    print('%-22s %5s %5s %7s %7s %7s | %6s %6s %5s %5s  kernel' %
          ('file', 'vgpr', 'sgpr', 'scratch', 'ldsB', 'instrs', 'valu', 'salu', 'lds', 'vmem'))
    for (src, name, d), dn in zip(rows, names):
        short = re.sub(r'\(.*', '', dn).replace('ssq::', '')
        if flt and flt not in short:
            continue
        print('%-22s %5d %5d %7d %7d %7d | %6d %6d %5d %5d  %s' %
              (src, d.get('vgpr', 0), d.get('sgpr', 0), d['scratch'], d['ldsb'], d.get('n', 0),
               d.get('valu', 0), d.get('salu', 0), d.get('lds', 0), d.get('vmem', 0), short[:110]))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '')
