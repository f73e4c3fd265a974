"""Did the default K1 instantiation change between two commits?  (round-3 verdict W4: 0.832 -> 0.808 of peak with one more
kernel argument.)  CPU box only (needs .git and hipcc):  python tools/k1_isa_diff.py f302264 HEAD
Compiles nemoflux_amd/csrc/nf_flux.hip of both commits to gfx950 assembly with the product's flags, extracts
nf::k_flux<double, 2, 10, true, 256, 1, 0> and compares instruction stream, register counts and kernel descriptor."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only']
PREFIX = '_ZN2nf6k_fluxIdLi2ELi10ELb1ELi256ELi1ELi0E'     # nf::k_flux<double, 2, 10, true, 256, 1, 0>
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(commit, tmp):
    d = os.path.join(tmp, commit.replace('/', '_'))
    for rel in ('nemoflux_amd/csrc/nf_flux.hip', 'nemoflux_amd/csrc/nf_common.h', 'include/nemoflux_amd.h'):
        os.makedirs(os.path.dirname(os.path.join(d, rel)), exist_ok=True)
        with open(os.path.join(d, rel), 'wb') as f:
            f.write(subprocess.check_output(['git', '-C', ROOT, 'show', f'{commit}:{rel}']))
    out = os.path.join(d, 'nf_flux.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', out, 'nf_flux.hip'], cwd=os.path.join(d, 'nemoflux_amd/csrc'),
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def default_kernel(txt):
    m = re.search(r'^(' + PREFIX + r'[^:\n]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M)
    name, body = m.group(1), m.group(2)
    ins = [re.sub(r'\s*;.*$', '', l.strip()) for l in body.splitlines()
           if l.strip() and not l.strip().startswith((';', '.', '//'))]
    desc = dict(re.findall(r'\.amdhsa_(\w+) (\S+)', re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n(.*?)\.end_amdhsa_kernel',
                                                                txt, re.S).group(1)))
    regs = {k: int(re.search(re.escape(name) + r'\.' + k + r', (\d+)', txt).group(1)) for k in ('num_vgpr', 'num_agpr', 'numbered_sgpr')}
    return name, ins, desc, regs


def main(a, b):
    with tempfile.TemporaryDirectory() as tmp:
        ka, kb = default_kernel(asm_of(a, tmp)), default_kernel(asm_of(b, tmp))
    for tag, (name, ins, desc, regs) in ((a, ka), (b, kb)):
        c = Counter(i.split()[0] for i in ins)
        print(f'{tag}: {name}')
        print(f'   {len(ins)} instructions, {regs}, kernarg {desc.get("kernarg_size")} B, scratch {desc.get("private_segment_fixed_size")}, '
              f'global_load_dwordx4 {c["global_load_dwordx4"]}, global_store_dwordx4 {c["global_store_dwordx4"]}, '
              f'global_store_dwordx2 {c["global_store_dwordx2"]}, v_fmac_f64 {c["v_fmac_f64_e32"]}, s_waitcnt {c["s_waitcnt"]}')
    strip = lambda ins: [re.sub(r'\.LBB\d+_', '.LBB_', i) for i in ins]      # labels are numbered per function in the file
    sa, sb = strip(ka[1]), strip(kb[1])
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(sa, sb)) if x != y]
    print(f'instruction-by-instruction (registers included, labels renumbered): {len(diff)} of {len(sa)} differ' +
          (', lengths differ' if len(sa) != len(sb) else ''))
    for i, x, y in diff[:20]:
        print(f'   [{i}]  {a}: {x}    {b}: {y}')
    dd = {k: (ka[2].get(k), kb[2].get(k)) for k in sorted(set(ka[2]) | set(kb[2])) if ka[2].get(k) != kb[2].get(k)}
    print('kernel descriptor fields that differ:', dd or 'none')


if __name__ == '__main__':
    main(*(sys.argv[1:3] if len(sys.argv) >= 3 else ('f302264', 'HEAD')))
