"""Builds dust3r_amd/csrc/libdust3r_hip.so for gfx950 with hipcc (in-tree, so the library travels
with the repository snapshot). Cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['gemm.hip', 'gemm_p4.hip', 'attention.hip', 'elementwise.hip', 'aligner.hip', 'bootstrap.hip', 'engine.hip', 'capi.hip']
HEADERS = ['common.hpp', 'kernels.hpp', 'aligner_math.hpp', os.path.join('..', '..', 'include', 'dust3r_hip.h')]
LIB = os.path.join(CSRC, 'libdust3r_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-inline-asm']
# attention.hip: no SLP vectorisation -- its scalar-VALU softmax slices (attention_x3_kernel<..., SC = true>) must stay v_fma_f32 / v_add_f32
# pairs; hipcc -O3 re-packs adjacent scalar fp32 operations into v_pk_* (an anti-lever beside MFMAs, MI355X_MICROARCH.md). The packed
# variants of the same kernel use explicit 2-vectors and are not affected.
EXTRA_FLAGS = {'attention.hip': ['-fno-slp-vectorize'], 'gemm_p4.hip': ['-fno-slp-vectorize', '-Rpass-analysis=kernel-resource-usage']}     # gemm_p4.hip: its drain's scalar GELU pieces sit between MFMAs too


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probes=None):
    """probes (default: the D3R_PROBES=1 environment switch, else off): compile with -DD3R_PROBES -- the ablation kernels and the probe-only environment
    switches of csrc/common.hpp (probe_env). The default library reads the twelve documented switches only."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    if probes is None:
        probes = os.environ.get('D3R_PROBES', '0') == '1'
    flags = FLAGS + (['-DD3R_PROBES'] if probes else []) + os.environ.get('D3R_BUILD_DEFINES', '').split()      # development: e.g. -DD3R_GEMM_ONLY_DT=3
    stamp = os.path.join(CSRC, '.build_flags')
    if not os.path.exists(stamp) or open(stamp).read() != ' '.join(flags):      # objects of the other flavour: rebuild everything
        force = True
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed:\n{r.stdout}\n{r.stderr}')
        if '-Rpass-analysis=kernel-resource-usage' in cmd:      # the register / scratch report of the kernels of this file, kept next to the object (tests/test_host_cpu.py reads it)
            with open(cmd[-1].replace('.o', '.resources.txt'), 'w') as f:
                f.write('\n'.join(ln for ln in r.stderr.splitlines() if 'remark:' in ln))
        return r

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    with open(stamp, 'w') as f:
        f.write(' '.join(flags))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
