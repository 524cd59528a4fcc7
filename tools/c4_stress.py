"""Stress harness for the aligner's create -> first evaluation path at BASELINE configs[3] size (20 views, 190 edges, 384 x 512).

Why: round 2 once read a loss 1.3 % off in tests/test_aligner_gpu.py::test_c4_loss_and_gradients_match_fp64 (0.38860 vs 0.38368) as the
first GPU process on a fresh box, never reproduced. This tool hunts it:

  python tools/c4_stress.py prepare            build the scene on the CPU ONCE and cache it (the fixture's loss pins the inputs)
  python tools/c4_stress.py child <mode>       ONE fresh process: load the cached scene, build the aligner, evaluate, print one line
  python tools/c4_stress.py run [N]            N fresh child processes per mode (default 6), all modes, summary table

Modes (each a fresh process, i.e. cold instruction caches, cold allocator, first hipMalloc of the handle):
  plain        the test's own sequence (default stream)
  side         everything on a non-default NON-BLOCKING torch stream
  split        create on one non-blocking stream, evaluate on ANOTHER (ordered only by the create-ready event of the handle)
  poison       D3R_ALIGNER_POISON=1: every allocation of the handle pre-filled with 0xFF (NaN) before its initialisation
  serial       AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 (every launch / copy waits for the previous one)
  rebuild      in ONE process: 8 x (destroy, garbage into freed memory, create, evaluate): all evaluations must be bit-identical
A child prints `C4 <mode> loss=<float> rel=<loss / fp64 - 1> depthgrad_nan=<0|1> ok=<0|1>`; ok means |rel| < 1e-5."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.environ.get('D3R_C4_CACHE', '/tmp/d3r_c4_scene.pt')
GOLD = os.path.join(ROOT, 'tests', 'golden', 'aligner_c4.pt')


def prepare():
    import torch
    from dust3r_amd.synthetic import synthetic_scene
    g = torch.load(GOLD, weights_only=False)
    t = time.time()
    out, init, gt = synthetic_scene(g['n_views'], g['H'], g['W'], seed=g['seed'], symmetrize=g['symmetrize'])
    torch.save(dict(out=out, init=init, or64_loss0=g['or64_loss0']), CACHE)
    print(f'prepared {CACHE} in {time.time() - t:.1f}s')


def child(mode):
    import torch
    from dust3r_amd.cloud_opt import global_aligner
    d = torch.load(CACHE, weights_only=False)
    dev = torch.device('cuda:0')
    want = d['or64_loss0']

    def build_and_eval(stream_create=None, stream_eval=None):
        with torch.cuda.stream(stream_create) if stream_create is not None else _null():
            scene = global_aligner(d['out'], dev, verbose=False)
            scene.load_state_dict(d['init'])
            if stream_eval is not None and stream_eval is not stream_create:
                stream_eval.wait_stream(stream_create)       # the caller's own tensors; the handle's create work (next line) is NOT covered by this wait
            scene._ensure_engine()
        with torch.cuda.stream(stream_eval) if stream_eval is not None else _null():
            loss, grads = scene.loss_and_grads()
            val = float(loss)
            nan = int(bool(torch.isnan(grads['im_depthmaps']).any()) or bool(torch.isnan(grads['pw_poses']).any()))
        return scene, val, nan, grads

    if mode == 'rebuild':
        vals = []
        for k in range(8):
            scene, val, nan, grads = build_and_eval()
            vals.append((val, float(grads['im_depthmaps'].double().abs().sum()), nan))
            scene._destroy_engine()
            del scene, grads
            torch.cuda.empty_cache()
            junk = torch.full((int(2.5e9 // 4),), float('nan'), device=dev)      # garbage into the memory the handle just freed
            del junk
            torch.cuda.empty_cache()
        ok = int(all(v == vals[0] for v in vals) and abs(vals[0][0] / want - 1) < 1e-5 and vals[0][2] == 0)
        print(f'C4 rebuild loss={vals[0][0]:.7f} rel={vals[0][0] / want - 1:+.2e} distinct={len(set(vals))} depthgrad_nan={vals[0][2]} ok={ok}')
        return
    s1 = s2 = None
    if mode == 'side':
        s1 = s2 = torch.cuda.Stream(device=dev)
    elif mode == 'split':
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    scene, val, nan, _ = build_and_eval(s1, s2)
    ok = int(abs(val / want - 1) < 1e-5 and nan == 0)
    print(f'C4 {mode} loss={val:.7f} rel={val / want - 1:+.2e} depthgrad_nan={nan} ok={ok}')


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def run(n):
    if not os.path.exists(CACHE):
        prepare()
    envs = {'plain': {}, 'side': {}, 'split': {}, 'poison': {'D3R_ALIGNER_POISON': '1'},
            'serial': {'AMD_SERIALIZE_KERNEL': '3', 'AMD_SERIALIZE_COPY': '3'}, 'rebuild': {}}
    bad = 0
    for mode, extra in envs.items():
        reps = 1 if mode == 'rebuild' else n
        for k in range(reps):
            env = dict(os.environ, **extra)
            t = time.time()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', mode], env=env, capture_output=True, text=True, timeout=300)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith('C4 ')), f'C4 {mode} FAILED rc={r.returncode} {r.stderr[-300:]!r}')
            print(f'{line}   [{time.time() - t:.1f}s]', flush=True)
            bad += 0 if ' ok=1' in line else 1
    print(f'c4_stress: {bad} bad evaluation(s)')
    return bad


if __name__ == '__main__':
    cmd = sys.argv[1] if len(sys.argv) > 1 else 'run'
    if cmd == 'prepare':
        prepare()
    elif cmd == 'child':
        child(sys.argv[2])
    else:
        sys.exit(1 if run(int(sys.argv[2]) if len(sys.argv) > 2 else 6) else 0)
