"""Same-process A/B of aligner handle options read at creation (D3R_ALIGNER_NWV=4|8 ...): BASELINE configs[3]'s scene (20 views, 190 edges, or
380 with --sym), one scene per arm created under its environment value, 300 cosine iterations timed with HIP events, alternating repetitions.
Usage: python tools/aligner_probe.py VAR=a,b [--reps=4] [--sym]"""
import os
import sys

sys.path.insert(0, '.')
import torch  # noqa: E402


def main():
    spec = next(a for a in sys.argv[1:] if '=' in a and not a.startswith('--'))
    var, vals = spec.split('=')
    vals = vals.split(',')
    reps = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--reps=')), 4))
    sym = '--sym' in sys.argv
    dev = torch.device('cuda:0')
    from dust3r_amd.cloud_opt import global_aligner
    from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
    from dust3r_amd.synthetic import synthetic_scene
    out, init, gt = synthetic_scene(20, 384, 512, seed=0, symmetrize=sym, device=dev, device_rng=True)
    scenes = {}
    for v in vals:
        os.environ[var] = v
        sc = global_aligner(out, dev, verbose=False)
        sc.load_state_dict(init)
        global_alignment_loop(sc, niter=5)
        scenes[v] = sc
    E, n, A = scenes[vals[0]].n_edges, scenes[vals[0]].n_imgs, 384 * 512
    gb = (E * A * 32 + n * A * 24) / 1e9
    acc = {v: [] for v in vals}
    loss = {}
    for r in range(reps):
        for v in vals:
            sc = scenes[v]
            sc.load_state_dict(init)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            loss[v] = global_alignment_loop(sc, niter=300, schedule='cosine', lr=0.01)
            e1.record()
            torch.cuda.synchronize()
            acc[v].append(e0.elapsed_time(e1))
    for v in vals:
        ms = sorted(acc[v])[len(acc[v]) // 2]
        print(f'{var}={v}: {E} edges, 300 iterations median {ms:.2f} ms ({300 / ms * 1e3:.1f} it/s = {gb * 300 / ms:.2f} TB/s = {gb * 300 / ms / 8:.3f} of 8 TB/s), runs {[round(x, 1) for x in acc[v]]}, final loss {loss[v]:.7f}')


if __name__ == '__main__':
    main()
