"""GPU parity of the fused aligner (C ABI d3r_aligner_* through dust3r_amd.cloud_opt) against the fp64/fp32
oracle restatement and the golden trace produced by the unmodified reference optimizer."""
import os

import pytest
import torch

from dust3r_amd.synthetic import synthetic_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def make_scene(gpu, n, H, W, seed=0, symmetrize=True, **kw):
    from dust3r_amd.cloud_opt import global_aligner
    out, init, gt = synthetic_scene(n, H, W, seed=seed, symmetrize=symmetrize, **kw)
    scene = global_aligner(out, gpu, verbose=False)
    scene.load_state_dict(init)
    return scene, out, init, gt


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('n,H,W,sym', [(4, 16, 24, True), (5, 32, 48, False), (3, 64, 64, True)])
def test_loss_and_gradients_match_autograd(gpu, n, H, W, sym):
    from oracle.aligner_ref import AlignerRef
    scene, out, init, gt = make_scene(gpu, n, H, W, seed=n, symmetrize=sym)
    ref = AlignerRef(out, dtype=torch.float64).load_state(init)
    loss_ref, g_ref = ref.grads()
    loss, g = scene.loss_and_grads()
    assert abs(float(loss) / loss_ref - 1) < 1e-5
    for k in ('pw_poses', 'im_poses', 'im_depthmaps', 'im_focals'):
        err = rel(g[k].reshape(g_ref[k].shape), g_ref[k])
        print(k, err)
        assert err < 5e-5, (k, err)
    assert abs(float(scene()) / loss_ref - 1) < 1e-5           # forward() == the loss


def test_dpp_and_shuffle_reductions_agree(gpu):
    scene, *_ = make_scene(gpu, 4, 32, 48)
    scene.set_reduction(True)
    l1, g1 = scene.loss_and_grads()
    scene.set_reduction(False)
    l2, g2 = scene.loss_and_grads()
    assert abs(float(l1) / float(l2) - 1) < 1e-6
    for k in g1:
        assert rel(g1[k], g2[k]) < 1e-5, k


def test_short_trajectory_matches_oracle(gpu):
    """First 20 Adam iterations: per-iteration losses and parameters track the fp32 oracle (torch Adam) closely."""
    from oracle.aligner_ref import AlignerRef
    scene, out, init, gt = make_scene(gpu, 4, 32, 48, seed=7)
    ref = AlignerRef(out).load_state(init)
    ref_losses = ref.run(niter=20, lr=0.01, schedule='cosine', lr_min=1e-6)
    # run the same 20 iterations as the first 20 of a 20-iteration schedule
    from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
    last = global_alignment_loop(scene, lr=0.01, niter=20, schedule='cosine', lr_min=1e-6)
    assert abs(last / ref_losses[-1] - 1) < 1e-3
    st = ref.state()
    assert rel(scene.im_poses.data, st['im_poses']) < 2e-3
    assert rel(scene.pw_poses.data, st['pw_poses']) < 2e-3
    assert rel(scene.im_depthmaps.data, st['im_depthmaps']) < 2e-3
    assert rel(scene.im_focals.data, st['im_focals']) < 1e-4


def test_reference_golden_trace(gpu):
    """300 iterations against the trace recorded from the unmodified reference PointCloudOptimizer.
    The loop is chaotic at the 1e-3 level (two fp32 evaluations of the REFERENCE differ by that much),
    so: early losses tight, end state within the reference's own reproducibility floor."""
    g = torch.load(os.path.join(GOLD, 'aligner_4v.pt'), weights_only=False)
    scene, out, init, gt = make_scene(gpu, g['n_views'], g['H'], g['W'], seed=g['seed'])
    loss0, grads = scene.loss_and_grads()
    assert abs(float(loss0) / g['loss0'] - 1) < 1e-5
    for k, ref in g['grads'].items():
        assert rel(grads[k].reshape(ref.shape), ref) < 3e-4, k
    final = scene.compute_global_alignment(init=None, niter=g['niter'], schedule='cosine', lr=0.01)
    assert abs(final / g['final_loss'] - 1) < 5e-3
    assert float((scene.get_im_poses().cpu() - g['im_poses']).abs().max()) < 5e-3
    assert float((scene.get_focals().cpu().flatten() / g['focals'].flatten() - 1).abs().max()) < 5e-3


def test_noise_free_ground_truth_is_a_fixed_point(gpu):
    """With exact pairwise geometry and the ground-truth state, the loss is ~0 and stays there."""
    scene, out, init, gt = make_scene(gpu, 4, 32, 48, seed=2, noise=0.0, perturb=False)
    l0 = float(scene())
    assert l0 < 1e-5
    final = scene.compute_global_alignment(init=None, niter=30, schedule='cosine', lr=0.001)
    assert final < 5e-3


def test_converges_to_ground_truth_up_to_similarity(gpu):
    from dust3r_amd.utils.rigid import rigid_points_registration
    scene, out, init, gt = make_scene(gpu, 6, 48, 64, seed=4, noise=0.002)
    scene.compute_global_alignment(init=None, niter=300, schedule='cosine', lr=0.01)
    est = scene.get_im_poses().cpu()
    R, t, s = rigid_points_registration(est[:, :3, 3], gt['cam2world'][:, :3, 3], compute_scaling=True)
    aligned = s * est[:, :3, 3] @ R.T + t
    assert float((aligned - gt['cam2world'][:, :3, 3]).norm(dim=-1).max()) < 0.05          # cameras sit on a radius-2 circle
    # focals start 10 % off and im_focals = 20 log f moves at most lr per Adam step: 300 cosine iterations bring the
    # REFERENCE loop to ~7 % (oracle: 1.067..1.072 x gt on this scene); the engine must land where the oracle lands
    from oracle.aligner_ref import AlignerRef
    ref = AlignerRef(out).load_state(init)
    ref.run(niter=300, lr=0.01, schedule='cosine')
    f_ref = ref.focals().detach().flatten()
    f_eng = scene.get_focals().detach().cpu().flatten()
    assert float((f_eng / f_ref - 1).abs().max()) < 5e-3
    assert float((f_eng / gt['focal'] - 1).abs().max()) < 0.08


def test_mst_init_then_align_full_api(gpu):
    """demo.py's call sequence: global_aligner -> compute_global_alignment(init='mst', ...) -> getters."""
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    out, _, gt = synthetic_scene(5, 32, 48, seed=5, symmetrize=True, noise=0.002)
    scene = global_aligner(out, gpu, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
    loss = scene.compute_global_alignment(init='mst', niter=100, schedule='linear', lr=0.01)
    assert loss < 0.02
    assert scene.get_im_poses().shape == (5, 4, 4) and scene.get_focals().shape == (5, 1)
    pts = scene.get_pts3d()
    assert len(pts) == 5 and pts[0].shape == (32, 48, 3) and len(scene.get_masks()) == 5
    assert scene.get_depthmaps()[0].shape == (32, 48) and scene.get_intrinsics().shape == (5, 3, 3)
    scene.clean_pointcloud()
    assert float((scene.get_focals().cpu().flatten() / gt['focal'] - 1).abs().max()) < 0.05


def test_preset_pose_and_focal_freeze_parameters(gpu):
    scene, out, init, gt = make_scene(gpu, 4, 32, 48, seed=6)
    scene.preset_focal([gt['focal']] * 4)
    scene.preset_pose([gt['cam2world'][i] for i in range(4)])
    f0, p0 = scene.im_focals.data.clone(), scene.im_poses.data.clone()
    scene.compute_global_alignment(init=None, niter=10, schedule='cosine', lr=0.01)
    assert torch.equal(scene.im_focals.data, f0) and torch.equal(scene.im_poses.data, p0)


def test_clean_pointcloud_kernel_matches_host_loop(gpu):
    """d3r_clean_pointcloud vs the restated host double loop of base_opt.py:369-405 on the same scene (mixed image sizes
    exercise the padded layout). fp32 projections computed in a different association order can flip a rounded pixel index at
    an exact .5 boundary, so a vanishing fraction of differing pixels is tolerated; everything else is bit-equal."""
    from dust3r_amd.cloud_opt.base_opt import clean_pointcloud, clean_pointcloud_hip
    from dust3r_amd.utils.geometry import inv
    scene, out, init, gt = make_scene(gpu, 5, 48, 64, seed=9, noise=0.05)
    scene.compute_global_alignment(init=None, niter=20, schedule='cosine', lr=0.01)
    with torch.no_grad():
        scene.im_depthmaps.data[0] -= 0.25      # pull image 0's points 22 % closer: many now sit in front of the other views' depth
        scene.im_depthmaps.data[3] -= 0.10
        confs = [c.clone() for c in scene.im_conf]
        K, cams = scene.get_intrinsics(), inv(scene.get_im_poses())
        depth, pts = scene.get_depthmaps(), scene.get_pts3d()
        ref = clean_pointcloud([c.clone() for c in confs], K, cams, depth, pts, tol=0.001, bad_conf=0)
        got = clean_pointcloud_hip([c.clone() for c in confs], K, cams, depth, pts, tol=0.001, bad_conf=0)
    changed = sum(int((r != c).sum()) for r, c in zip(ref, confs))
    diff = sum(int((r != g).sum()) for r, g in zip(ref, got))
    total = sum(c.numel() for c in confs)
    print(f'clean_pointcloud: {changed} of {total} confidences clipped by the host loop, {diff} differ between kernel and host loop')
    assert changed > 0 and diff <= max(2, total // 5000)
    scene.clean_pointcloud()            # the method routes to the kernel on a CUDA scene
    assert all(torch.equal(a, b) for a, b in zip(scene.im_conf, got))
