"""CPU tests (-m "not gpu") of the host logic and of the C-ABI library surface (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dust3r_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'dust3r_hip.h')).read()
    declared = set(re.findall(r'\b(d3r_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(_lib.lib, name), f'{name} declared in include/dust3r_hip.h but not exported'
    assert set(_lib.EXPORTED) <= declared
    assert b'gfx950' in _lib.lib.d3r_version()


def test_product_fails_loudly_without_gpu():
    from dust3r_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, synthetic_views
    m = AsymmetricCroCo3DStereo(**MODEL_CONFIGS['tiny_linear'])
    with pytest.raises(_lib.D3RError):
        m(*synthetic_views(1, 32, 32))
    with pytest.raises(_lib.D3RError):
        m.to('cuda')


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'dust3r_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or '/root/reference' in src and f.endswith('.py'):
                    bad.append(f)
    assert not bad, bad


def test_oracle_never_imports_the_product_algorithms():
    """The inverse direction (round 6): the only product module anything under oracle/ may import is dust3r_amd.synthetic -- the seeded INPUT generators the
    goldens are rebuilt from. Until round 5 oracle/shims/cv2.py forwarded cv2.solvePnPRansac to dust3r_amd.cloud_opt.pnp, so the PnP goldens held the
    product's own answers."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'oracle')):
        for f in files:
            if f.endswith('.py'):
                for m in re.finditer(r'^\s*(?:from|import)\s+(dust3r_amd[\w.]*)', open(os.path.join(dirpath, f)).read(), re.M):
                    if m.group(1) != 'dust3r_amd.synthetic':
                        bad.append((f, m.group(1)))
    assert not bad, bad


@pytest.mark.parametrize('graph,sym', [('complete', True), ('complete', False), ('swin-3', True), ('swin-2-noncyclic', False),
                                       ('logwin-3', True), ('oneref-2', True)])
def test_make_pairs_matches_reference_semantics(graph, sym):
    from dust3r_amd.image_pairs import make_pairs
    n = 9
    imgs = [dict(idx=i) for i in range(n)]
    pairs = [(a['idx'], b['idx']) for a, b in make_pairs(imgs, graph, None, sym)]
    base = pairs[:len(pairs) // 2] if sym else pairs
    if sym:
        assert pairs[len(pairs) // 2:] == [(j, i) for i, j in base]        # reversed pairs appended at the end (image_pairs.py:58-59)
    if graph == 'complete':
        assert base == [(i, j) for i in range(n) for j in range(i)]
    if graph == 'swin-3':
        assert len(base) == 3 * n and all((j - i) % n in (1, 2, 3) or (i - j) % n in (1, 2, 3) for i, j in base)
    if graph == 'oneref-2':
        assert base == [(2, j) for j in range(n) if j != 2]
    from oracle.ref_import import reference_available
    if reference_available():
        from oracle.ref_import import import_reference
        import_reference()
        from dust3r.image_pairs import make_pairs as ref_make_pairs
        assert pairs == [(a['idx'], b['idx']) for a, b in ref_make_pairs(imgs, graph, None, sym)]


def test_make_pairs_prefilter():
    from dust3r_amd.image_pairs import make_pairs
    imgs = [dict(idx=i) for i in range(8)]
    seq = make_pairs(imgs, 'complete', 'seq2', True)
    assert all(abs(a['idx'] - b['idx']) <= 2 for a, b in seq)
    cyc = make_pairs(imgs, 'complete', 'cyc1', False)
    assert {(a['idx'], b['idx']) for a, b in cyc} >= {(7, 0)}


def test_collate_and_symmetry_helpers():
    from dust3r_amd.inference import make_batch_symmetric
    from dust3r_amd.utils.device import collate_with_cat
    from dust3r_amd.utils.misc import interleave, is_symmetrized
    a = dict(img=torch.zeros(1, 3, 4, 4), idx=0, instance='0', true_shape=np.int32([[4, 4]]))
    b = dict(img=torch.ones(1, 3, 4, 4), idx=1, instance='1', true_shape=np.int32([[4, 4]]))
    v1, v2 = collate_with_cat([(a, b), (b, a)])
    assert v1['img'].shape == (2, 3, 4, 4) and v1['idx'] == [0, 1] and v2['instance'] == ['1', '0']
    assert is_symmetrized(v1, v2)
    s1, s2 = make_batch_symmetric(collate_with_cat([(a, b)]))
    assert s1['instance'] == ['0', '1'] and s2['instance'] == ['1', '0'] and is_symmetrized(s1, s2)
    x, y = interleave(torch.tensor([1, 2]), torch.tensor([3, 4]))
    assert x.tolist() == [1, 3, 2, 4] and y.tolist() == [3, 1, 4, 2]


def test_geotrf_inv_roundtrip_and_rigid_helpers():
    from dust3r_amd.utils.geometry import geotrf, inv, xy_grid
    from dust3r_amd.utils.rigid import quat_translation_to_homogeneous, rigid_points_registration, rotmat_to_unitquat, unitquat_to_rotmat
    g = torch.Generator().manual_seed(0)
    q = torch.randn((5, 4), generator=g)
    T = quat_translation_to_homogeneous(q, torch.randn((5, 3), generator=g))
    x = torch.randn((5, 6, 7, 3), generator=g)
    assert torch.allclose(geotrf(inv(T), geotrf(T, x)), x, atol=1e-5)
    R = unitquat_to_rotmat(q / q.norm(dim=-1, keepdim=True))
    q2 = rotmat_to_unitquat(R)
    assert torch.allclose(unitquat_to_rotmat(q2), R, atol=1e-5)
    grid = xy_grid(4, 3, device='cpu')
    assert grid.shape == (3, 4, 2) and grid[2, 1].tolist() == [1, 2]
    # similarity Procrustes recovers a known (s, R, t)
    pts = torch.randn((200, 3), generator=g)
    s, t = 1.7, torch.tensor([0.3, -0.2, 0.9])
    Rr, tt, ss = rigid_points_registration(pts, s * pts @ R[0].T + t, weights=torch.rand(200, generator=g) + 0.1, compute_scaling=True)
    assert torch.allclose(Rr, R[0], atol=1e-4) and abs(float(ss) - s) < 1e-4 and torch.allclose(tt, t, atol=1e-4)


def test_pnp_recovers_pose():
    from dust3r_amd.cloud_opt.pnp import rodrigues_to_rotmat, rotmat_to_rodrigues, solve_pnp_ransac
    rng = np.random.RandomState(0)
    R = rodrigues_to_rotmat(np.array([0.2, -0.4, 0.1]))
    T = np.array([0.1, -0.2, 3.0])
    X = rng.uniform(-1, 1, size=(500, 3))
    K = np.array([[300., 0, 160], [0, 300., 120], [0, 0, 1]])
    Xc = X @ R.T + T
    pix = np.stack((K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]), 1)
    pix[:25] += rng.uniform(-40, 40, size=(25, 2))          # 5 % outliers
    ok, R2, T2, inl = solve_pnp_ransac(X, pix, K, iterations=30, reproj_err=3)
    assert ok and len(inl) >= 470 and np.abs(R2 - R).max() < 1e-6 and np.abs(T2 - T).max() < 1e-6
    assert np.allclose(rodrigues_to_rotmat(rotmat_to_rodrigues(R)), R, atol=1e-10)


def test_scene_bootstrap_needs_a_gpu():
    """init='mst' and PairViewer are evaluated by HIP kernels: on a CPU scene they raise instead of falling back."""
    from dust3r_amd import _lib
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.synthetic import synthetic_scene
    out, init, gt = synthetic_scene(3, 16, 24, seed=1, symmetrize=True)
    scene = global_aligner(out, 'cpu', verbose=False)
    with pytest.raises(_lib.D3RError):
        scene.compute_global_alignment(init='mst', niter=1)
    out2, _, _ = synthetic_scene(2, 16, 24, seed=2, symmetrize=True)
    pv = global_aligner(out2, 'cpu', mode=GlobalAlignerMode.PairViewer, verbose=False)
    with pytest.raises(_lib.D3RError):
        pv.get_focals()


def test_analytic_aligner_gradients_on_host():
    """aligner_math.hpp (shared with the kernels) vs autograd of the fp64 oracle, through the host self-test hook."""
    from dust3r_amd import _lib
    from dust3r_amd.synthetic import synthetic_scene
    from oracle.aligner_ref import AlignerRef
    out, init, gt = synthetic_scene(4, 16, 24, seed=3, symmetrize=True)
    al = AlignerRef(out, dtype=torch.float64).load_state(init)
    loss, grads = al.grads()
    E, n, H, W = len(al.edges), al.n_imgs, al.H, al.W
    ei = np.array([e[0] for e in al.edges], np.int32)
    ej = np.array([e[1] for e in al.edges], np.int32)
    f32 = lambda t: np.ascontiguousarray(t.detach().float().numpy())  # noqa: E731
    arrs = [f32(al.pred_i), f32(al.pred_j), f32(al.weight_i), f32(al.weight_j), f32(init['pw_poses']), f32(init['im_poses']),
            f32(init['im_depthmaps']), f32(init['im_focals'])]
    lo, gpw, gimp, gdep, gfoc = np.zeros(1), np.zeros((E, 8)), np.zeros((n, 7)), np.zeros((n, H * W)), np.zeros(n)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = _lib.lib.d3r_selftest_aligner_math_host(n, E, p(ei), p(ej), H, W, *[p(a) for a in arrs], 0.5, 20.0, p(lo), p(gpw), p(gimp), p(gdep), p(gfoc))
    assert rc == 0 and abs(lo[0] / loss - 1) < 1e-5
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    assert rel(gpw, grads['pw_poses'].numpy()) < 2e-5
    assert rel(gimp, grads['im_poses'].numpy()) < 2e-5
    assert rel(gdep, grads['im_depthmaps'].numpy()) < 2e-5
    assert rel(gfoc, grads['im_focals'].numpy().ravel()) < 2e-5


def test_model_state_dict_duplication_and_parsing():
    from dust3r_amd.model import AsymmetricCroCo3DStereo, expected_state, parse_model_string
    from dust3r_amd.synthetic import MODEL_CONFIGS, synthetic_state_dict
    kw = parse_model_string("AsymmetricCroCo3DStereo(pos_embed='RoPE100', patch_embed_cls='ManyAR_PatchEmbed', img_size=(512, 512), "
                            "head_type='dpt', output_mode='pts3d', depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), "
                            "enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768, dec_depth=12, dec_num_heads=12)")
    assert kw['img_size'] == (512, 512) and kw['depth_mode'] == ('exp', -float('inf'), float('inf')) and kw['enc_depth'] == 24
    m = AsymmetricCroCo3DStereo(**MODEL_CONFIGS['tiny_linear'])
    spec = {k: torch.zeros(v) for k, v in m._spec.items()}
    sd = synthetic_state_dict(spec, 0)
    no_dec2 = {k: v for k, v in sd.items() if not k.startswith('dec_blocks2')}
    r = m.load_state_dict(no_dec2, strict=True)                      # model.py:91-98 duplication
    assert not r.missing_keys
    assert torch.equal(m.state_dict()['dec_blocks2.0.attn.qkv.weight'], sd['dec_blocks.0.attn.qkv.weight'])


def test_c_abi_is_usable_from_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/dust3r_hip.h must compile as C99 (-pedantic), every entry point a C caller uses
    must link against libdust3r_hip.so, and argument errors must come back as D3R_ERR_* codes without a device (tests/c_abi/probe.c)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    from dust3r_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / 'c_abi_probe')
    cmd = [gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'c_abi', 'probe.c'),
           '-o', exe, '-L', libdir, '-ldust3r_hip', f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'c_abi probe ok' in r.stdout, r.stdout + r.stderr


def _write_case_png(tmp_path, k, W, H):
    import PIL.Image
    from dust3r_amd.synthetic import synthetic_photo
    path = os.path.join(str(tmp_path), f'img{k}.png')
    PIL.Image.fromarray(synthetic_photo(W, H, seed=k)).save(path)
    return path


def test_load_images_matches_reference_golden(tmp_path):
    """utils/image.py::load_images against tensors produced by the unmodified reference's load_images on the same synthetic
    pictures (tests/golden/load_images.pt): resize rule, LANCZOS / BICUBIC choice, crop rule (4:3 for squares, multiples of the patch
    size, the 224 rule), normalisation -- bit for bit."""
    from dust3r_amd.synthetic import LOAD_IMAGES_CASES
    from dust3r_amd.utils.image import load_images
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'load_images.pt'), weights_only=False)
    assert len(g['cases']) == len(LOAD_IMAGES_CASES)
    for k, c in enumerate(g['cases']):
        W, H = c['src']
        v = load_images([_write_case_png(tmp_path, k, W, H)], size=c['size'], square_ok=c['square_ok'], verbose=False)[0]
        ref = (c['u8'].float().div(255) - 0.5) / 0.5
        assert v['img'].shape == (1,) + tuple(ref.shape) and v['img'].dtype == torch.float32, (k, v['img'].shape, ref.shape)
        assert torch.equal(v['img'][0], ref), k
        assert (v['true_shape'] == c['true_shape']).all() and v['true_shape'].dtype == np.int32 and v['idx'] == 0 and v['instance'] == '0'


def test_load_images_equals_live_reference(tmp_path):
    """Same comparison against the LIVE reference function over a sweep of source sizes at the release resolutions (512, 224); needs
    /root/reference (build container only)."""
    from oracle.ref_import import import_reference, reference_available
    if not reference_available():
        pytest.skip('/root/reference only exists in the build container')
    import_reference()
    from dust3r.utils.image import load_images as ref_load
    from dust3r_amd.utils.image import fit_geometry, load_images
    paths, k = [], 0
    for (W, H) in [(1280, 960), (960, 1280), (800, 800), (1023, 577), (400, 300), (513, 512), (511, 512), (2000, 500), (64, 48)]:
        paths.append(_write_case_png(tmp_path, k, W, H))
        k += 1
    for size, sq in ((512, False), (512, True), (224, False)):
        ours, refs = load_images(paths, size=size, square_ok=sq, verbose=False), ref_load(paths, size=size, square_ok=sq, verbose=False)
        assert len(ours) == len(refs)
        for a, b in zip(ours, refs):
            assert torch.equal(a['img'], b['img']) and (a['true_shape'] == b['true_shape']).all() and a['idx'] == b['idx'] and a['instance'] == b['instance']
    # the geometry alone, exhaustively over small sizes, against what the reference's arithmetic gives (image.py:98-116)
    for W1 in range(30, 140, 7):
        for H1 in range(30, 140, 11):
            (W, H), box = fit_geometry(W1, H1, 96)
            assert max(W, H) == 96 and box[2] - box[0] <= W and (box[2] - box[0]) % 16 == 0


def test_collated_views_from_shared_images_equal_plain_collation():
    """inference()'s view dicts: the gather from the stack of distinct images (done on the GPU in production) gives exactly what the
    reference's concatenation of every pair's views gives."""
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import _collate_views, _shared_images
    from dust3r_amd.synthetic import synthetic_image_list
    from dust3r_amd.utils.device import collate_with_cat
    pairs = make_pairs(synthetic_image_list(5, 16, 32, seed=2), 'swin-2', None, symmetrize=True)
    shared = _shared_images(pairs)
    assert shared is not None and len(shared[0]) == 5
    v1, v2 = _collate_views(pairs, shared, torch.cat(shared[0], dim=0))
    r1, r2 = collate_with_cat(list(pairs))
    for a, b in ((v1, r1), (v2, r2)):
        assert set(a) == set(b) and torch.equal(a['img'], b['img']) and a['idx'] == b['idx'] and a['instance'] == b['instance']
        assert torch.equal(torch.as_tensor(a['true_shape']), torch.as_tensor(b['true_shape']))


def test_inference_groups_mixed_shapes_and_coalesces_batches():
    """`inference()` on a model that declares `engine_batch`: pairs of mixed image sizes are grouped by shape and batched, same-size
    lists run `engine_batch` pairs per call whatever batch_size says -- and the returned structure and values equal the reference's
    schedule (one pair per call for mixed sizes, batch_size per call otherwise), here on a deterministic stand-in network."""
    import torch
    from dust3r_amd.inference import inference

    class StandIn:
        calls = None

        def __init__(self, engine_batch=None):
            self.calls = []
            if engine_batch:
                self.engine_batch = engine_batch

        def __call__(self, view1, view2):
            a, b = view1['img'], view2['img']
            self.calls.append(a.shape[0])
            pts1 = torch.stack((a.mean((1, 2, 3)), b.mean((1, 2, 3)), (a * 2).amax((1, 2, 3))), -1)[:, None, None, :].expand(-1, a.shape[2], a.shape[3], -1) + a.permute(0, 2, 3, 1)
            pts2 = b.permute(0, 2, 3, 1) * 3 - b.mean((1, 2, 3))[:, None, None, None]
            return dict(pts3d=pts1.contiguous(), conf=1 + a.sum(1).abs()), dict(pts3d_in_other_view=pts2.contiguous(), conf=1 + b.sum(1).abs())

    g = torch.Generator().manual_seed(0)
    shapes = [(16, 32), (32, 16), (16, 32), (16, 16), (32, 16), (16, 32)]
    imgs = [dict(img=torch.rand((1, 3, h, w), generator=g), true_shape=torch.tensor([[h, w]], dtype=torch.int32), idx=k, instance=str(k))
            for k, (h, w) in enumerate(shapes)]
    pairs = [(imgs[i], imgs[j]) for i in range(len(imgs)) for j in range(len(imgs)) if i != j]
    plain, eng = StandIn(), StandIn(engine_batch=4)
    ref = inference(pairs, plain, 'cpu', batch_size=1, verbose=False)
    out = inference(pairs, eng, 'cpu', batch_size=1, verbose=False)
    assert set(plain.calls) == {1} and max(eng.calls) == 4 and len(eng.calls) < len(plain.calls) // 2
    for view in ('view1', 'view2', 'pred1', 'pred2'):
        assert ref[view].keys() == out[view].keys()
        for k in ref[view]:
            a, b = ref[view][k], out[view][k]
            assert type(a) is type(b) and len(a) == len(b) == len(pairs), (view, k)
            for x, y in zip(a, b):
                assert (torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y), (view, k)
    # same-size list: engine_batch pairs per call instead of batch_size, identical collated tensors
    same = [(imgs[i], imgs[j]) for i in (0, 2, 5) for j in (0, 2, 5) if i != j]
    plain, eng = StandIn(), StandIn(engine_batch=4)
    ref = inference(same, plain, 'cpu', batch_size=1, verbose=False)
    out = inference(same, eng, 'cpu', batch_size=1, verbose=False)
    assert plain.calls == [1] * 6 and eng.calls == [4, 2]
    for view in ('pred1', 'pred2'):
        for k in ref[view]:
            assert torch.equal(ref[view][k], out[view][k])


def test_fp16_fp8_row_layout_on_the_host():
    """The host-side packers of the fp16 + fp8 operand rows (dust3r_amd/ops.py: the layout csrc/common.hpp documents for
    Traits<D3R_F16F8>) and the fp64 emulation of the contraction the GPU tests compare the kernel with: byte layout of a 256-byte
    super-group, the activation / weight encodings, the decoded value (15-16 bits), saturation of the e4m3 copies at 448, and the
    accuracy class of the scheme (an fp16-only product is ~30x worse)."""
    from dust3r_amd import ops
    from oracle.f8_ref import e4m3, f16f8_matmul
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 128, generator=g) * 3
    w = torch.randn(7, 128, generator=g) * 0.03
    px, pw = ops.pack_f8(x), ops.pack_f8(w, weight=True)
    assert px.dtype == torch.uint8 and px.shape == (5, 512) and pw.shape == (7, 512)
    hi, a8, b8 = ops.unpack_f8(px, parts=True)
    assert torch.equal(hi, x.half().float())                                       # bytes 0..127 of a super-group: 64 fp16 values
    assert torch.equal(a8, ops._e4m3(hi)) and torch.equal(b8, ops._e4m3((x - hi) * 2048))   # then a8 x64, then b8 x64
    assert torch.equal(px[0, :4], x[0, :2].half().view(torch.uint8)) and px[0, 128] == a8[0, 0] and px[0, 192] == b8[0, 0]
    whi, wa8, wb8 = ops.unpack_f8(pw, weight=True, parts=True)
    assert torch.equal(wa8, ops._e4m3((w - whi) * 131072)) and torch.equal(wb8, ops._e4m3(whi * 64))    # weights: [lo | hi] with the 2^6 shift
    assert float((ops.unpack_f8(px) - x).abs().max() / x.abs().max()) < 2e-5
    assert float((ops.unpack_f8(pw, weight=True) - w).abs().max() / w.abs().max()) < 2e-5
    big = torch.full((1, 64), 1000.0)
    assert int(ops.unpack_f8(ops.pack_f8(big), parts=True)[1][0, 0]) == 0x7E        # e4m3(1000) saturates at 448, never NaN
    # the product's packers against the oracle's independent restatement of the encodings: decode the packed bytes and contract them
    f8v = lambda t: t.contiguous().view(torch.float8_e4m3fn).double()      # noqa: E731
    from_bytes = hi.double() @ whi.double().T + (f8v(a8) @ f8v(wa8).T + f8v(b8) @ f8v(wb8).T) / 131072.0
    assert torch.equal(from_bytes, f16f8_matmul(x, w))
    assert torch.equal(f8v(a8), e4m3(hi))
    exact = x.double() @ w.double().T
    err = float((f16f8_matmul(x, w) - exact).abs().max() / exact.abs().max())
    err16 = float((x.half().double() @ w.half().double().T - exact).abs().max() / exact.abs().max())
    assert err < 3e-5 and err16 > 10 * err


def test_gemm_tile_dispatch_table(monkeypatch):
    """The GEMM tile heuristic as a host function (d3r_gemm_tile_config, no device): the shapes of the BASELINE forward at 32 pairs per
    step and at one pair per call land on the tile configurations DESIGN.md section 4.1 / 6 report, and the probe variables move them."""
    from dust3r_amd._lib import DTYPE_BF16, DTYPE_F16F8, DTYPE_F16X3, lib
    for v in ('D3R_GEMM_CFG', 'D3R_GEMM_T64', 'D3R_GEMM_T256', 'D3R_GEMM_R', 'D3R_GEMM_MID', 'D3R_GEMM_F32CFG', 'D3R_GEMM_PP', 'D3R_GEMM_T384'):
        monkeypatch.delenv(v, raising=False)
    PLAIN, F32, GELU = 0, 1, 2
    cfg = lambda dt, M, N, K, epi=PLAIN, res=0: lib.d3r_gemm_tile_config(dt, M, N, K, epi, res)   # noqa: E731
    x3 = DTYPE_F16X3
    # round 6: the persistent kernel (configuration 10) takes the launches whose 256 x 128 tiles fill whole rounds of the CUs and whose epilogue is a large share of
    # a tile's life -- fc1 + GELU and the plain typed stores of the 32-pair step; never the fp32-residual epilogue, never small batches (fewer tiles than CUs)
    assert cfg(x3, 49152, 4096, 1024, GELU) == 10 and cfg(x3, 49152, 1024, 1024) == 10 and cfg(x3, 24576, 3072, 768, GELU) == 10
    assert cfg(x3, 49152, 1024, 4096, F32, 1) == 1 and cfg(x3, 1536, 4096, 1024, GELU) == 0 and cfg(DTYPE_BF16, 49152, 4096, 1024, GELU) == 1
    monkeypatch.setenv('D3R_GEMM_PERSIST', '0')        # the one-tile-per-block table below it
    # 32 pairs = 64 images x 768 tokens: the encoder's four linears, the decoder's 768-wide ones (one side = 24576 rows)
    assert cfg(x3, 49152, 4096, 1024, GELU) == 1 and cfg(x3, 49152, 1024, 4096, F32, 1) == 1 and cfg(x3, 49152, 1024, 1024) == 1
    assert cfg(x3, 49152, 1024, 1024, F32, 1) == 7           # fp32-residual projection at K <= 1024: two blocks per CU, weights in registers
    assert cfg(x3, 49152, 1024, 1024, F32, 0) == 1           # ... only with a residual
    # round 4: the decoder's 24576-row GEMMs whose (M / 384) x (N / 192) tiles fill whole rounds of 256 CUs take the 384 x 192 tile (configuration 9)
    assert cfg(x3, 24576, 768, 3072, F32, 1) == 9 and cfg(x3, 24576, 3072, 768, GELU) == 9 and cfg(x3, 24576, 2304, 768) == 9 and cfg(x3, 24576, 768, 768, F32, 1) == 9
    assert cfg(DTYPE_BF16, 24576, 768, 3072, F32, 1) != 9 and cfg(x3, 24576 - 384 * 20, 768, 3072, F32, 1) != 9      # split-fp16 only; 176 tiles do not fill the chip
    probes = bool(lib.d3r_build_has_probes())      # the probe-only switches move the table in probe builds only (D3R_PROBES=1 python -m dust3r_amd.build)
    if probes:
        monkeypatch.setenv('D3R_GEMM_T384', '0')
        assert cfg(x3, 24576, 768, 3072, F32, 1) == 0 and cfg(x3, 24576, 3072, 768, GELU) == 1 and cfg(x3, 24576, 768, 768, F32, 1) == 0
        monkeypatch.delenv('D3R_GEMM_T384')
    assert cfg(x3, 6291456, 128, 1152) == 3 and cfg(x3, 196608, 128, 1152) == 2 and cfg(x3, 1000, 96, 768) == 0   # N <= 128: the head's shapes
    # one pair per call = 1536 encoder rows / 768 decoder rows per side: small problems on the 64 x 64 tile, mid-size ones stay on 128 x 128
    assert cfg(x3, 1536, 1024, 4096, F32, 1) == 8 and cfg(x3, 1536, 1024, 1024, F32, 1) == 8 and cfg(x3, 768, 768, 768, F32, 1) == 8
    assert cfg(x3, 1536, 3072, 1024) == 0 and cfg(x3, 1536, 4096, 1024, GELU) == 0
    # round 6: M 96 x N 64 (configuration 11) for the long K loops (K >= 2048) whose 96 x 64 tiles come to 1.5 ... 2 per CU -- the encoder's fc2 at two pairs (512 tiles)
    # and three (384), the decoder's fc2 at four (384); not at one tile per CU (1536 x 1024: 256 tiles stay on 64 x 64), not at K = 768 / 1024 (no gain inside the forward)
    assert cfg(x3, 3072, 1024, 4096, F32, 1) == 11 and cfg(x3, 2304, 1024, 4096, F32, 1) == 11 and cfg(x3, 3072, 768, 3072, F32, 1) == 11
    assert cfg(x3, 768, 3072, 768, GELU) == 8 and cfg(x3, 1536, 1536, 768) == 8 and cfg(x3, 1152, 768, 3072, F32, 1) == 8 and cfg(DTYPE_BF16, 3072, 1024, 4096, F32, 1) == 0
    # the small tile exists for split-fp16 only
    assert cfg(DTYPE_BF16, 1536, 1024, 4096, F32, 1) == 0 and cfg(DTYPE_F16F8, 1536, 1024, 4096, F32, 1) == 0
    # fp16 + fp8 rows go to the 256-wide tile from one round of resident blocks on
    assert cfg(DTYPE_F16F8, 24576, 768, 768, F32, 1) == 1
    monkeypatch.delenv('D3R_GEMM_PERSIST')
    # probes
    monkeypatch.setenv('D3R_GEMM_T64', '0')
    assert cfg(x3, 768, 768, 768, F32, 1) == (0 if probes else 8)
    monkeypatch.delenv('D3R_GEMM_T64')
    monkeypatch.setenv('D3R_GEMM_CFG', '2')
    assert cfg(x3, 49152, 4096, 1024, GELU) == 2
    assert lib.d3r_gemm_tile_config(x3, 0, 8, 8, 0, 0) < 0 or lib.d3r_gemm_tile_config(x3, 0, 8, 8, 0, 0) > 8     # invalid arguments: an error code, not a configuration


def test_postprocess_mode_keywords_follow_the_reference():
    """model.py:58-62 / heads/postprocess.py:23-58: every depth / conf mode of the reference is accepted (the engine implements them,
    include/dust3r_hip.h d3r_model_set_postprocess), an unknown mode raises the reference's ValueError, depth bounds are asserted away."""
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS
    inf = float('inf')
    for dm in ('exp', 'linear', 'square'):
        for cm in (('exp', 1, inf), ('exp', 0, 5), ('sigmoid', 0, 1)):
            m = AsymmetricCroCo3DStereo(depth_mode=(dm, -inf, inf), conf_mode=cm, landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])
            assert m.depth_mode == (dm, -inf, inf) and m.conf_mode == cm
    with pytest.raises(ValueError):
        AsymmetricCroCo3DStereo(depth_mode=('cube', -inf, inf), landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])
    with pytest.raises(ValueError):
        AsymmetricCroCo3DStereo(conf_mode=('tanh', 0, 1), landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])
    with pytest.raises(AssertionError):
        AsymmetricCroCo3DStereo(depth_mode=('exp', 0, 10), landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])
    with pytest.raises(AssertionError):
        AsymmetricCroCo3DStereo(conf_mode=('sigmoid', 0, inf), landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])


def test_2p5_unit_weight_rows_host_packer_against_the_emulation():
    """dust3r_amd.ops.pack_w5 (the five-chunk weight rows of D3R_DTYPE_F16X2F8) decoded chunk by chunk reproduces the factors oracle/f8_ref.py's
    emulation uses: w_hi, fp16(w_lo) and e4m3(w_hi 2^6), in the documented positions (include/dust3r_hip.h)."""
    import torch
    from dust3r_amd.ops import pack_w5
    from oracle.f8_ref import W_SHIFT, e4m3, split
    g = torch.Generator().manual_seed(3)
    w = torch.randn((5, 256), generator=g) * 0.05
    rows = pack_w5(w)
    assert rows.shape == (5, 5 * 256) and rows.dtype == torch.uint8
    wh, wl = split(w)
    blk = rows.reshape(5, 2, 640)
    for b in range(2):
        for half in range(2):
            k0 = b * 128 + half * 64
            hi = blk[:, b, (2 * half) * 128:(2 * half + 1) * 128].contiguous().view(torch.float16).double()
            lo = blk[:, b, (2 * half + 1) * 128:(2 * half + 2) * 128].contiguous().view(torch.float16).double()
            assert torch.equal(hi, wh[:, k0:k0 + 64]) and torch.equal(lo, wl[:, k0:k0 + 64].float().half().double())
        h8 = blk[:, b, 512:640].contiguous().view(torch.float8_e4m3fn).double()
        assert torch.equal(h8, e4m3(wh[:, b * 128:(b + 1) * 128] * W_SHIFT))


def test_bench_emits_strictly_valid_json(capsys):
    """bench.py prints ONE line that every JSON parser accepts: non-finite floats (a NaN loss of a degenerate alignment, an infinite bound) become null."""
    import json
    import bench
    bench.emit({'value': 1.5, 'stages': {'final_loss': float('nan'), 'bound': float('inf'), 'list': [1, float('-inf'), 'x']}, 'ok': True, 'none': None})
    line = capsys.readouterr().out.strip()
    assert '\n' not in line and 'NaN' not in line and 'Infinity' not in line
    d = json.loads(line)
    assert d == {'value': 1.5, 'stages': {'final_loss': None, 'bound': None, 'list': [1, None, 'x']}, 'ok': True, 'none': None}


def test_bench_workload_flop_accounting_matches_the_survey():
    """1856.8 GFLOP per 512x384 DPT pair = 2 x 523.0 (encoder) + 2 x 218.6 (decoder) + 2 x 186.7 (DPT head); the encode-once workloads of
    bench.py count executed flops with the same constants."""
    import bench
    assert abs(2 * bench.ENC_GFLOP_PER_IMAGE + bench.DEC_HEAD_GFLOP_PER_PAIR - bench.GFLOP_PER_PAIR) < 0.5


def test_host_result_tensors_and_image_upload_helpers():
    """utils/device.py: host_tensor (the result memory of inference(): zero-filled, on huge pages when large, an ordinary CPU tensor for every
    consumer) and upload_stack (falls back to cat + to for anything that is not a list of equally shaped one-image CPU tensors going to a GPU)."""
    import gc

    import numpy as np
    from dust3r_amd.utils.device import host_tensor, upload_stack
    big = host_tensor((12, 96, 128, 3))                      # 1.7 MB: plain route
    huge = host_tensor((24, 384, 512, 3))                    # 56 MB: the mapped route where the platform has it
    for t in (big, huge):
        assert t.dtype == torch.float32 and t.device.type == 'cpu' and t.is_contiguous() and float(t.abs().max()) == 0.0
        t[1, 2, 3] = torch.tensor([1.0, 2.0, 3.0])
        assert np.asarray(t.numpy()[1, 2, 3]).tolist() == [1.0, 2.0, 3.0]
    keep = huge[5:7]                                         # a view must keep the mapping alive after the parent name is gone
    keep.fill_(4.0)
    del huge
    gc.collect()
    assert float(keep.sum()) == 4.0 * keep.numel()
    assert torch.equal(torch.cat((keep, keep)), keep.repeat(2, 1, 1, 1)) and host_tensor((3, 5), dtype=torch.int64).dtype == torch.int64
    imgs = [torch.full((1, 3, 4, 4), float(k)) for k in range(5)]
    assert torch.equal(upload_stack(imgs, 'cpu'), torch.cat(imgs))
    assert torch.equal(upload_stack([torch.ones(2, 3), torch.zeros(1, 3)], 'cpu'), torch.tensor([[1.0] * 3, [1.0] * 3, [0.0] * 3]))


def test_load_images_thread_pool_keeps_pixels_and_order(tmp_path, monkeypatch):
    """load_images decodes / resamples the files of a folder on a thread pool (PIL releases the GIL): same tensors, same idx / instance order as the
    serial loop (DUST3R_AMD_LOAD_THREADS=1), files that are not images skipped as in the reference (image.py:84-86)."""
    from dust3r_amd.utils.image import load_images
    sizes = [(640, 480), (480, 640), (500, 500), (1280, 720), (333, 222), (64, 48), (800, 600)]
    for k, (W, H) in enumerate(sizes):
        _write_case_png(tmp_path, k, W, H)
    (tmp_path / 'notes.txt').write_text('not an image')
    monkeypatch.setenv('DUST3R_AMD_LOAD_THREADS', '8')
    many = load_images(str(tmp_path), size=512, verbose=False)
    monkeypatch.setenv('DUST3R_AMD_LOAD_THREADS', '1')
    one = load_images(str(tmp_path), size=512, verbose=False)
    assert len(many) == len(one) == len(sizes)
    for k, (a, b) in enumerate(zip(many, one)):
        assert torch.equal(a['img'], b['img']) and (a['true_shape'] == b['true_shape']).all() and a['idx'] == b['idx'] == k and a['instance'] == b['instance'] == str(k)


def test_host_thread_pool_is_capped_at_the_usable_cpus(monkeypatch):
    """utils/device.py: usable_cpus() = min(affinity, cgroup quota); fit_host_threads() lowers torch's intra-op pool to it, never raises it, and
    DUST3R_AMD_KEEP_TORCH_THREADS=1 leaves torch alone. Importing the package does NOT apply it (no import side effect on the host application's pool, round 6):
    the first inference() / global_aligner() call does, once (fit_host_threads_once)."""
    import subprocess
    import sys
    code = ('import torch; torch.set_num_threads(3); import dust3r_amd, dust3r_amd.inference, dust3r_amd.cloud_opt; '
            'from dust3r_amd.utils import device as D; D.usable_cpus = lambda cap=64: 2; a = torch.get_num_threads(); '
            'D.fit_host_threads_once(); b = torch.get_num_threads(); torch.set_num_threads(3); D.fit_host_threads_once(); print(a, b, torch.get_num_threads())')
    env = {k: v for k, v in os.environ.items() if k not in ('DUST3R_AMD_KEEP_TORCH_THREADS', 'LOCAL_WORLD_SIZE')}
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-3:] == ['3', '2', '3'], out.stdout       # import: untouched; first entry-point call: capped; later calls: nothing
    from dust3r_amd.utils import device as D
    n = D.usable_cpus()
    assert 1 <= n <= max(1, len(os.sched_getaffinity(0)))
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(D, 'usable_cpus', lambda cap=64: 1)
        monkeypatch.setenv('DUST3R_AMD_KEEP_TORCH_THREADS', '1')
        assert D.fit_host_threads() == before
        monkeypatch.delenv('DUST3R_AMD_KEEP_TORCH_THREADS')
        assert D.fit_host_threads() == 1 and torch.get_num_threads() == 1
        monkeypatch.setattr(D, 'usable_cpus', lambda cap=64: 64)
        assert D.fit_host_threads() == 1                      # never raised
    finally:
        torch.set_num_threads(before)


@pytest.mark.parametrize('fmt', ['safetensors', 'bin'])
def test_from_pretrained_loads_a_local_hub_snapshot_directory(tmp_path, fmt):
    """dust3r/model.py:76-85: from_pretrained(<not a file>) is huggingface_hub.PyTorchModelHubMixin.from_pretrained -- for a local directory: config.json = the
    constructor's keyword arguments, model.safetensors / pytorch_model.bin = the state dict (non-strict). Built here from a synthetic state dict with the
    released configs' quirks: lists for tuples, Infinity bounds, ManyAR patch embed, a `freeze` entry, no dec_blocks2 keys (model.py:91-98)."""
    import json
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, synthetic_state_dict
    cfg = dict(MODEL_CONFIGS['tiny_dpt'])
    ref = AsymmetricCroCo3DStereo(landscape_only=False, **cfg)
    state = synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in ref._spec.items()}, 5, 1.0)
    state = {k: v for k, v in state.items() if not k.startswith('dec_blocks2')}
    snap = tmp_path / 'snapshot'
    snap.mkdir()
    cfg_json = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    cfg_json.update(landscape_only=False, patch_embed_cls='ManyAR_PatchEmbed', freeze='none', depth_mode=['exp', float('-inf'), float('inf')], conf_mode=['exp', 1, float('inf')])
    (snap / 'config.json').write_text(json.dumps(cfg_json))
    assert 'Infinity' in (snap / 'config.json').read_text()
    if fmt == 'safetensors':
        from safetensors.torch import save_file
        # as huggingface_hub's mixin writes it: ONE name per shared tensor (the DPT head's layer_rn convolutions are registered under two)
        save_file({k: v.contiguous().clone() for k, v in state.items() if not re.search(r'scratch\.layer\d_rn\.', k)}, str(snap / 'model.safetensors'))
    else:
        torch.save(state, str(snap / 'pytorch_model.bin'))
    m = AsymmetricCroCo3DStereo.from_pretrained(str(snap), precision='fp32')
    assert m.precision == 'fp32' and m.depth_mode == ('exp', float('-inf'), float('inf')) and m.conf_mode == ('exp', 1, float('inf'))
    got = m.state_dict()
    assert set(got) == set(ref._spec)
    for k, v in state.items():
        assert torch.equal(got[k], v.float()), k
        if k.startswith('dec_blocks.'):
            assert torch.equal(got[k.replace('dec_blocks', 'dec_blocks2')], v.float())     # the second decoder starts as a copy of the first
    # a hub id without a local snapshot: the reference's failure message, no connection attempt that could hang
    with pytest.raises(Exception, match='tried to load naver/DUSt3R_does_not_exist from huggingface, but failed'):
        AsymmetricCroCo3DStereo.from_pretrained('naver/DUSt3R_does_not_exist')
    with pytest.raises(Exception, match='no config.json'):
        AsymmetricCroCo3DStereo.from_pretrained(str(tmp_path))


def test_persistent_gemm_has_no_scratch():
    """gemm_p4.hip keeps TWO accumulator sets (256 registers) per wave at one wave per SIMD: it only works while hipcc keeps every accumulator in a register --
    one dynamically indexed access, one unrolled loop past the size cap, and the arrays move to scratch (90 instead of 400 TFLOP/s: DESIGN.md 4.1). The build
    writes hipcc's resource report of the file next to its object (dust3r_amd/build.py): every kernel instance must show no scratch and no spilled VGPR."""
    from dust3r_amd.build import CSRC, build
    rep = os.path.join(CSRC, 'gemm_p4.resources.txt')
    if not os.path.exists(rep) or os.path.getmtime(rep) < os.path.getmtime(os.path.join(CSRC, 'gemm_p4.hip')):
        os.utime(os.path.join(CSRC, 'gemm_p4.hip'))
        build(force=False, verbose=False)
    txt = open(rep).read()
    kernels = re.findall(r'Function Name: (\S*gemm_p4_kernel\S*)', txt)
    scratch = [int(x) for x in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', txt)]
    spills = [int(x) for x in re.findall(r'VGPRs Spill: (\d+)', txt)]
    occ = [int(x) for x in re.findall(r'Occupancy \[waves/SIMD\]: (\d+)', txt)]
    assert len(kernels) >= 5 and len(scratch) == len(kernels) == len(spills)
    assert all(v == 0 for v in scratch) and all(v == 0 for v in spills), list(zip(kernels, scratch, spills))
    assert all(v == 1 for v in occ)          # one wave per SIMD: the whole register file
