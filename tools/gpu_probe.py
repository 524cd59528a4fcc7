"""On-GPU performance probe (not a test): times the dominant kernels and the full forward / aligner step.
Usage: python tools/gpu_probe.py [gemm] [attn] [forward] [aligner]"""
import math
import sys
import time

import torch

sys.path.insert(0, '.')
from dust3r_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def gemm_probe():
    import os
    print('== GEMM: ms and TFLOP/s per tile configuration (D3R_GEMM_CFG: 0 = 128x128, 1 = 256x256, 2 = 256x128, 3 = 512x128, 4 = 256x128 4-wave 2 blocks/CU, 5 = 256x256 4-stage, 6 = 256x256 ping-pong, auto = heuristic)')
    from dust3r_amd._lib import lib, ptr, current_stream, check
    shapes = [(49152, 3072, 1024), (49152, 1024, 1024), (49152, 4096, 1024), (49152, 1024, 4096), (24576, 2304, 768), (24576, 768, 768),
              (24576, 3072, 768), (24576, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]
    for (M, N, K) in shapes:
        for dt in (torch.bfloat16, torch.float32):
            if dt == torch.float32 and M * N * K > 2e11:
                continue
            a = torch.randn((M, K), device=dev).to(dt)
            w = ops.pad_rows((torch.randn((N, K), device=dev) / math.sqrt(K)).to(dt))
            b = ops.pad_rows(torch.randn(N, device=dev))
            out = torch.empty((M, N), dtype=dt, device=dev)

            def run():
                check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), None, M, N, K, 0, ops._dt(a), current_stream()))
            line = f'  M={M} N={N} K={K} {str(dt)[6:]:9s}'
            for cfg in (('0', '1', '5', '6', None) if dt == torch.bfloat16 else (None,)):
                if cfg is None:
                    os.environ.pop('D3R_GEMM_CFG', None)
                else:
                    os.environ['D3R_GEMM_CFG'] = cfg
                ms = timeit(run)
                line += f' | cfg {cfg or "auto"}: {ms:7.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF/s'
            os.environ.pop('D3R_GEMM_CFG', None)
            if dt == torch.bfloat16:
                os.environ['D3R_GEMM_NOSTORE'] = '1'     # same launch without the epilogue's memory traffic
                ms = timeit(run)
                os.environ.pop('D3R_GEMM_NOSTORE', None)
                line += f' | auto/no-store: {ms:7.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF/s'
                os.environ['D3R_GEMM_NOWIDE'] = '1'      # A/B: direct fragment stores instead of the LDS-staged wide rows
                ms = timeit(run)
                os.environ.pop('D3R_GEMM_NOWIDE', None)
                line += f' | auto/narrow-store: {ms:7.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF/s'
                ms2 = timeit(lambda: torch.nn.functional.linear(a, w[:N], None))
                line += f' | hipBLASLt {ms2:7.3f} ms {2 * M * N * K / ms2 / 1e9:7.1f} TF/s'
            print(line)
    from oracle import tune_threads
    print('cpu threads chosen:', tune_threads(verbose=True))


def gemmtrace_probe():
    """Where does a tile's time go? Per-block wall-clock stamps (d3r_gemm_set_trace) of isolated GEMMs in the split-fp16 mode with the
    network's epilogues: entry -> K-loop start (prologue), K loop, epilogue until its last store is issued, store drain; and the gap
    between a block's end and the start of the next block on the same CU (dispatch + launch overhead)."""
    import numpy as np
    import os
    from dust3r_amd._lib import lib, ptr, current_stream, check, DTYPE_F16X3, DTYPE_F16F8
    f8 = os.environ.get('D3R_PROBE_DT', 'fp16x3') == 'fp16f8'     # D3R_PROBE_DT=fp16f8: the fp16 + fp8 operand rows of the default engine's block linears
    print(f"== GEMM phase trace ({'fp16f8' if f8 else 'fp16x3'}; wall-clock ticks converted to us)")
    rate = 100e6
    for (M, N, K, epi, name) in [(49152, 1024, 1024, 1, 'proj + fp32 residual'), (49152, 4096, 1024, 2, 'fc1 + GELU'), (49152, 1024, 4096, 1, 'fc2 + fp32 residual'),
                                 (49152, 3072, 1024, 0, 'plain typed store'), (24576, 768, 768, 1, 'decoder 768 + fp32 residual')]:
        if f8:
            a = ops.pack_f8(torch.randn((M, K), device=dev))
            w = ops.pad_rows(ops.pack_f8(torch.randn((N, K), device=dev) / math.sqrt(K), weight=True))
        else:
            a = ops.pack_x3(torch.randn((M, K), device=dev))
            w = ops.pad_rows(ops.pack_x3(torch.randn((N, K), device=dev) / math.sqrt(K)))
        b = ops.pad_rows(torch.randn(N, device=dev))
        res = torch.randn((M, N), device=dev) if epi == 1 else None
        out = torch.empty((M, N), dtype=torch.float32, device=dev) if epi == 1 else torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
        nblk = ((M + 127) // 128) * ((N + 127) // 128)
        buf = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
        dtc = DTYPE_F16F8 if f8 else DTYPE_F16X3

        def run():
            check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), ptr(res), M, N, K, epi, dtc, current_stream()))

        def traced(tag):
            for _ in range(2):
                run()
            ms = timeit(run, warm=1, reps=5)
            buf.zero_()
            check(lib.d3r_gemm_set_trace(ptr(buf), nblk))
            run()
            torch.cuda.synchronize()
            check(lib.d3r_gemm_set_trace(None, 0))
            t = buf.cpu().numpy()
            t = t[t[:, 0] > 0]
            us = lambda x: x / rate * 1e6  # noqa: E731
            pro, kl, ep, dr = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 4] - t[:, 3])
            span = us(t[:, 4].max() - t[:, 0].min())
            # per CU: (xcc, se, cu) from XCC_ID / HW_ID; gap between consecutive blocks on the same CU; "round" = position in the CU's sequence
            hw, xcc = t[:, 5], t[:, 6] & 0xF
            cu_key = (xcc << 16) | (hw & 0xFF00)            # se_id[15:13] sh_id[12] cu_id[11:8]
            gaps, rnd = [], np.zeros(len(t), dtype=np.int64)
            for key in np.unique(cu_key):
                idx = np.where(cu_key == key)[0]
                idx = idx[np.argsort(t[idx, 0])]
                rnd[idx] = np.arange(len(idx))
                if len(idx) > 1:
                    gaps.extend(us(t[idx[1:], 0] - t[idx[:-1], 4]))
            gaps = np.array(gaps) if gaps else np.zeros(1)
            print(f'  {name:28s} [{tag}] M={M} N={N} K={K}: {ms:.3f} ms = {2 * M * N * K / ms / 1e9:6.1f} TF/s; {len(t)} blocks on {len(np.unique(cu_key))} CUs, traced span {span:.0f} us')
            print(f'      per block (mean / p90 us): prologue {pro.mean():5.1f} / {np.percentile(pro, 90):5.1f}   K loop {kl.mean():6.1f} / {np.percentile(kl, 90):6.1f}   '
                  f'epilogue {ep.mean():5.1f} / {np.percentile(ep, 90):5.1f}   drain {dr.mean():5.1f} / {np.percentile(dr, 90):5.1f}   gap to next block on the CU {gaps.mean():5.1f} / {np.percentile(gaps, 90):5.1f}')
            # does a tile's K loop run slower when it starts behind other tiles' epilogue traffic? K loop / epilogue by position on the CU,
            # and the spread of epilogue start times inside a round (lockstep = all CUs store at once)
            nr = int(rnd.max()) + 1
            rows = []
            for r in range(min(nr, 6)):
                sel = rnd == r
                rows.append(f'r{r}: K {kl[sel].mean():6.1f} epi {ep[sel].mean():5.1f} epi-start spread {us(np.percentile(t[sel, 2], 90) - np.percentile(t[sel, 2], 10)):6.1f}')
            print('      by position on the CU (us): ' + ' | '.join(rows))
        traced('random operands')
        if not f8:
            os.environ['D3R_GEMM_CFG'] = '7'
            traced('cfg 7: 256x128, weights in registers, 2 blocks / CU')
            os.environ.pop('D3R_GEMM_CFG')
        if os.environ.get('D3R_PROBE_EXTRA', '1') == '1':
            os.environ['D3R_GEMM_NOSTORE'] = '1'
            traced('same, epilogue skipped (D3R_GEMM_NOSTORE)')
            os.environ.pop('D3R_GEMM_NOSTORE')
            a.zero_()
            traced('activation operand all zero (data-dependent MFMA power)')
            if not f8:
                a.copy_(ops.pack_x3(torch.randn((M, K), device=dev)))
            os.environ['D3R_GEMM_STAGGER'] = '1.0'
            traced('first-round stagger 1.0')
            os.environ.pop('D3R_GEMM_STAGGER')
        del a, w, out, buf


def conv_probe():
    import os
    print('== DPT-head convolutions (implicit GEMM, NHWC bf16, B = 32 images): ms and TFLOP/s per tile configuration')
    for (H, W, Cin, Cout) in [(192, 256, 256, 128), (384, 512, 128, 128), (96, 128, 256, 256), (192, 256, 256, 256), (48, 64, 256, 256)]:
        B = 32
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((Cout, Cin, 3, 3), device=dev) / math.sqrt(9 * Cin)).to(torch.bfloat16)
        b = torch.randn(Cout, device=dev)
        fl = 2 * B * H * W * Cout * 9 * Cin
        line = f'  {H}x{W} {Cin}->{Cout}'
        for cfg in ('1', '3', '6', None):
            if cfg is None:
                os.environ.pop('D3R_GEMM_CFG', None)
            else:
                os.environ['D3R_GEMM_CFG'] = cfg
            ms = timeit(lambda: ops.conv2d_nhwc(x, w, b, stride=1, pad=1), warm=2, reps=5)
            line += f' | cfg {cfg or "auto"}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s'
        os.environ.pop('D3R_GEMM_CFG', None)
        print(line)
        del x


def attn_probe():
    print('== attention (N=768, d=64)')
    for (B, H, dt) in [(64, 16, torch.bfloat16), (64, 16, torch.float16), (32, 12, torch.bfloat16), (8, 16, torch.float32)]:
        N = 768
        q = torch.randn((B, H, N, 64), device=dev).to(dt)
        k = torch.randn((B, H, N, 64), device=dev).to(dt)
        vt = torch.randn((B, H, 64, N), device=dev).to(dt)
        ms = timeit(lambda: ops.attention(q, k, vt))
        fl = 4 * B * H * N * N * 64
        print(f'  B={B} H={H} {str(dt)[6:]:9s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s')


def attndma_probe():
    """Split-fp16 attention: K / V^T tiles staged through registers (D3R_ATTN_DMA=0) vs by global_load_lds DMA into swizzled rows (=1), alternating,
    encoder (64 x 16 heads) and decoder (32 x 12) launch shapes of the 32-pair step, random data."""
    import os
    from dust3r_amd import _lib
    from dust3r_amd._lib import lib
    from dust3r_amd.ops import check, current_stream, pack_x3, ptr
    N = 768
    print('== split-fp16 attention: register staging vs DMA staging (us per launch, TF/s algorithmic)')
    for (b, h) in ((64, 16), (32, 12)):
        q, k = torch.randn((b, h, N, 64), device=dev) * 0.5, torch.randn((b, h, N, 64), device=dev) * 0.5
        vt = torch.randn((b, h, 64, N), device=dev)
        qp, kp, vp = pack_x3(q), pack_x3(k), pack_x3(vt)
        out = torch.empty((b, N, h * 64 * 2), dtype=torch.float16, device=dev)
        for dma in ('0', '1', '0', '1'):
            os.environ['D3R_ATTN_DMA'] = dma
            ms = timeit(lambda: check(lib.d3r_attention(ptr(qp), ptr(kp), ptr(vp), ptr(out), b, h, N, N, N, 0.125, _lib.DTYPE_F16X3, current_stream()), 'attention'), warm=3, reps=20)
            print(f'  D3R_ATTN_DMA={dma} B={b} H={h}: {ms * 1e3:8.1f} us  {4 * b * h * N * N * 64 / ms / 1e9:7.1f} TF/s', flush=True)
    os.environ.pop('D3R_ATTN_DMA')


def attnparts_probe():
    """The split-fp16 attention kernel with parts of its instruction stream removed (D3R_ATTN_PROBE bit mask: 1 no softmax / split VALU,
    2 no MFMAs, 4 no per-tile barrier, 8 no staging of the next tiles; results invalid): what each part costs next to the others."""
    import os
    from dust3r_amd import _lib
    from dust3r_amd._lib import lib
    from dust3r_amd.ops import check, current_stream, pack_x3, ptr
    B, H, N = 64, 16, 768
    q, k = torch.randn((B, H, N, 64), device=dev), torch.randn((B, H, N, 64), device=dev)
    vt = torch.randn((B, H, 64, N), device=dev)
    qp, kp, vp = pack_x3(q), pack_x3(k), pack_x3(vt)
    out = torch.empty((B, N, H * 64 * 2), dtype=torch.float16, device=dev)
    fl = 4 * B * H * N * N * 64
    print(f'== split-fp16 attention, B={B} H={H} N={N}: ablation (D3R_ATTN_PROBE)')
    names = {0: 'full kernel', 1: 'no VALU', 2: 'no MFMA', 4: 'no barrier', 8: 'no staging', 12: 'no barrier, no staging', 13: 'MFMA + fragment reads only',
             14: 'VALU only', 16: 'no LDS writes of the staging', 32: 'no global loads of the staging'}
    for probe, name in names.items():
        os.environ['D3R_ATTN_PROBE'] = str(probe)
        ms = timeit(lambda: check(lib.d3r_attention(ptr(qp), ptr(kp), ptr(vp), ptr(out), B, H, N, N, N, 0.125, _lib.DTYPE_F16X3, current_stream()), 'attention'), warm=3, reps=20)
        print(f'  probe {probe:2d} {name:28s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s-equivalent', flush=True)
    os.environ.pop('D3R_ATTN_PROBE')
    for nw in ('4', '8', '4', '8'):         # 128 vs 256 queries per workgroup
        os.environ['D3R_ATTN_NW'] = nw
        for (b, h) in ((64, 16), (32, 12)):
            q, k = torch.randn((b, h, N, 64), device=dev) * 0.5, torch.randn((b, h, N, 64), device=dev) * 0.5
            vt = torch.randn((b, h, 64, N), device=dev)
            qp, kp, vp = pack_x3(q), pack_x3(k), pack_x3(vt)
            out = torch.empty((b, N, h * 64 * 2), dtype=torch.float16, device=dev)
            ms = timeit(lambda: check(lib.d3r_attention(ptr(qp), ptr(kp), ptr(vp), ptr(out), b, h, N, N, N, 0.125, _lib.DTYPE_F16X3, current_stream()), 'attention'), warm=3, reps=20)
            print(f'  D3R_ATTN_NW={nw} B={b} H={h}: {ms * 1e3:8.1f} us  {4 * b * h * N * N * 64 / ms / 1e9:7.1f} TF/s', flush=True)
    os.environ.pop('D3R_ATTN_NW')


def forward_probe():
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_state_dict, synthetic_views
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    print('== forward', cfg)
    m = AsymmetricCroCo3DStereo(precision='bf16', landscape_only=False, **MODEL_CONFIGS[cfg])
    t = time.time()
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device=dev))
    print(f'  synthetic weights {time.time() - t:.1f}s')
    t = time.time()
    m.to(dev)
    print(f'  upload+pack {time.time() - t:.1f}s  weights {m.device_bytes() / 2**30:.2f} GiB')
    # precision of the 16-bit modes against the engine's own exact-fp32 mode, full-size model, 2 pairs
    v1, v2 = synthetic_views(2, 384, 512, seed=0, device=dev)
    m.set_precision('fp32')
    r1, r2 = m(v1, v2)
    ref = [r1['pts3d'].clone(), r2['pts3d_in_other_view'].clone(), r1['conf'].clone()]
    for prec in ('fp16x3', 'fp16', 'bf16'):
        m.set_precision(prec)
        e1, e2 = m(v1, v2)
        for name, a, b in (('pts1', e1['pts3d'], ref[0]), ('pts2', e2['pts3d_in_other_view'], ref[1])):
            rel = (a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-8)
            print(f'  {prec} vs fp32-engine {name}: rel err max {float(rel.max()):.3e} p99 {float(rel.flatten().quantile(0.99)):.3e} mean {float(rel.mean()):.3e}')
        rc = ((e1['conf'] - ref[2]).abs() / ref[2]).max()
        print(f'  {prec} vs fp32-engine conf1: rel err max {float(rc):.3e}')
    for prec in ('bf16', 'fp16x3', 'fp32'):
        m.set_precision(prec)
        for B in ((1, 8, 32) if prec != 'fp32' else (4,)):
            v1, v2 = synthetic_views(B, 384, 512, seed=0, device=dev)
            ms = timeit(lambda: m(v1, v2), warm=2, reps=3 if B >= 16 else 5)
            print(f'  {prec} B={B:3d}: {ms:9.2f} ms/forward  {B / ms * 1e3:8.2f} pairs/s  {B * 1856.8 / ms:8.1f} TF/s eff  mem {m.device_bytes() / 2**30:.1f} GiB')


def cache_probe():
    """BASELINE configs[2] on one GPU: 20 views -> 190 pairs (complete graph), forward only, device-resident tensors:
    pair-by-pair (every pair re-encodes both views, as the reference does) vs encode-once (20 encoder passes + 190 decodes)."""
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, scene_edges, synthetic_state_dict
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    print('== 20 views -> 190 pairs, forward only', cfg)
    m = AsymmetricCroCo3DStereo(precision='bf16', landscape_only=False, **MODEL_CONFIGS[cfg])
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device=dev))
    m.to(dev)
    n, H, W, bs = 20, 384, 512, 32
    imgs = torch.rand((n, 3, H, W), device=dev) * 2 - 1
    edges = scene_edges(n, 'complete', False)
    i1 = torch.tensor([i for i, j in edges], device=dev)
    i2 = torch.tensor([j for i, j in edges], device=dev)

    def plain():
        for s in range(0, len(edges), bs):
            m(dict(img=imgs[i1[s:s + bs]]), dict(img=imgs[i2[s:s + bs]]))

    def cached():
        feats = m.encode_images(imgs)
        for s in range(0, len(edges), bs):
            m.decode_pairs(feats.index_select(0, torch.cat((i1[s:s + bs], i2[s:s + bs]))), H, W)
    for name, fn in (('pair-by-pair', plain), ('encode-once', cached)):
        ms = timeit(fn, warm=1, reps=3)
        print(f'  {name:13s}: {ms:8.1f} ms for {len(edges)} pairs = {len(edges) / ms * 1e3:7.1f} pairs/s')


def aligner_probe():
    from dust3r_amd.cloud_opt import global_aligner
    from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
    from dust3r_amd.synthetic import synthetic_scene
    print('== aligner')
    for (n, sym) in ((20, False), (20, True)):
        out, init, gt = synthetic_scene(n, 384, 512, seed=0, symmetrize=sym, device=dev)
        scene = global_aligner(out, dev, verbose=False)
        scene.load_state_dict(init)
        E = scene.n_edges
        global_alignment_loop(scene, niter=10)
        torch.cuda.synchronize()
        t = time.time()
        loss = global_alignment_loop(scene, niter=300)
        torch.cuda.synchronize()
        dt = time.time() - t
        by = E * 196608 * 32 + n * 196608 * 4 * 6
        print(f'  n={n} E={E}: 300 iters {dt * 1e3:.1f} ms  {300 / dt:.1f} it/s  {by * 300 / dt / 1e12:.2f} TB/s algorithmic  final loss {loss:.5f}')
        del scene, out


def tune_probe():
    """A/B switches measured on the whole forward (B = 32, bf16, two-stream, as bench.py times it) and on isolated GEMMs with the
    network's epilogues (bf16 out / fp32 residual stream): D3R_GEMM_NT=0 (plain instead of non-temporal epilogue stores),
    D3R_GEMM_T256 (256x256 / 128x128 crossover, default 700 tiles)."""
    import os
    from dust3r_amd._lib import lib, ptr, current_stream, check
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_state_dict, synthetic_views
    print('== tune: isolated GEMMs (TF/s): bf16 store | fp32 out + fp32 residual (the residual-stream epilogue) | GELU; nt on / off')
    for (M, N, K) in ((49152, 1024, 1024), (49152, 1024, 4096), (24576, 768, 768), (24576, 768, 3072)):
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = ops.pad_rows((torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16))
        b = ops.pad_rows(torch.randn(N, device=dev))
        out16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        out32 = torch.randn((M, N), dtype=torch.float32, device=dev)
        line = f'  M={M} N={N} K={K}'
        for name, epi, o, r in (('store', 0, out16, None), ('f32+res', 1, out32, out32), ('gelu', 2, out16, None)):
            def run():
                check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(o), ptr(r), M, N, K, epi, ops._dt(a), current_stream()))
            for c in ((None, '0', '2', '4') if epi == 1 else (None,)):
                if c is None:
                    os.environ.pop('D3R_GEMM_CFG', None)
                else:
                    os.environ['D3R_GEMM_CFG'] = c
                ms = timeit(run, warm=2, reps=6)
                line += f' | {name} cfg {c or "auto"}: {2 * M * N * K / ms / 1e9:6.1f}'
            os.environ.pop('D3R_GEMM_CFG', None)
        print(line)
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    m = AsymmetricCroCo3DStereo(precision='bf16', landscape_only=False, **MODEL_CONFIGS[cfg])
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device=dev))
    m.to(dev)
    v1, v2 = synthetic_views(32, 384, 512, seed=0, device=dev)
    print('== tune: forward B=32 bf16 (two streams)')
    variants = [('default', {}), ('f32 epilogue K<=1024 on 128x128', {'D3R_GEMM_F32CFG': '0'}), ('f32 epilogue K<=1024 on 256x128', {'D3R_GEMM_F32CFG': '2'}),
                ('f32 epilogue K<=1024 on 256x128 w4', {'D3R_GEMM_F32CFG': '4'}), ('T256=250', {'D3R_GEMM_T256': '250'}), ('default again', {})]
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        ms = timeit(lambda: m(v1, v2), warm=2, reps=4)
        for k in env:
            os.environ.pop(k, None)
        print(f'  {name:26s}: {ms:8.2f} ms/forward  {32 / ms * 1e3:7.2f} pairs/s')
    m.set_two_streams(False)
    ms = timeit(lambda: m(v1, v2), warm=2, reps=4)
    print(f'  {"single stream":26s}: {ms:8.2f} ms/forward  {32 / ms * 1e3:7.2f} pairs/s')


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'attn', 'forward', 'aligner']
    print(torch.cuda.get_device_name(0))
    for w in which:
        try:
            {'gemm': gemm_probe, 'gemmtrace': gemmtrace_probe, 'tune': tune_probe, 'cache': cache_probe, 'conv': conv_probe, 'attn': attn_probe, 'attnparts': attnparts_probe, 'attndma': attndma_probe, 'forward': forward_probe, 'aligner': aligner_probe}[w]()
        except Exception as e:  # keep going: this is a probe
            import traceback
            traceback.print_exc()
            print('!!', w, 'failed:', e)
